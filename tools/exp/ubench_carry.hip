// Instruction issue rates of the integer VALU instructions the Goldilocks / 28-bit-limb kernels are made
// of (gfx950).  Each kernel runs ITER iterations of a block of 16 instructions over 8 independent
// register chains; the grid puts exactly W waves on every SIMD.  Reported: wave-instructions per clock
// per SIMD at the nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32;
typedef unsigned long long u64;

#define KERNEL(name, ASMBLOCK)                                                         \
__global__ __launch_bounds__(256) void name(u32* out, u32 seed, int iters)              \
{                                                                                       \
    u32 a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    u32 b0 = a0 ^ 0x9e3779b9u, b1 = a1 ^ 0x7f4a7c15u, b2 = a2 + 1, b3 = a3 + 2, b4 = a4 + 3, b5 = a5 + 4, b6 = a6 + 5, b7 = a7 + 6; \
    for (int i = 0; i < iters; i++) {                                                   \
        asm volatile(ASMBLOCK                                                           \
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
              "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) \
            : : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");         \
    }                                                                                   \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7; \
}

// 16 instructions per block in every variant
KERNEL(k_add_u32,
    "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %9\n v_add_u32 %2, %2, %10\n v_add_u32 %3, %3, %11\n"
    "v_add_u32 %4, %4, %12\n v_add_u32 %5, %5, %13\n v_add_u32 %6, %6, %14\n v_add_u32 %7, %7, %15\n"
    "v_add_u32 %8, %8, %0\n v_add_u32 %9, %9, %1\n v_add_u32 %10, %10, %2\n v_add_u32 %11, %11, %3\n"
    "v_add_u32 %12, %12, %4\n v_add_u32 %13, %13, %5\n v_add_u32 %14, %14, %6\n v_add_u32 %15, %15, %7\n")
KERNEL(k_addco_vcc_pairs,       // 64-bit adds: add_co (vcc out) + addc_co (vcc in/out), VOP2 encodings
    "v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %9, vcc\n v_add_co_u32 %2, vcc, %2, %10\n v_addc_co_u32 %3, vcc, %3, %11, vcc\n"
    "v_add_co_u32 %4, vcc, %4, %12\n v_addc_co_u32 %5, vcc, %5, %13, vcc\n v_add_co_u32 %6, vcc, %6, %14\n v_addc_co_u32 %7, vcc, %7, %15, vcc\n"
    "v_add_co_u32 %8, vcc, %8, %0\n v_addc_co_u32 %9, vcc, %9, %1, vcc\n v_add_co_u32 %10, vcc, %10, %2\n v_addc_co_u32 %11, vcc, %11, %3, vcc\n"
    "v_add_co_u32 %12, vcc, %12, %4\n v_addc_co_u32 %13, vcc, %13, %5, vcc\n v_add_co_u32 %14, vcc, %14, %6\n v_addc_co_u32 %15, vcc, %15, %7, vcc\n")
KERNEL(k_addco_sgpr_pairs,      // the same with SGPR-pair carries (VOP3), four carry registers in rotation
    "v_add_co_u32 %0, s[20:21], %0, %8\n v_add_co_u32 %2, s[22:23], %2, %10\n v_add_co_u32 %4, s[24:25], %4, %12\n v_add_co_u32 %6, s[26:27], %6, %14\n"
    "v_addc_co_u32 %1, s[20:21], %1, %9, s[20:21]\n v_addc_co_u32 %3, s[22:23], %3, %11, s[22:23]\n v_addc_co_u32 %5, s[24:25], %5, %13, s[24:25]\n v_addc_co_u32 %7, s[26:27], %7, %15, s[26:27]\n"
    "v_add_co_u32 %8, s[20:21], %8, %0\n v_add_co_u32 %10, s[22:23], %10, %2\n v_add_co_u32 %12, s[24:25], %12, %4\n v_add_co_u32 %14, s[26:27], %14, %6\n"
    "v_addc_co_u32 %9, s[20:21], %9, %1, s[20:21]\n v_addc_co_u32 %11, s[22:23], %11, %3, s[22:23]\n v_addc_co_u32 %13, s[24:25], %13, %5, s[24:25]\n v_addc_co_u32 %15, s[26:27], %15, %7, s[26:27]\n")
KERNEL(k_addco_only_vcc,        // carry-out only (result of the carry unused)
    "v_add_co_u32 %0, vcc, %0, %8\n v_add_co_u32 %1, vcc, %1, %9\n v_add_co_u32 %2, vcc, %2, %10\n v_add_co_u32 %3, vcc, %3, %11\n"
    "v_add_co_u32 %4, vcc, %4, %12\n v_add_co_u32 %5, vcc, %5, %13\n v_add_co_u32 %6, vcc, %6, %14\n v_add_co_u32 %7, vcc, %7, %15\n"
    "v_add_co_u32 %8, vcc, %8, %0\n v_add_co_u32 %9, vcc, %9, %1\n v_add_co_u32 %10, vcc, %10, %2\n v_add_co_u32 %11, vcc, %11, %3\n"
    "v_add_co_u32 %12, vcc, %12, %4\n v_add_co_u32 %13, vcc, %13, %5\n v_add_co_u32 %14, vcc, %14, %6\n v_add_co_u32 %15, vcc, %15, %7\n")
KERNEL(k_add3_u32,
    "v_add3_u32 %0, %0, %8, %1\n v_add3_u32 %1, %1, %9, %2\n v_add3_u32 %2, %2, %10, %3\n v_add3_u32 %3, %3, %11, %4\n"
    "v_add3_u32 %4, %4, %12, %5\n v_add3_u32 %5, %5, %13, %6\n v_add3_u32 %6, %6, %14, %7\n v_add3_u32 %7, %7, %15, %0\n"
    "v_add3_u32 %8, %8, %0, %9\n v_add3_u32 %9, %9, %1, %10\n v_add3_u32 %10, %10, %2, %11\n v_add3_u32 %11, %11, %3, %12\n"
    "v_add3_u32 %12, %12, %4, %13\n v_add3_u32 %13, %13, %5, %14\n v_add3_u32 %14, %14, %6, %15\n v_add3_u32 %15, %15, %7, %8\n")
KERNEL(k_cmp_cndmask,           // compare into vcc + select
    "v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_u32 vcc, %1, %9\n v_cndmask_b32 %1, %1, %9, vcc\n"
    "v_cmp_lt_u32 vcc, %2, %10\n v_cndmask_b32 %2, %2, %10, vcc\n v_cmp_lt_u32 vcc, %3, %11\n v_cndmask_b32 %3, %3, %11, vcc\n"
    "v_cmp_lt_u32 vcc, %4, %12\n v_cndmask_b32 %4, %4, %12, vcc\n v_cmp_lt_u32 vcc, %5, %13\n v_cndmask_b32 %5, %5, %13, vcc\n"
    "v_cmp_lt_u32 vcc, %6, %14\n v_cndmask_b32 %6, %6, %14, vcc\n v_cmp_lt_u32 vcc, %7, %15\n v_cndmask_b32 %7, %7, %15, vcc\n")
KERNEL(k_mul_lo_u32,
    "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %9\n v_mul_lo_u32 %2, %2, %10\n v_mul_lo_u32 %3, %3, %11\n"
    "v_mul_lo_u32 %4, %4, %12\n v_mul_lo_u32 %5, %5, %13\n v_mul_lo_u32 %6, %6, %14\n v_mul_lo_u32 %7, %7, %15\n"
    "v_mul_lo_u32 %8, %8, %0\n v_mul_lo_u32 %9, %9, %1\n v_mul_lo_u32 %10, %10, %2\n v_mul_lo_u32 %11, %11, %3\n"
    "v_mul_lo_u32 %12, %12, %4\n v_mul_lo_u32 %13, %13, %5\n v_mul_lo_u32 %14, %14, %6\n v_mul_lo_u32 %15, %15, %7\n")
KERNEL(k_mul_u32_u24,
    "v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %9\n v_mul_u32_u24 %2, %2, %10\n v_mul_u32_u24 %3, %3, %11\n"
    "v_mul_u32_u24 %4, %4, %12\n v_mul_u32_u24 %5, %5, %13\n v_mul_u32_u24 %6, %6, %14\n v_mul_u32_u24 %7, %7, %15\n"
    "v_mul_u32_u24 %8, %8, %0\n v_mul_u32_u24 %9, %9, %1\n v_mul_u32_u24 %10, %10, %2\n v_mul_u32_u24 %11, %11, %3\n"
    "v_mul_u32_u24 %12, %12, %4\n v_mul_u32_u24 %13, %13, %5\n v_mul_u32_u24 %14, %14, %6\n v_mul_u32_u24 %15, %15, %7\n")
KERNEL(k_mad_u32_u24,
    "v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %9, %2\n v_mad_u32_u24 %2, %2, %10, %3\n v_mad_u32_u24 %3, %3, %11, %4\n"
    "v_mad_u32_u24 %4, %4, %12, %5\n v_mad_u32_u24 %5, %5, %13, %6\n v_mad_u32_u24 %6, %6, %14, %7\n v_mad_u32_u24 %7, %7, %15, %0\n"
    "v_mad_u32_u24 %8, %8, %0, %9\n v_mad_u32_u24 %9, %9, %1, %10\n v_mad_u32_u24 %10, %10, %2, %11\n v_mad_u32_u24 %11, %11, %3, %12\n"
    "v_mad_u32_u24 %12, %12, %4, %13\n v_mad_u32_u24 %13, %13, %5, %14\n v_mad_u32_u24 %14, %14, %6, %15\n v_mad_u32_u24 %15, %15, %7, %8\n")
KERNEL(k_mul_hi_u32_u24,
    "v_mul_hi_u32_u24 %0, %0, %8\n v_mul_hi_u32_u24 %1, %1, %9\n v_mul_hi_u32_u24 %2, %2, %10\n v_mul_hi_u32_u24 %3, %3, %11\n"
    "v_mul_hi_u32_u24 %4, %4, %12\n v_mul_hi_u32_u24 %5, %5, %13\n v_mul_hi_u32_u24 %6, %6, %14\n v_mul_hi_u32_u24 %7, %7, %15\n"
    "v_mul_hi_u32_u24 %8, %8, %0\n v_mul_hi_u32_u24 %9, %9, %1\n v_mul_hi_u32_u24 %10, %10, %2\n v_mul_hi_u32_u24 %11, %11, %3\n"
    "v_mul_hi_u32_u24 %12, %12, %4\n v_mul_hi_u32_u24 %13, %13, %5\n v_mul_hi_u32_u24 %14, %14, %6\n v_mul_hi_u32_u24 %15, %15, %7\n")
KERNEL(k_alignbit_and,          // the shift_down pair of montx_dev + v_and
    "v_alignbit_b32 %0, %1, %0, 28\n v_lshrrev_b32 %1, 28, %1\n v_and_b32 %2, %2, %8\n v_alignbit_b32 %3, %4, %3, 28\n"
    "v_lshrrev_b32 %4, 28, %4\n v_and_b32 %5, %5, %9\n v_alignbit_b32 %6, %7, %6, 28\n v_lshrrev_b32 %7, 28, %7\n"
    "v_and_b32 %8, %8, %10\n v_alignbit_b32 %9, %10, %9, 28\n v_lshrrev_b32 %10, 28, %10\n v_and_b32 %11, %11, %12\n"
    "v_alignbit_b32 %12, %13, %12, 28\n v_lshrrev_b32 %13, 28, %13\n v_and_b32 %14, %14, %15\n v_xor_b32 %15, %15, %0\n")
KERNEL(k_pk_add_u16,
    "v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %9\n v_pk_add_u16 %2, %2, %10\n v_pk_add_u16 %3, %3, %11\n"
    "v_pk_add_u16 %4, %4, %12\n v_pk_add_u16 %5, %5, %13\n v_pk_add_u16 %6, %6, %14\n v_pk_add_u16 %7, %7, %15\n"
    "v_pk_add_u16 %8, %8, %0\n v_pk_add_u16 %9, %9, %1\n v_pk_add_u16 %10, %10, %2\n v_pk_add_u16 %11, %11, %3\n"
    "v_pk_add_u16 %12, %12, %4\n v_pk_add_u16 %13, %13, %5\n v_pk_add_u16 %14, %14, %6\n v_pk_add_u16 %15, %15, %7\n")


KERNEL(k_add_nop0,               // every VALU instruction followed by s_nop 0 (the padding of the hand-written carry chains)
    "v_add_u32 %0, %0, %8\n s_nop 0\n v_add_u32 %1, %1, %9\n s_nop 0\n v_add_u32 %2, %2, %10\n s_nop 0\n v_add_u32 %3, %3, %11\n s_nop 0\n"
    "v_add_u32 %4, %4, %12\n s_nop 0\n v_add_u32 %5, %5, %13\n s_nop 0\n v_add_u32 %6, %6, %14\n s_nop 0\n v_add_u32 %7, %7, %15\n s_nop 0\n"
    "v_add_u32 %8, %8, %0\n s_nop 0\n v_add_u32 %9, %9, %1\n s_nop 0\n v_add_u32 %10, %10, %2\n s_nop 0\n v_add_u32 %11, %11, %3\n s_nop 0\n"
    "v_add_u32 %12, %12, %4\n s_nop 0\n v_add_u32 %13, %13, %5\n s_nop 0\n v_add_u32 %14, %14, %6\n s_nop 0\n v_add_u32 %15, %15, %7\n s_nop 0\n")
KERNEL(k_add_nop1,
    "v_add_u32 %0, %0, %8\n s_nop 1\n v_add_u32 %1, %1, %9\n s_nop 1\n v_add_u32 %2, %2, %10\n s_nop 1\n v_add_u32 %3, %3, %11\n s_nop 1\n"
    "v_add_u32 %4, %4, %12\n s_nop 1\n v_add_u32 %5, %5, %13\n s_nop 1\n v_add_u32 %6, %6, %14\n s_nop 1\n v_add_u32 %7, %7, %15\n s_nop 1\n"
    "v_add_u32 %8, %8, %0\n s_nop 1\n v_add_u32 %9, %9, %1\n s_nop 1\n v_add_u32 %10, %10, %2\n s_nop 1\n v_add_u32 %11, %11, %3\n s_nop 1\n"
    "v_add_u32 %12, %12, %4\n s_nop 1\n v_add_u32 %13, %13, %5\n s_nop 1\n v_add_u32 %14, %14, %6\n s_nop 1\n v_add_u32 %15, %15, %7\n s_nop 1\n")
KERNEL(k_add_salu,               // every VALU instruction followed by a scalar ALU instruction
    "v_add_u32 %0, %0, %8\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %9\n s_add_u32 s21, s21, 1\n v_add_u32 %2, %2, %10\n s_add_u32 s22, s22, 1\n v_add_u32 %3, %3, %11\n s_add_u32 s23, s23, 1\n"
    "v_add_u32 %4, %4, %12\n s_add_u32 s20, s20, 1\n v_add_u32 %5, %5, %13\n s_add_u32 s21, s21, 1\n v_add_u32 %6, %6, %14\n s_add_u32 s22, s22, 1\n v_add_u32 %7, %7, %15\n s_add_u32 s23, s23, 1\n"
    "v_add_u32 %8, %8, %0\n s_add_u32 s20, s20, 1\n v_add_u32 %9, %9, %1\n s_add_u32 s21, s21, 1\n v_add_u32 %10, %10, %2\n s_add_u32 s22, s22, 1\n v_add_u32 %11, %11, %3\n s_add_u32 s23, s23, 1\n"
    "v_add_u32 %12, %12, %4\n s_add_u32 s20, s20, 1\n v_add_u32 %13, %13, %5\n s_add_u32 s21, s21, 1\n v_add_u32 %14, %14, %6\n s_add_u32 s22, s22, 1\n v_add_u32 %15, %15, %7\n s_add_u32 s23, s23, 1\n")

#define KERNEL64(name, ASMBLOCK)                                                       \
__global__ __launch_bounds__(256) void name(u32* out, u32 seed, int iters)              \
{                                                                                       \
    u64 a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    u32 b0 = (u32)a0 ^ 0x9e3779b9u, b1 = (u32)a1 ^ 0x7f4a7c15u, b2 = b0 + 1, b3 = b1 + 2, b4 = b0 + 3, b5 = b1 + 4, b6 = b0 + 5, b7 = b1 + 6; \
    for (int i = 0; i < iters; i++) {                                                   \
        asm volatile(ASMBLOCK                                                           \
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
              "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) \
            : : "vcc");                                                                 \
    }                                                                                   \
    out[blockIdx.x * 256 + threadIdx.x] = (u32)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7; \
}
KERNEL64(k_lshl_add_u64,          // 64-bit add in one instruction (no carry out)
    "v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %4\n"
    "v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 0, %6\n v_lshl_add_u64 %6, %6, 0, %7\n v_lshl_add_u64 %7, %7, 0, %0\n"
    "v_lshl_add_u64 %0, %0, 0, %2\n v_lshl_add_u64 %1, %1, 0, %3\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %5\n"
    "v_lshl_add_u64 %4, %4, 0, %6\n v_lshl_add_u64 %5, %5, 0, %7\n v_lshl_add_u64 %6, %6, 0, %0\n v_lshl_add_u64 %7, %7, 0, %1\n")
KERNEL64(k_mad_u64_u32,
    "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %10, %11, %1\n v_mad_u64_u32 %2, vcc, %12, %13, %2\n v_mad_u64_u32 %3, vcc, %14, %15, %3\n"
    "v_mad_u64_u32 %4, vcc, %9, %10, %4\n v_mad_u64_u32 %5, vcc, %11, %12, %5\n v_mad_u64_u32 %6, vcc, %13, %14, %6\n v_mad_u64_u32 %7, vcc, %15, %8, %7\n"
    "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %10, %11, %1\n v_mad_u64_u32 %2, vcc, %12, %13, %2\n v_mad_u64_u32 %3, vcc, %14, %15, %3\n"
    "v_mad_u64_u32 %4, vcc, %9, %10, %4\n v_mad_u64_u32 %5, vcc, %11, %12, %5\n v_mad_u64_u32 %6, vcc, %13, %14, %6\n v_mad_u64_u32 %7, vcc, %15, %8, %7\n")
KERNEL64(k_mad_u64_u32_carry,     // multiply-add whose carry-out feeds an add-with-carry (32-bit-limb style)
    "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_addc_co_u32 %10, vcc, 0, %10, vcc\n v_mad_u64_u32 %1, vcc, %10, %11, %1\n v_addc_co_u32 %12, vcc, 0, %12, vcc\n"
    "v_mad_u64_u32 %2, vcc, %12, %13, %2\n v_addc_co_u32 %14, vcc, 0, %14, vcc\n v_mad_u64_u32 %3, vcc, %14, %15, %3\n v_addc_co_u32 %8, vcc, 0, %8, vcc\n"
    "v_mad_u64_u32 %4, vcc, %9, %10, %4\n v_addc_co_u32 %11, vcc, 0, %11, vcc\n v_mad_u64_u32 %5, vcc, %11, %12, %5\n v_addc_co_u32 %13, vcc, 0, %13, vcc\n"
    "v_mad_u64_u32 %6, vcc, %13, %14, %6\n v_addc_co_u32 %15, vcc, 0, %15, vcc\n v_mad_u64_u32 %7, vcc, %15, %8, %7\n v_addc_co_u32 %9, vcc, 0, %9, vcc\n")

KERNEL64(k_mad_two_chains,        // two dependent chains, alternating (montx_dev::mul2 / macx2)
    "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %10, %11, %1\n v_mad_u64_u32 %0, vcc, %12, %13, %0\n v_mad_u64_u32 %1, vcc, %14, %15, %1\n"
    "v_mad_u64_u32 %0, vcc, %9, %10, %0\n v_mad_u64_u32 %1, vcc, %11, %12, %1\n v_mad_u64_u32 %0, vcc, %13, %14, %0\n v_mad_u64_u32 %1, vcc, %15, %8, %1\n"
    "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %10, %11, %1\n v_mad_u64_u32 %0, vcc, %12, %13, %0\n v_mad_u64_u32 %1, vcc, %14, %15, %1\n"
    "v_mad_u64_u32 %0, vcc, %9, %10, %0\n v_mad_u64_u32 %1, vcc, %11, %12, %1\n v_mad_u64_u32 %0, vcc, %13, %14, %0\n v_mad_u64_u32 %1, vcc, %15, %8, %1\n")
KERNEL64(k_mad_one_chain,         // one dependent chain (montx_dev::operator*)
    "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %0, vcc, %10, %11, %0\n v_mad_u64_u32 %0, vcc, %12, %13, %0\n v_mad_u64_u32 %0, vcc, %14, %15, %0\n"
    "v_mad_u64_u32 %0, vcc, %9, %10, %0\n v_mad_u64_u32 %0, vcc, %11, %12, %0\n v_mad_u64_u32 %0, vcc, %13, %14, %0\n v_mad_u64_u32 %0, vcc, %15, %8, %0\n"
    "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %0, vcc, %10, %11, %0\n v_mad_u64_u32 %0, vcc, %12, %13, %0\n v_mad_u64_u32 %0, vcc, %14, %15, %0\n"
    "v_mad_u64_u32 %0, vcc, %9, %10, %0\n v_mad_u64_u32 %0, vcc, %11, %12, %0\n v_mad_u64_u32 %0, vcc, %13, %14, %0\n v_mad_u64_u32 %0, vcc, %15, %8, %0\n")
KERNEL64(k_mad_four_chains,
    "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %10, %11, %1\n v_mad_u64_u32 %2, vcc, %12, %13, %2\n v_mad_u64_u32 %3, vcc, %14, %15, %3\n"
    "v_mad_u64_u32 %0, vcc, %9, %10, %0\n v_mad_u64_u32 %1, vcc, %11, %12, %1\n v_mad_u64_u32 %2, vcc, %13, %14, %2\n v_mad_u64_u32 %3, vcc, %15, %8, %3\n"
    "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %10, %11, %1\n v_mad_u64_u32 %2, vcc, %12, %13, %2\n v_mad_u64_u32 %3, vcc, %14, %15, %3\n"
    "v_mad_u64_u32 %0, vcc, %9, %10, %0\n v_mad_u64_u32 %1, vcc, %11, %12, %1\n v_mad_u64_u32 %2, vcc, %13, %14, %2\n v_mad_u64_u32 %3, vcc, %15, %8, %3\n")

typedef void (*kern_t)(u32*, u32, int);
struct entry { const char* name; kern_t k; };
static entry table[] = {
    {"v_add_u32", k_add_u32}, {"v_add_u32 + s_nop 0 (VALU only counted)", k_add_nop0}, {"v_add_u32 + s_nop 1 (VALU only counted)", k_add_nop1}, {"v_add_u32 + s_add_u32 (VALU only counted)", k_add_salu}, {"v_add_co + v_addc_co (vcc)", k_addco_vcc_pairs}, {"v_add_co + v_addc_co (sgpr pairs)", k_addco_sgpr_pairs},
    {"v_add_co_u32 (vcc, carry unused)", k_addco_only_vcc}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_add3_u32", k_add3_u32},
    {"v_cmp_lt_u32 + v_cndmask_b32", k_cmp_cndmask}, {"v_mad_u64_u32 (8 chains)", k_mad_u64_u32}, {"v_mad_u64_u32, 1 dependent chain", k_mad_one_chain}, {"v_mad_u64_u32, 2 dependent chains", k_mad_two_chains}, {"v_mad_u64_u32, 4 dependent chains", k_mad_four_chains}, {"v_mad_u64_u32 + v_addc_co (vcc)", k_mad_u64_u32_carry}, {"v_mul_lo_u32", k_mul_lo_u32},
    {"v_mul_u32_u24", k_mul_u32_u24}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24},
    {"alignbit/lshrrev/and mix", k_alignbit_and}, {"v_pk_add_u16", k_pk_add_u16},
};

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = 100000;
    u32* out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%d CUs, clock %d kHz (nominal 2.4 GHz used below)\n", cus, prop.clockRate);
    printf("%-36s", "16-instruction block, 8 chains");
    for (int w : {1, 2, 3, 4, 6, 8}) printf(" | %d w/SIMD", w);
    printf("   (wave-instr / clk / SIMD)\n");
    hipLaunchKernelGGL(k_add_u32, dim3(cus * 8), dim3(256), 0, 0, out, 1u, 4000000);      // ~1 s: clocks up
    hipDeviceSynchronize();
    for (auto& t : table) {
        printf("%-36s", t.name);
        for (int w : {1, 2, 3, 4, 6, 8}) {
            dim3 grid(cus * w), block(256);              // one 4-wave block per CU and per wave of occupancy
            hipLaunchKernelGGL(t.k, grid, block, 0, 0, out, 1u, 100);
            hipDeviceSynchronize();
            float ms = 1e9;
            for (int r = 0; r < 3; r++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(t.k, grid, block, 0, 0, out, 1u, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float m; hipEventElapsedTime(&m, e0, e1); if (m < ms) ms = m;
            }
            double instr_per_simd = (double)w * iters * 16;
            printf(" | %8.3f", instr_per_simd / (ms * 1e-3 * 2.4e9));
        }
        printf("\n");
    }
    return 0;
}
