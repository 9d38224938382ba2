// Device test hooks and micro-benchmarks (exported as sppark_devtest_*).
// They exist so that tests/ can check the DEVICE field / point arithmetic
// element-by-element against the oracle, and so that DESIGN.md's instruction
// costs are measured on the box rather than assumed.  Not part of the
// reference's surface.
#include "../msm/curve_select.hpp"
#include "../util/runtime.hpp"
#include <vector>

using namespace sppark_amd;

#define SPPARK_FFI extern "C" __attribute__((visibility("default")))

template<class F>
__global__ void k_field_op(u32* out, const u32* a, const u32* b, unsigned n, int op)
{
    constexpr int N = F::N;
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F x = F::from_wire(a + (size_t)i * N), y = F::from_wire(b + (size_t)i * N), r;
    switch (op) {
        case 0: r = x + y; break;
        case 1: r = x - y; break;
        case 2: r = x * y; break;
        case 3: r = x.sqr(); break;
        case 4: r = x.neg(); break;
        case 5: r = x.from(); break;
        case 6: r = x.to(); break;
        default: r = x.dbl(); break;
    }
    r.to_wire(out + (size_t)i * N);
}

// op 0: a += b (xyzz)   1: a += affine(b)   2: a -= affine(b)   3: a = 2a
__global__ void k_xyzz_op(wire_bucket_m* out, const wire_bucket_m* a, const unsigned char* b, unsigned n, int op)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    wire_bucket_d p = wire_bucket_d::load(&a[i]);
    if (op == 0) {
        p.add(wire_bucket_d::load(reinterpret_cast<const wire_bucket_m*>(b) + i));
    } else if (op == 3) {
        p.dbl();
    } else {
        affine_dev<fp_d> q = load_affine<fp_d, false>(b, i, 8 * fp_d::N);
        p.madd(q, op == 2);
    }
    p.store(&out[i]);
}

template<class Fn> static RustError guarded(Fn&& fn)
{
    try { fn(); return rust_ok(); }
    catch (const hip_error& e) { (void)hipGetLastError(); return rust_err(e.code(), e.what()); }
    catch (const std::exception& e) { return rust_err(-1, e.what()); }
}

template<class F>
static void run_field_op(int op, void* out, const void* a, const void* b, size_t n)
{
    (void)select_gpu(-1);
    size_t bytes = n * sizeof(F);
    u32 *d_a, *d_b, *d_o;
    HIP_OK(hipMalloc((void**)&d_a, bytes)); HIP_OK(hipMalloc((void**)&d_b, bytes)); HIP_OK(hipMalloc((void**)&d_o, bytes));
    HIP_OK(hipMemcpy(d_a, a, bytes, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_b, b ? b : a, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_field_op<F>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, d_o, d_a, d_b, (unsigned)n, op);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpy(out, d_o, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_o);
}

// field 0 = base field fp, 1 = scalar field fr
SPPARK_FFI RustError sppark_devtest_field_op(int field, int op, void* out, const void* a, const void* b, size_t n)
{
    return guarded([&] {
        if (field == 0) run_field_op<fp_d>(op, out, a, b, n);
        else            run_field_op<fr_d>(op, out, a, b, n);
    });
}

SPPARK_FFI RustError sppark_devtest_xyzz_op(int op, void* out, const void* a, const void* b, size_t n)
{
    return guarded([&] {
        (void)select_gpu(-1);
        size_t ab = n * sizeof(wire_bucket_m), bb = n * (op == 0 ? sizeof(wire_bucket_m) : 8 * fp_d::N);
        wire_bucket_m *d_a, *d_o; unsigned char* d_b;
        HIP_OK(hipMalloc((void**)&d_a, ab)); HIP_OK(hipMalloc((void**)&d_o, ab)); HIP_OK(hipMalloc((void**)&d_b, bb ? bb : 16));
        HIP_OK(hipMemcpy(d_a, a, ab, hipMemcpyHostToDevice));
        if (op != 3) HIP_OK(hipMemcpy(d_b, b, bb, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_xyzz_op, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, d_o, d_a, d_b, (unsigned)n, op);
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpy(out, d_o, ab, hipMemcpyDeviceToHost));
        (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_o);
    });
}

// ---------------------------------------------------------------------------
// The bucket pipeline's own field (ff/montx_dev.hpp on BLS12-381): element-wise operations on
// internal limbs, and point operations computed in the loosely-reduced form but fed and read
// back in the reference's wire form, so that they can be compared bit for bit with the
// 32-bit-limb class and with the oracle.
// ---------------------------------------------------------------------------
template<class F>
__global__ void k_bucket_field_op(u32* out, const u32* a, const u32* b, unsigned n, int op)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if constexpr (field_is_internal<F>::value) {
        constexpr int NL = F::NL;
        F x = F::from_wire(a + (size_t)i * NL), y = F::from_wire(b + (size_t)i * NL), r = F::zero();
        switch (op) {
            case 0: r = x * y; break;                               // x may be fat (limbs < 2^31), y normalised
            case 1: r = x.sqr(); break;                             // x normalised
            case 2: r = F::template sub<3>(x, y).norm(); break;     // y normalised, < 2p
            case 3: r = (x + y).norm(); break;
            case 4: r.l[0] = x.template is_zero_mod<13>(); break;   // x normalised, < 13p
            case 5: r = F::from_std(a + (size_t)i * NL); break;     // first NW words: a wire-form element
            default: { u32 w[NL] = {}; x.to_std(w); for (int j = 0; j < NL; j++) r.l[j] = w[j]; } break;
        }
        r.to_wire(out + (size_t)i * NL);
    }
}

// op 0: a += b (xyzz)   1: a += affine(b)   2: a -= affine(b)   3: a = 2a ; all in wire form
template<class F>
__global__ void k_bucket_xyzz_op(wire_bucket_m* out, const wire_bucket_m* a, const unsigned char* b, unsigned n, int op)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if constexpr (field_is_internal<F>::value) {
        constexpr int NW = fp_d::N;
        auto to_internal = [](const wire_bucket_m* src) {
            xyzz_dev<F> r;
            bool inf = true;
            for (int k = 2 * NW; k < 4 * NW; k++) inf &= src->w[k] == 0;
            if (inf) { r.set_inf(); return r; }
            r.X = F::from_std(src->w); r.Y = F::from_std(src->w + NW);
            r.ZZZ = F::from_std(src->w + 2 * NW); r.ZZ = F::from_std(src->w + 3 * NW);
            return r;
        };
        xyzz_dev<F> p = to_internal(&a[i]);
        if (op == 0) p.add(to_internal(reinterpret_cast<const wire_bucket_m*>(b) + i));
        else if (op == 3) p.dbl();
        else {
            affine_dev<fp_d> qs = load_affine<fp_d, false>(b, i, 8 * NW);
            u32 wx[NW], wy[NW];
            qs.X.to_wire(wx); qs.Y.to_wire(wy);
            affine_dev<F> q; q.X = F::from_std(wx); q.Y = F::from_std(wy); q.inf = qs.inf;
            p.madd(q, op == 2);
        }
        p.store_std(&out[i]);
    }
}

// element size: NL words for ops on internal limbs (see k_bucket_field_op)
SPPARK_FFI RustError sppark_devtest_bucket_field_op(int op, void* out, const void* a, const void* b, size_t n)
{
    return guarded([&] {
        if (!field_is_internal<msm_fp_d>::value) HIP_OK(hipErrorNotSupported);
        (void)select_gpu(-1);
        size_t bytes = n * msm_fp_d::N * 4;
        u32 *d_a, *d_b, *d_o;
        HIP_OK(hipMalloc((void**)&d_a, bytes)); HIP_OK(hipMalloc((void**)&d_b, bytes)); HIP_OK(hipMalloc((void**)&d_o, bytes));
        HIP_OK(hipMemcpy(d_a, a, bytes, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_b, b ? b : a, bytes, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_bucket_field_op<msm_fp_d>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, d_o, d_a, d_b, (unsigned)n, op);
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpy(out, d_o, bytes, hipMemcpyDeviceToHost));
        (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_o);
    });
}
SPPARK_FFI int sppark_devtest_bucket_field_limbs(void)
{   return field_is_internal<msm_fp_d>::value ? (int)msm_fp_d::N : 0;   }

SPPARK_FFI RustError sppark_devtest_bucket_xyzz_op(int op, void* out, const void* a, const void* b, size_t n)
{
    return guarded([&] {
        if (!field_is_internal<msm_fp_d>::value) HIP_OK(hipErrorNotSupported);
        (void)select_gpu(-1);
        size_t ab = n * sizeof(wire_bucket_m), bb = n * (op == 0 ? sizeof(wire_bucket_m) : 8 * fp_d::N);
        wire_bucket_m *d_a, *d_o; unsigned char* d_b;
        HIP_OK(hipMalloc((void**)&d_a, ab)); HIP_OK(hipMalloc((void**)&d_o, ab)); HIP_OK(hipMalloc((void**)&d_b, bb ? bb : 16));
        HIP_OK(hipMemcpy(d_a, a, ab, hipMemcpyHostToDevice));
        if (op != 3) HIP_OK(hipMemcpy(d_b, b, bb, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_bucket_xyzz_op<msm_fp_d>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, d_o, d_a, d_b, (unsigned)n, op);
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpy(out, d_o, ab, hipMemcpyDeviceToHost));
        (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_o);
    });
}

// ---------------------------------------------------------------------------
// micro-benchmarks: every wave runs |iters| iterations of a fixed instruction
// block and reports its own s_memtime delta; the host also times the launch.
// ---------------------------------------------------------------------------
#define UB_REP8(x) x x x x x x x x

// WHICH is a template parameter so that the timed loop contains nothing but the
// instruction block (a runtime switch inside the loop dominated the first version).
template<int WHICH>
__global__ void k_ub(int iters, u64* clocks, u32* sink)
{
    u32 a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 977u;
    u64 acc0 = a, acc1 = b, acc2 = a ^ b, acc3 = a + b, acc4 = 1, acc5 = 2, acc6 = 3, acc7 = 4;
    u32 c0 = 0, c1 = 0, c2 = 1, c3 = 2;
    double f0 = a, f1 = b, f2 = 1.0000001, f3 = 3, f4 = 5, f5 = 7, f6 = 9, f7 = 11;
    u64 t0 = __builtin_readcyclecounter();
    #pragma unroll 1
    for (int i = 0; i < iters; i++) {
        if (WHICH == 0)         // 8 independent v_mad_u64_u32, each with its own carry SGPR pair
            asm volatile(
                "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n\tv_mad_u64_u32 %1, s[22:23], %8, %9, %1\n\t"
                "v_mad_u64_u32 %2, s[24:25], %8, %9, %2\n\tv_mad_u64_u32 %3, s[26:27], %8, %9, %3\n\t"
                "v_mad_u64_u32 %4, s[20:21], %8, %9, %4\n\tv_mad_u64_u32 %5, s[22:23], %8, %9, %5\n\t"
                "v_mad_u64_u32 %6, s[24:25], %8, %9, %6\n\tv_mad_u64_u32 %7, s[26:27], %8, %9, %7"
                : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(acc4), "+v"(acc5), "+v"(acc6), "+v"(acc7)
                : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        else if (WHICH == 1)    // 8 dependent v_mad_u64_u32
            asm volatile(UB_REP8("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t") : "+v"(acc0) : "v"(a), "v"(b) : "vcc");
        else if (WHICH == 2)    // 8 dependent mad + addc pairs (the mac96 primitive)
            asm volatile(UB_REP8("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t")
                         : "+v"(acc0), "+v"(c0) : "v"(a), "v"(b) : "vcc");
        else if (WHICH == 3)    // 4 independent mad + addc chains x 2
            asm volatile(
                "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
                "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
                "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"
                "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_addc_co_u32 %7, vcc, 0, %7, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
                "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
                "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"
                "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_addc_co_u32 %7, vcc, 0, %7, vcc"
                : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                : "v"(a), "v"(b) : "vcc");
        else if (WHICH == 4)    // 8 dependent v_mul_lo_u32
            asm volatile(UB_REP8("v_mul_lo_u32 %0, %0, %1\n\t") : "+v"(c0) : "v"(a));
        else if (WHICH == 5)    // 8 independent v_mul_lo_u32 (4 chains x 2)
            asm volatile(
                "v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4\n\t"
                "v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a));
        else if (WHICH == 6)    // 8 independent v_mad_u32_u24 (4 chains x 2)
            asm volatile(
                "v_mad_u32_u24 %0, %0, %4, %0\n\tv_mad_u32_u24 %1, %1, %4, %1\n\tv_mad_u32_u24 %2, %2, %4, %2\n\tv_mad_u32_u24 %3, %3, %4, %3\n\t"
                "v_mad_u32_u24 %0, %0, %4, %0\n\tv_mad_u32_u24 %1, %1, %4, %1\n\tv_mad_u32_u24 %2, %2, %4, %2\n\tv_mad_u32_u24 %3, %3, %4, %3"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a));
        else if (WHICH == 7)    // 8 independent v_add_u32 (4 chains x 2)
            asm volatile(
                "v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t"
                "v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a));
        else if (WHICH == 8)    // 8 dependent v_add_u32
            asm volatile(UB_REP8("v_add_u32 %0, %0, %1\n\t") : "+v"(c0) : "v"(a));
        else if (WHICH == 9)    // 8 independent v_fma_f64
            asm volatile(
                "v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.0000001), "v"(0.5));
        else if (WHICH == 10)   // 8 dependent v_fma_f64
            asm volatile(UB_REP8("v_fma_f64 %0, %0, %1, %2\n\t") : "+v"(f0) : "v"(1.0000001), "v"(0.5));
        else if (WHICH == 11)   // 8 independent v_lshl_add_u64
            asm volatile(
                "v_lshl_add_u64 %0, %0, 0, %4\n\tv_lshl_add_u64 %1, %1, 0, %4\n\tv_lshl_add_u64 %2, %2, 0, %4\n\tv_lshl_add_u64 %3, %3, 0, %4\n\t"
                "v_lshl_add_u64 %0, %0, 0, %4\n\tv_lshl_add_u64 %1, %1, 0, %4\n\tv_lshl_add_u64 %2, %2, 0, %4\n\tv_lshl_add_u64 %3, %3, 0, %4"
                : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(acc4));
        else if (WHICH == 12)   // 8 independent add_co+addc pairs (64-bit adds, 4 chains x 2)
            asm volatile(
                "v_add_co_u32 %0, vcc, %0, %4\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_add_co_u32 %2, vcc, %2, %4\n\tv_addc_co_u32 %3, vcc, %3, %4, vcc\n\t"
                "v_add_co_u32 %0, vcc, %0, %4\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_add_co_u32 %2, vcc, %2, %4\n\tv_addc_co_u32 %3, vcc, %3, %4, vcc"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a) : "vcc");
        else if (WHICH == 13)   // 8 independent v_mul_hi_u32
            asm volatile(
                "v_mul_hi_u32 %0, %0, %4\n\tv_mul_hi_u32 %1, %1, %4\n\tv_mul_hi_u32 %2, %2, %4\n\tv_mul_hi_u32 %3, %3, %4\n\t"
                "v_mul_hi_u32 %0, %0, %4\n\tv_mul_hi_u32 %1, %1, %4\n\tv_mul_hi_u32 %2, %2, %4\n\tv_mul_hi_u32 %3, %3, %4"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a));
    }
    u64 t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) clocks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    u64 s = acc0 ^ acc1 ^ acc2 ^ acc3 ^ acc4 ^ acc5 ^ acc6 ^ acc7 ^ c0 ^ c1 ^ c2 ^ c3 ^ (u64)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
    if (s == 0x123456789abcdefULL) sink[0] = (u32)s;
}

typedef void (*ub_fn)(int, u64*, u32*);
static ub_fn ub_table[] = { k_ub<0>, k_ub<1>, k_ub<2>, k_ub<3>, k_ub<4>, k_ub<5>, k_ub<6>, k_ub<7>, k_ub<8>,
                            k_ub<9>, k_ub<10>, k_ub<11>, k_ub<12>, k_ub<13> };

// returns elapsed ms; *cycles_per_iter = mean cycle-counter delta per iteration over waves
SPPARK_FFI RustError sppark_devtest_ubench(int which, int iters, unsigned blocks, unsigned threads,
                                           float* ms, double* cycles_per_iter)
{
    return guarded([&] {
        (void)select_gpu(-1);
        if (which < 0 || which >= (int)(sizeof(ub_table) / sizeof(ub_table[0]))) HIP_OK(hipErrorInvalidValue);
        size_t nw = (size_t)blocks * threads / 64;
        u64* d_clk; u32* d_sink;
        HIP_OK(hipMalloc((void**)&d_clk, nw * 8)); HIP_OK(hipMalloc((void**)&d_sink, 64));
        hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
        hipLaunchKernelGGL(ub_table[which], dim3(blocks), dim3(threads), 0, 0, 16, d_clk, d_sink);   // warm-up
        HIP_OK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(ub_table[which], dim3(blocks), dim3(threads), 0, 0, iters, d_clk, d_sink);
        HIP_OK(hipEventRecord(e1, 0));
        HIP_OK(hipEventSynchronize(e1));
        HIP_OK(hipEventElapsedTime(ms, e0, e1));
        std::vector<u64> clk(nw);
        HIP_OK(hipMemcpy(clk.data(), d_clk, nw * 8, hipMemcpyDeviceToHost));
        double sum = 0; for (auto c : clk) sum += (double)c;
        *cycles_per_iter = sum / nw / iters;
        (void)hipFree(d_clk); (void)hipFree(d_sink); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    });
}

// field-level throughput: x = x*y (op 0), x = x.sqr() (1), x = x+y (2), xyzz madd (3), xyzz add (4)
__global__ void k_fieldbench(int op, int iters, u32* io)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    u32 wx[fp_d::N], wy[fp_d::N];
    for (int k = 0; k < fp_d::N; k++) { wx[k] = io[k] ^ (i * 2654435761u >> 3); wy[k] = io[k + fp_d::N] + i; }
    wx[fp_d::N - 1] &= 0x0fffffff; wy[fp_d::N - 1] &= 0x0fffffff;
    fp_d x = fp_d::from_wire(wx), y = fp_d::from_wire(wy);
    if (op <= 2) {
        for (int it = 0; it < iters; it++) {
            if (op == 0) x = x * y; else if (op == 1) x = x.sqr(); else x = x + y;
        }
        if (x.is_zero()) io[0] = 1;
    } else {
        wire_bucket_d p; p.X = x; p.Y = y; p.ZZ = y; p.ZZZ = x;
        affine_dev<fp_d> q; q.X = y; q.Y = x; q.inf = false;
        wire_bucket_d r = p; r.X = y;
        for (int it = 0; it < iters; it++) {
            if (op == 3) p.madd(q, it & 1); else p.add(r);
        }
        if (p.X.is_zero()) io[0] = 1;
    }
}

SPPARK_FFI RustError sppark_devtest_fieldbench(int op, int iters, unsigned blocks, unsigned threads, float* ms)
{
    return guarded([&] {
        (void)select_gpu(-1);
        u32* d_io; HIP_OK(hipMalloc((void**)&d_io, 4096)); HIP_OK(hipMemset(d_io, 0x5a, 4096));
        hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_fieldbench, dim3(blocks), dim3(threads), 0, 0, op, 2, d_io);
        HIP_OK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_fieldbench, dim3(blocks), dim3(threads), 0, 0, op, iters, d_io);
        HIP_OK(hipEventRecord(e1, 0));
        HIP_OK(hipEventSynchronize(e1));
        HIP_OK(hipEventElapsedTime(ms, e0, e1));
        (void)hipFree(d_io); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    });
}
