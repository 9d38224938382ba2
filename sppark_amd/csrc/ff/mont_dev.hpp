// Device-side N-limb Montgomery field for gfx950 (CDNA4), 32-bit limbs.
//
// Replaces, for the MI355X path, the role of the reference's CUDA class
// ff/mont_t.cuh:33-1217 (PTX mad.lo.cc/madc.hi.cc chains) -- this is a fresh
// design around the one wide integer multiplier CDNA4 has:
//
//     v_mad_u64_u32  D64, carry = S0_32 * S1_32 + S2_64
//
// The product is accumulated column-wise (product scanning / FIPS Montgomery):
// one 64-bit accumulator pair plus a 32-bit overflow counter.  Each partial
// product costs exactly two VALU instructions,
//     v_mad_u64_u32  acc, vcc, a_i, b_j, acc        (quarter rate)
//     v_addc_co_u32  ovf, vcc, 0, ovf, vcc          (full rate)
// because v_mad_u64_u32 has a carry-out but no carry-in.  hipcc does not use
// that carry-out when given C (it emits v_lshl_add_u64 + v_cmp + v_cndmask per
// product), hence the inline asm for this one primitive; everything around it
// is plain C++ so that hipcc allocates registers and schedules.
//
// Wire format = the reference's: little-endian 32-bit limbs, Montgomery form
// with R = 2^(32*N) (ff/bls12-381.hpp:13-51, ff/alt_bn128.hpp:13-48).
// All values are kept canonical (< p) between operations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// SPPARK_HOST_EMULATION is defined ONLY by the test harness under tests/emu/,
// which compiles these same kernels' per-thread bodies for the host CPU so that
// index/recoding/flush logic can be exercised in the GPU-less build container.
// The shipped libraries are built without it: every function below is then
// __device__-only and there is no CPU path of any kind.
#ifdef SPPARK_HOST_EMULATION
# define SPPARK_DEVFN __host__ __device__ inline
#else
# define SPPARK_DEVFN __device__ __forceinline__
#endif

namespace sppark_amd {

typedef uint32_t u32;
typedef uint64_t u64;

// acc(64) + ovf(32) += a*b
SPPARK_DEVFN void mac96(u64& acc, u32& ovf, u32 a, u32 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(ovf) : "v"(a), "v"(b) : "vcc");
#else   // host emulation (tests only)
    u64 p = (u64)a * b, s = acc + p;
    ovf += s < p;
    acc = s;
#endif
}
// same with b a wave-uniform constant kept in an SGPR (modulus limbs)
SPPARK_DEVFN void mac96s(u64& acc, u32& ovf, u32 a, u32 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(ovf) : "v"(a), "s"(b) : "vcc");
#else
    mac96(acc, ovf, a, b);
#endif
}

template<class P> struct mont_dev {
    static constexpr int N = P::N;
    u32 v[N];

    // wire format == register format for the 32-bit-limb class
    SPPARK_DEVFN static mont_dev from_wire(const u32* w)
    {   mont_dev r; for (int i = 0; i < N; i++) r.v[i] = w[i]; return r;   }
    SPPARK_DEVFN void to_wire(u32* w) const
    {   for (int i = 0; i < N; i++) w[i] = v[i];   }

    SPPARK_DEVFN static mont_dev zero()
    {   mont_dev r; for (int i = 0; i < N; i++) r.v[i] = 0; return r;   }
    SPPARK_DEVFN static mont_dev one()
    {   mont_dev r; for (int i = 0; i < N; i++) r.v[i] = P::ONE[i]; return r;   }
    SPPARK_DEVFN static mont_dev modulus()
    {   mont_dev r; for (int i = 0; i < N; i++) r.v[i] = P::MOD[i]; return r;   }

    SPPARK_DEVFN bool is_zero() const
    {
        u32 acc = v[0];
        #pragma unroll
        for (int i = 1; i < N; i++) acc |= v[i];
        return acc == 0;
    }
    SPPARK_DEVFN bool equals(const mont_dev& b) const
    {
        u32 acc = v[0] ^ b.v[0];
        #pragma unroll
        for (int i = 1; i < N; i++) acc |= v[i] ^ b.v[i];
        return acc == 0;
    }

    // r = t - p if t >= p (top = carry word above the top limb)
    SPPARK_DEVFN static mont_dev final_sub(const u32 t[N], u32 top)
    {
        u32 u[N], bw = 0;
        #pragma unroll
        for (int i = 0; i < N; i++) u[i] = __builtin_subc(t[i], (u32)P::MOD[i], bw, &bw);
        bool ge = (top != 0) | (bw == 0);
        mont_dev r;
        #pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = ge ? u[i] : t[i];
        return r;
    }

    SPPARK_DEVFN friend mont_dev operator+(const mont_dev& a, const mont_dev& b)
    {
        u32 t[N], c = 0;
        #pragma unroll
        for (int i = 0; i < N; i++) t[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
        return final_sub(t, c);
    }
    SPPARK_DEVFN friend mont_dev operator-(const mont_dev& a, const mont_dev& b)
    {
        u32 t[N], bw = 0, c = 0;
        #pragma unroll
        for (int i = 0; i < N; i++) t[i] = __builtin_subc(a.v[i], b.v[i], bw, &bw);
        u32 mask = 0u - bw;                              // add p back on borrow
        mont_dev r;
        #pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = __builtin_addc(t[i], (u32)P::MOD[i] & mask, c, &c);
        return r;
    }
    SPPARK_DEVFN mont_dev dbl() const { return *this + *this; }
    SPPARK_DEVFN mont_dev neg() const
    {
        mont_dev z = zero();
        return is_zero() ? z : modulus_minus(*this);
    }
    SPPARK_DEVFN static mont_dev modulus_minus(const mont_dev& a)
    {
        mont_dev r; u32 bw = 0;
        #pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = __builtin_subc((u32)P::MOD[i], a.v[i], bw, &bw);
        return r;
    }
    SPPARK_DEVFN mont_dev cneg(bool flag) const
    {
        mont_dev n = neg(), r;
        #pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = flag ? n.v[i] : v[i];
        return r;
    }
    SPPARK_DEVFN static mont_dev select(bool pick_a, const mont_dev& a, const mont_dev& b)
    {
        mont_dev r;
        #pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = pick_a ? a.v[i] : b.v[i];
        return r;
    }

    // Montgomery product a*b/R mod p, column-wise.
    SPPARK_DEVFN friend mont_dev operator*(const mont_dev& a, const mont_dev& b)
    {
        u32 m[N], t[N];
        u64 acc = 0; u32 ovf = 0;
        #pragma unroll
        for (int k = 0; k < N; k++) {
            #pragma unroll
            for (int i = 0; i <= k; i++) mac96(acc, ovf, a.v[i], b.v[k - i]);
            #pragma unroll
            for (int i = 0; i < k; i++) mac96s(acc, ovf, m[i], (u32)P::MOD[k - i]);
            m[k] = (u32)acc * (u32)P::M0;
            mac96s(acc, ovf, m[k], (u32)P::MOD[0]);       // low word becomes 0
            acc = (acc >> 32) | ((u64)ovf << 32); ovf = 0;
        }
        #pragma unroll
        for (int k = N; k < 2 * N; k++) {
            #pragma unroll
            for (int i = k - N + 1; i < N; i++) {
                mac96(acc, ovf, a.v[i], b.v[k - i]);
                mac96s(acc, ovf, m[i], (u32)P::MOD[k - i]);
            }
            t[k - N] = (u32)acc;
            acc = (acc >> 32) | ((u64)ovf << 32); ovf = 0;
        }
        return final_sub(t, (u32)acc);
    }

    // Montgomery square: cross products once, doubled, plus the diagonal.
    SPPARK_DEVFN mont_dev sqr() const
    {
        // 2N-limb square first (column-wise), then N reduction columns.
        u32 w[2 * N];
        {
            u64 acc = 0; u32 ovf = 0;
            #pragma unroll
            for (int k = 0; k < 2 * N - 1; k++) {
                // cross terms i<j, i+j=k
                u64 c = 0; u32 co = 0;
                #pragma unroll
                for (int i = 0; i < N; i++) {
                    int j = k - i;
                    if (j > i && j < N) mac96(c, co, v[i], v[j]);
                }
                // acc += 2*c (+ diagonal)
                u32 c0 = (u32)c, c1 = (u32)(c >> 32);
                u32 d0 = c0 << 1, d1 = (c1 << 1) | (c0 >> 31), d2 = (co << 1) | (c1 >> 31);
                u32 a0 = (u32)acc, a1 = (u32)(acc >> 32), cy = 0;
                a0 = __builtin_addc(a0, d0, cy, &cy);
                a1 = __builtin_addc(a1, d1, cy, &cy);
                ovf = ovf + d2 + cy;
                acc = ((u64)a1 << 32) | a0;
                if ((k & 1) == 0) mac96(acc, ovf, v[k / 2], v[k / 2]);
                w[k] = (u32)acc;
                acc = (acc >> 32) | ((u64)ovf << 32); ovf = 0;
            }
            w[2 * N - 1] = (u32)acc;
        }
        return reduce_wide(w);
    }

    // Montgomery reduction of a 2N-limb value (< p*R): returns w/R mod p.
    SPPARK_DEVFN static mont_dev reduce_wide(const u32 w[2 * N])
    {
        u32 m[N], t[N];
        u64 acc = 0; u32 ovf = 0;
        #pragma unroll
        for (int k = 0; k < N; k++) {
            u32 a0 = (u32)acc, a1 = (u32)(acc >> 32), cy = 0;
            a0 = __builtin_addc(a0, w[k], cy, &cy);
            a1 = __builtin_addc(a1, 0u, cy, &cy);
            ovf += cy;
            acc = ((u64)a1 << 32) | a0;
            #pragma unroll
            for (int i = 0; i < k; i++) mac96s(acc, ovf, m[i], (u32)P::MOD[k - i]);
            m[k] = (u32)acc * (u32)P::M0;
            mac96s(acc, ovf, m[k], (u32)P::MOD[0]);
            acc = (acc >> 32) | ((u64)ovf << 32); ovf = 0;
        }
        #pragma unroll
        for (int k = N; k < 2 * N; k++) {
            u32 a0 = (u32)acc, a1 = (u32)(acc >> 32), cy = 0;
            a0 = __builtin_addc(a0, w[k], cy, &cy);
            a1 = __builtin_addc(a1, 0u, cy, &cy);
            ovf += cy;
            acc = ((u64)a1 << 32) | a0;
            #pragma unroll
            for (int i = k - N + 1; i < N; i++) mac96s(acc, ovf, m[i], (u32)P::MOD[k - i]);
            t[k - N] = (u32)acc;
            acc = (acc >> 32) | ((u64)ovf << 32); ovf = 0;
        }
        return final_sub(t, (u32)acc);
    }

    // out of Montgomery form: a * 1 / R
    SPPARK_DEVFN mont_dev from() const
    {
        u32 w[2 * N];
        #pragma unroll
        for (int i = 0; i < N; i++) { w[i] = v[i]; w[N + i] = 0; }
        return reduce_wide(w);
    }
    // into Montgomery form: a * RR / R
    SPPARK_DEVFN mont_dev to() const
    {
        mont_dev rr;
        #pragma unroll
        for (int i = 0; i < N; i++) rr.v[i] = P::RR[i];
        return *this * rr;
    }
};

} // namespace sppark_amd
