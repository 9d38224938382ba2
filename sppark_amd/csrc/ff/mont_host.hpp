// Host-side Montgomery field (64-bit limbs, CIOS) used by the MSM driver for
// the O(windows) tail of an MSM -- Horner over the per-window sums -- and by the
// multi-GPU combine step.  The reference does this part on the host as well
// (msm/pippenger.cuh:627-727 collect/integrate_row) with blst's field classes
// (un-vendored); this is the product's own replacement for that dependency.
// It is NOT a CPU fallback for the device kernels: nothing here can evaluate an
// MSM or an NTT.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#if defined(__x86_64__) && (defined(__clang__) || defined(__GNUC__)) && !defined(__HIP_DEVICE_COMPILE__) && !defined(SPPARK_HOST_NO_MULX)
# include "mont_host_x86.hpp"
# define SPPARK_HOST_MULX 1
#endif

namespace sppark_amd {

#ifdef SPPARK_HOST_MULX
// BMI2 (mulx) and ADX (adcx / adox) on this host?  (0: the adc-chain C product below)
static inline bool host_has_mulx_adx()
{
    static const bool have = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx");
    return have;
}
#endif

template<class P> struct mont_host {
    static constexpr int N = P::N64;
    typedef unsigned __int128 u128;
    uint64_t v[N];

    static mont_host zero() { mont_host r; memset(r.v, 0, sizeof(r.v)); return r; }
    static mont_host one()  { mont_host r; for (int i = 0; i < N; i++) r.v[i] = P::ONE64[i]; return r; }
    bool is_zero() const { uint64_t a = 0; for (int i = 0; i < N; i++) a |= v[i]; return a == 0; }
    friend bool operator==(const mont_host& a, const mont_host& b)
    {   uint64_t x = 0; for (int i = 0; i < N; i++) x |= a.v[i] ^ b.v[i]; return x == 0;   }

#if defined(__has_builtin)
# if __has_builtin(__builtin_addcll) && __has_builtin(__builtin_subcll) && defined(__x86_64__)
#  define SPPARK_HOST_ADC 1
# endif
#endif
#ifdef SPPARK_HOST_ADC
    typedef unsigned long long ull;
    // r = t - MOD if (top : t) >= MOD else t
    static inline void cond_sub(uint64_t r[N], const uint64_t t[N], uint64_t top)
    {
        ull u[N], bw = 0;
        for (int i = 0; i < N; i++) u[i] = __builtin_subcll(t[i], P::MOD64[i], bw, &bw);
        const bool ge = top | !bw;
        for (int i = 0; i < N; i++) r[i] = ge ? u[i] : t[i];
    }
    friend mont_host operator+(const mont_host& a, const mont_host& b)
    {
        uint64_t t[N]; ull c = 0; mont_host r;
        for (int i = 0; i < N; i++) t[i] = __builtin_addcll(a.v[i], b.v[i], c, &c);
        cond_sub(r.v, t, c);
        return r;
    }
    friend mont_host operator-(const mont_host& a, const mont_host& b)
    {
        mont_host r; ull bw = 0, c = 0;
        for (int i = 0; i < N; i++) r.v[i] = __builtin_subcll(a.v[i], b.v[i], bw, &bw);
        const uint64_t mask = 0 - (uint64_t)bw;
        for (int i = 0; i < N; i++) r.v[i] = __builtin_addcll(r.v[i], P::MOD64[i] & mask, c, &c);
        return r;
    }
#else
    static void cond_sub(uint64_t r[N], const uint64_t t[N], uint64_t top)
    {
        uint64_t u[N], bw = 0;
        for (int i = 0; i < N; i++) {
            u128 d = (u128)t[i] - P::MOD64[i] - bw;
            u[i] = (uint64_t)d; bw = (uint64_t)(d >> 127);
        }
        bool ge = top | !bw;
        for (int i = 0; i < N; i++) r[i] = ge ? u[i] : t[i];
    }
    friend mont_host operator+(const mont_host& a, const mont_host& b)
    {
        uint64_t t[N], c = 0; mont_host r;
        for (int i = 0; i < N; i++) { u128 s = (u128)a.v[i] + b.v[i] + c; t[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        cond_sub(r.v, t, c);
        return r;
    }
    friend mont_host operator-(const mont_host& a, const mont_host& b)
    {
        mont_host r; uint64_t bw = 0;
        for (int i = 0; i < N; i++) { u128 d = (u128)a.v[i] - b.v[i] - bw; r.v[i] = (uint64_t)d; bw = (uint64_t)(d >> 127); }
        uint64_t mask = 0 - bw, c = 0;
        for (int i = 0; i < N; i++) { u128 s = (u128)r.v[i] + (P::MOD64[i] & mask) + c; r.v[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        return r;
    }
#endif
#ifdef SPPARK_HOST_ADC
    // Every MSM ends in ~255 host doublings (Horner over the window sums): at 2^10 .. 2^16 points that is a tenth of the
    // call, so the product is written for the host's adc chains -- each row a[.] * b_i as N independent 64x64 products,
    // their low and high halves added in two carry chains.
    // t[0..N] += x[0..N-1] * y  (t[N+1] takes the carry out)
    static inline void mac_row(ull t[N + 2], const uint64_t x[N], uint64_t y)
    {
        ull lo[N], hi[N], c = 0, c2 = 0;
        for (int j = 0; j < N; j++) { u128 p = (u128)x[j] * y; lo[j] = (ull)p; hi[j] = (ull)(p >> 64); }
        for (int j = 0; j < N; j++) t[j] = __builtin_addcll(t[j], lo[j], c, &c);
        t[N] = __builtin_addcll(t[N], 0, c, &c); t[N + 1] += c;
        for (int j = 0; j < N; j++) t[j + 1] = __builtin_addcll(t[j + 1], hi[j], c2, &c2);
        t[N + 1] += c2;
    }
    // one Montgomery step: t = (t + m * MOD) / 2^64 with m = t[0] * M0
    static inline void red_row(ull t[N + 2])
    {
        const uint64_t m = t[0] * P::M0_64;
        ull lo[N], hi[N], c = 0, c2 = 0;
        for (int j = 0; j < N; j++) { u128 p = (u128)m * P::MOD64[j]; lo[j] = (ull)p; hi[j] = (ull)(p >> 64); }
        for (int j = 0; j < N; j++) t[j] = __builtin_addcll(t[j], lo[j], c, &c);
        t[N] = __builtin_addcll(t[N], 0, c, &c); t[N + 1] += c;
        for (int j = 0; j < N; j++) t[j] = __builtin_addcll(t[j + 1], hi[j], c2, &c2);
        t[N] = t[N + 1] + c2; t[N + 1] = 0;
    }
    friend mont_host operator*(const mont_host& a, const mont_host& b)
    {
#ifdef SPPARK_HOST_MULX
        // 4 / 6 limbs on hosts with BMI2 + ADX: the generated mulx / adcx / adox product (EPYC 9575F: 6 limbs 27 -> 19 ns,
        // profiles/r06_host_field.log); it needs the modulus below 2^(64 N - 1)
        if constexpr ((N == 4 || N == 6) && (P::MOD64[N - 1] >> 63) == 0) {
            if (host_has_mulx_adx()) {
                uint64_t u[N];
                if constexpr (N == 4) mont_mul_x86_4(u, a.v, b.v, P::MOD64, P::M0_64);
                else                  mont_mul_x86_6(u, a.v, b.v, P::MOD64, P::M0_64);
                mont_host r; cond_sub(r.v, u, 0);
                return r;
            }
        }
#endif
        ull t[N + 2] = {0};
        _Pragma("unroll")
        for (int i = 0; i < N; i++) { mac_row(t, a.v, b.v[i]); red_row(t); }
        uint64_t u[N + 1];
        for (int i = 0; i <= N; i++) u[i] = t[i];
        mont_host r; cond_sub(r.v, u, u[N]);
        return r;
    }
    // (a dedicated square -- off-diagonal products once, doubled -- measured 29.3 ns against this product's 27.1 ns on the
    //  EPYC 9575F host, tools/host_field_bench.cpp: the shifts and the extra chain cost more than 15 products there)
    mont_host sqr() const { return *this * *this; }
#else
    // coarsely integrated operand scanning
    friend mont_host operator*(const mont_host& a, const mont_host& b)
    {
        uint64_t t[N + 2] = {0};
        for (int i = 0; i < N; i++) {
            uint64_t c = 0;
            for (int j = 0; j < N; j++) { u128 s = (u128)a.v[j] * b.v[i] + t[j] + c; t[j] = (uint64_t)s; c = (uint64_t)(s >> 64); }
            u128 s = (u128)t[N] + c; t[N] = (uint64_t)s; t[N + 1] = (uint64_t)(s >> 64);
            uint64_t m = t[0] * P::M0_64;
            c = (uint64_t)(((u128)m * P::MOD64[0] + t[0]) >> 64);
            for (int j = 1; j < N; j++) { u128 q = (u128)m * P::MOD64[j] + t[j] + c; t[j - 1] = (uint64_t)q; c = (uint64_t)(q >> 64); }
            s = (u128)t[N] + c; t[N - 1] = (uint64_t)s; t[N] = t[N + 1] + (uint64_t)(s >> 64);
        }
        mont_host r; cond_sub(r.v, t, t[N]);
        return r;
    }
    mont_host sqr() const { return *this * *this; }
#endif
    mont_host dbl() const { return *this + *this; }
    mont_host neg() const { return is_zero() ? *this : zero() - *this; }

    mont_host inverse() const               // a^(p-2), 1/0 = 0
    {
        uint64_t e[N], bw = 2;
        for (int i = 0; i < N; i++) { u128 d = (u128)P::MOD64[i] - bw; e[i] = (uint64_t)d; bw = (uint64_t)(d >> 127); }
        mont_host r = one(), b = *this;
        for (int i = 0; i < 64 * N; i++) { if ((e[i / 64] >> (i % 64)) & 1) r = r * b; b = b.sqr(); }
        return r;
    }
};

} // namespace sppark_amd
