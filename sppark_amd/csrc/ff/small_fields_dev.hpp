// Device-side single-word NTT fields for gfx950.
//
//   gl64_dev : Goldilocks p = 2^64 - 2^32 + 1, canonical u64, NOT Montgomery --
//              the wire format of the reference's gl64_t (ff/gl64_t.cuh:39-587;
//              test inputs are `random::<u64>() % p`, poc/ntt-cuda/tests/ntt.rs:14-17).
//              64x64->128 is four v_mad_u64_u32 (hipcc lowers unsigned __int128 /
//              __umul64hi to exactly that), then 2^64 = 2^32-1, 2^96 = -1 folding.
//   bb31_dev : BabyBear p = 0x78000001, Montgomery with R = 2^32 -- the wire format
//              of mont32_t<31,0x78000001,0x77ffffff,0x45dddde3,0x0ffffffe>
//              (ff/baby_bear.hpp:19, ff/mont32_t.cuh:19-425).
//
// Root-of-unity conventions follow ntt/parameters/goldilocks.h:84-159 and
// ntt/parameters/baby_bear.h:76-175 (default branches): every table entry
// forward_roots_of_unity[k] is a repeated square of the last one, so only that
// last entry and the coset generator are stored here.
#pragma once
#include "mont_dev.hpp"     // SPPARK_DEVFN, u32/u64

namespace sppark_amd {

struct gl64_dev {
    static constexpr u64 MOD = 0xffffffff00000001ULL;
    static constexpr unsigned TWO_ADICITY = 32;
    static constexpr u64 TOP_ROOT = 0x185629dcda58878cULL;      // 7^((p-1)/2^32)
    static constexpr u64 GROUP_GEN = 7;
    typedef u64 word_t;
    u64 v;

    SPPARK_DEVFN static gl64_dev from_raw(u64 x) { gl64_dev r; r.v = x; return r; }
    SPPARK_DEVFN static gl64_dev one() { return from_raw(1); }
    SPPARK_DEVFN static gl64_dev top_root() { return from_raw(TOP_ROOT); }
    SPPARK_DEVFN static gl64_dev group_gen() { return from_raw(GROUP_GEN); }

    SPPARK_DEVFN friend gl64_dev operator+(gl64_dev a, gl64_dev b)
    {
        u64 s = a.v + b.v;
        u64 c = s < a.v;                        // wrapped: s + 2^64 = s + (2^32-1) mod p
        s += (0 - c) & 0xffffffffULL;           // cannot wrap again: a,b < p
        s -= (s >= MOD) ? MOD : 0;
        return from_raw(s);
    }
    SPPARK_DEVFN friend gl64_dev operator-(gl64_dev a, gl64_dev b)
    {
        u64 d = a.v - b.v;
        u64 bw = a.v < b.v;
        d -= (0 - bw) & 0xffffffffULL;          // - 2^64 = -(2^32-1) mod p ; a,b canonical => no second wrap
        return from_raw(d);
    }
    SPPARK_DEVFN static u64 mulhi(u64 a, u64 b)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        return __umul64hi(a, b);
#else
        return (u64)(((unsigned __int128)a * b) >> 64);
#endif
    }
    SPPARK_DEVFN friend gl64_dev operator*(gl64_dev a, gl64_dev b)
    {
        u64 lo = a.v * b.v, hi = mulhi(a.v, b.v);
        u64 hl = hi & 0xffffffffULL, hh = hi >> 32;
        // x = lo + hl*2^64 + hh*2^96  =  lo - hh + hl*(2^32-1)   (mod p)
        u64 t = lo - hh;
        t -= (0 - (u64)(lo < hh)) & 0xffffffffULL;
        u64 m = (hl << 32) - hl;
        u64 s = t + m;
        s += (0 - (u64)(s < t)) & 0xffffffffULL;
        s -= (s >= MOD) ? MOD : 0;
        return from_raw(s);
    }
};

struct bb31_dev {
    static constexpr u32 MOD = 0x78000001u, M = 0x77ffffffu, RR = 0x45dddde3u, ONE = 0x0ffffffeu;
    static constexpr unsigned TWO_ADICITY = 27;
    static constexpr u32 TOP_ROOT = 0x1ffffedcu;                // Montgomery form
    typedef u32 word_t;
    u32 v;

    SPPARK_DEVFN static bb31_dev from_raw(u32 x) { bb31_dev r; r.v = x; return r; }
    SPPARK_DEVFN static bb31_dev one() { return from_raw(ONE); }
    SPPARK_DEVFN static bb31_dev top_root() { return from_raw(TOP_ROOT); }
    SPPARK_DEVFN static bb31_dev group_gen() { return from_raw(3) * from_raw(RR); }    // 3 in Montgomery form

    SPPARK_DEVFN friend bb31_dev operator+(bb31_dev a, bb31_dev b)
    {   u32 s = a.v + b.v; s -= (s >= MOD) ? MOD : 0; return from_raw(s);   }
    SPPARK_DEVFN friend bb31_dev operator-(bb31_dev a, bb31_dev b)
    {   u32 d = a.v - b.v; d += (a.v < b.v) ? MOD : 0; return from_raw(d);   }
    SPPARK_DEVFN friend bb31_dev operator*(bb31_dev a, bb31_dev b)
    {
        u64 t = (u64)a.v * b.v;
        u32 m = (u32)t * M;
        u64 u = (t + (u64)m * MOD) >> 32;       // < 2p
        u32 r = (u32)u; r -= (r >= MOD) ? MOD : 0;
        return from_raw(r);
    }
};

template<class F> SPPARK_DEVFN F field_pow(F b, u64 e)
{
    F r = F::one();
    while (e) { if (e & 1) r = r * b; b = b * b; e >>= 1; }
    return r;
}

} // namespace sppark_amd
