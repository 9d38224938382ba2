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

#if defined(__HIP_DEVICE_COMPILE__) && !defined(SPPARK_GL64_PLAIN_C)
# define SPPARK_GL64_ASM 1
#endif

namespace sppark_amd {

struct gl64_dev {
    static constexpr u64 MOD = 0xffffffff00000001ULL;
    static constexpr unsigned TWO_ADICITY = 32;
    // root convention: ntt/parameters/goldilocks.h:84-159 by default, :7-82 with -DGOLDILOCKS_PLONKY2
    // (plonky2's generator; libsppark_gl64_plonky2.so)
#ifdef GOLDILOCKS_PLONKY2
    static constexpr u64 TOP_ROOT = 0x64fdd1a46201e246ULL;      // 0xc65c18b67785d900^((p-1)/2^32)
    static constexpr u64 GROUP_GEN = 0xc65c18b67785d900ULL;
#else
    static constexpr u64 TOP_ROOT = 0x185629dcda58878cULL;      // 7^((p-1)/2^32)
    static constexpr u64 GROUP_GEN = 7;
#endif
    typedef u64 word_t;
    u64 v;

    SPPARK_DEVFN static gl64_dev from_raw(u64 x) { gl64_dev r; r.v = x; return r; }
    SPPARK_DEVFN static gl64_dev one() { return from_raw(1); }
    SPPARK_DEVFN static gl64_dev top_root() { return from_raw(TOP_ROOT); }
    SPPARK_DEVFN static gl64_dev group_gen() { return from_raw(GROUP_GEN); }

    // Carry-generating VALU ops are the expensive ones on gfx950 (v_add_co/v_addc_co
    // issue at half the rate of plain 32-bit ops, and hipcc's u64 compare+select
    // lowering adds v_cmp_*_u64 + v_cndmask + s_nop on top), so the modular
    // corrections are written as explicit carry chains:
    //   a + b: both corrections ("sum wrapped 2^64" and "sum >= p") are the same
    //          operation, + (2^32 - 1) mod 2^64, needed iff either carry is set.
    //   a - b: on borrow, - (2^32 - 1).
    SPPARK_DEVFN friend gl64_dev operator+(gl64_dev a, gl64_dev b)
    {
#if defined(SPPARK_GL64_ASM)
        u32 a0 = (u32)a.v, a1 = (u32)(a.v >> 32), b0 = (u32)b.v, b1 = (u32)(b.v >> 32);
        u32 lo, hi, ulo, uhi;
        u64 c1, c2;
        // gfx940+ needs two wait states between a VALU that writes an SGPR pair
        // (carry-out) and a VALU that reads it; independent instructions are
        // placed in those slots where possible, s_nop otherwise.
        asm("v_add_co_u32 %0, %4, %6, %8\n\t"
            "v_add_co_u32 %2, %5, -1, %0\n\t"
            "s_nop 0\n\t"
            "v_addc_co_u32 %1, %4, %7, %9, %4\n\t"
            "s_nop 0\n\t"
            "v_addc_co_u32 %3, %5, 0, %1, %5\n\t"
            "s_nop 1\n\t"
            "s_or_b64 %4, %4, %5\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32 %0, %0, %2, %4\n\t"
            "v_cndmask_b32 %1, %1, %3, %4"
            : "=&v"(lo), "=&v"(hi), "=&v"(ulo), "=&v"(uhi), "=&s"(c1), "=&s"(c2)
            : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "scc");      // s_or_b64 writes SCC
        return from_raw(((u64)hi << 32) | lo);
#else
        u64 s = a.v + b.v;
        u64 c = s < a.v;                        // wrapped: s + 2^64 = s + (2^32-1) mod p
        s += (0 - c) & 0xffffffffULL;           // cannot wrap again: a,b < p
        s -= (s >= MOD) ? MOD : 0;
        return from_raw(s);
#endif
    }
    SPPARK_DEVFN friend gl64_dev operator-(gl64_dev a, gl64_dev b)
    {
#if defined(SPPARK_GL64_ASM)
        u32 a0 = (u32)a.v, a1 = (u32)(a.v >> 32), b0 = (u32)b.v, b1 = (u32)(b.v >> 32);
        u32 lo, hi, t;
        asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
            "s_nop 1\n\t"
            "v_subb_co_u32 %1, vcc, %4, %6, vcc\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32 %2, 0, -1, vcc\n\t"
            "v_sub_co_u32 %0, vcc, %0, %2\n\t"
            "s_nop 1\n\t"
            "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
            : "=&v"(lo), "=&v"(hi), "=&v"(t)
            : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "vcc");
        return from_raw(((u64)hi << 32) | lo);
#else
        u64 d = a.v - b.v;
        u64 bw = a.v < b.v;
        d -= (0 - bw) & 0xffffffffULL;          // - 2^64 = -(2^32-1) mod p ; a,b canonical => no second wrap
        return from_raw(d);
#endif
    }
    SPPARK_DEVFN friend gl64_dev operator*(gl64_dev a, gl64_dev b)
    {
        // 64x64 -> 128 as exactly four v_mad_u64_u32 (schoolbook on 32-bit halves; each
        // partial product absorbs a 32-bit carry-in without overflowing 64 bits)
        const u32 a0 = (u32)a.v, a1 = (u32)(a.v >> 32), b0 = (u32)b.v, b1 = (u32)(b.v >> 32);
        const u64 p00 = (u64)a0 * b0;
        const u64 p01 = (u64)a0 * b1 + (p00 >> 32);
        const u64 p10 = (u64)a1 * b0 + (u32)p01;
        const u64 p11 = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
        return reduce128((u32)p00, (u32)p10, (u32)p11, (u32)(p11 >> 32));
    }
    // w0 + w1*2^32 + w2*2^64 + w3*2^96 (mod p), canonical.  2^64 = 2^32 - 1 =: E and
    // 2^96 = -1, so the value is V = (w1:w0) + w2*E - w3.  One mad (carry c) and one
    // 64-bit subtraction (borrow b) leave u = V - (c - b)*2^64, and (c - b)*2^64 =
    // (c - b)*E is added back as ONE 64-bit constant D in {-E, 0, +E} (neither
    // direction can wrap again: V < 2^65 - 2^33 and V > -2^32); then x >= p ? x - p.
    // Seven carry-chain instructions instead of the twelve of canon + sub + add.
    SPPARK_DEVFN static gl64_dev reduce128(u32 w0, u32 w1, u32 w2, u32 w3)
    {
#if defined(SPPARK_GL64_ASM)
        u64 t, c, c2;
        const u64 lo64 = ((u64)w1 << 32) | w0;
        asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=&v"(t), "=&s"(c) : "v"(w2), "v"(lo64));
        const u32 t0 = (u32)t, t1 = (u32)(t >> 32);
        u32 u0, u1, x0, x1, k, kb, nk, r0, r1;
        asm("v_sub_co_u32 %[u0], vcc, %[t0], %[w3]\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32 %[k], 0, -1, %[c]\n\t"
            "v_subbrev_co_u32 %[u1], vcc, 0, %[t1], vcc\n\t"
            "v_not_b32 %[nk], %[k]\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32 %[kb], 0, -1, vcc\n\t"
            "v_sub_u32 %[k], %[k], %[kb]\n\t"             // D.lo = k - kb
            "v_and_b32 %[nk], %[nk], %[kb]\n\t"           // D.hi = ~k & kb
            "v_add_co_u32 %[u0], vcc, %[u0], %[k]\n\t"
            "v_add_co_u32 %[x0], %[c2], -1, %[u0]\n\t"
            "s_nop 0\n\t"
            "v_addc_co_u32 %[u1], vcc, %[u1], %[nk], vcc\n\t"
            "s_nop 0\n\t"
            "v_addc_co_u32 %[x1], %[c2], 0, %[u1], %[c2]\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32 %[r0], %[u0], %[x0], %[c2]\n\t"
            "v_cndmask_b32 %[r1], %[u1], %[x1], %[c2]"
            : [u0]"=&v"(u0), [u1]"=&v"(u1), [x0]"=&v"(x0), [x1]"=&v"(x1), [k]"=&v"(k), [kb]"=&v"(kb),
              [nk]"=&v"(nk), [r0]"=&v"(r0), [r1]"=&v"(r1), [c2]"=&s"(c2)
            : [t0]"v"(t0), [t1]"v"(t1), [w3]"v"(w3), [c]"s"(c) : "vcc");
        return from_raw(((u64)r1 << 32) | r0);
#else
        const u64 lo64 = ((u64)w1 << 32) | w0, E = 0xffffffffULL;
        u64 t = lo64 + (u64)w2 * E;
        const u64 c = t < lo64;
        const u64 b = t < w3;
        u64 u = t - w3;
        if (c && !b) u += E; else if (b && !c) u -= E;
        return from_raw(u >= MOD ? u - MOD : u);
#endif
    }
    // (a + b, a - b) in one interleaved sequence: the add chain (SGPR-pair carries)
    // and the sub chain (VCC) fill each other's mandatory wait states.
    SPPARK_DEVFN static void bfly(gl64_dev a, gl64_dev b, gl64_dev& s, gl64_dev& d)
    {
#if defined(SPPARK_GL64_ASM)
        u32 a0 = (u32)a.v, a1 = (u32)(a.v >> 32), b0 = (u32)b.v, b1 = (u32)(b.v >> 32);
        u32 lo, hi, ulo, uhi, dlo, dhi, t;
        u64 c1, c2;
        asm("v_add_co_u32 %[lo], %[c1], %[a0], %[b0]\n\t"
            "v_sub_co_u32 %[dlo], vcc, %[a0], %[b0]\n\t"
            "v_add_co_u32 %[ulo], %[c2], -1, %[lo]\n\t"
            "v_addc_co_u32 %[hi], %[c1], %[a1], %[b1], %[c1]\n\t"
            "v_subb_co_u32 %[dhi], vcc, %[a1], %[b1], vcc\n\t"
            "v_addc_co_u32 %[uhi], %[c2], 0, %[hi], %[c2]\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32 %[t], 0, -1, vcc\n\t"
            "s_nop 0\n\t"
            "s_or_b64 %[c1], %[c1], %[c2]\n\t"
            "v_sub_co_u32 %[dlo], vcc, %[dlo], %[t]\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32 %[lo], %[lo], %[ulo], %[c1]\n\t"
            "v_cndmask_b32 %[hi], %[hi], %[uhi], %[c1]\n\t"
            "v_subbrev_co_u32 %[dhi], vcc, 0, %[dhi], vcc"
            : [lo]"=&v"(lo), [hi]"=&v"(hi), [ulo]"=&v"(ulo), [uhi]"=&v"(uhi), [dlo]"=&v"(dlo), [dhi]"=&v"(dhi),
              [t]"=&v"(t), [c1]"=&s"(c1), [c2]"=&s"(c2)
            : [a0]"v"(a0), [a1]"v"(a1), [b0]"v"(b0), [b1]"v"(b1) : "vcc", "scc");
        s = from_raw(((u64)hi << 32) | lo);
        d = from_raw(((u64)dhi << 32) | dlo);
#else
        s = a + b; d = a - b;
#endif
    }
    // x >= p ? x - p : x      (x - p = x + 2^32 - 1 mod 2^64, and x >= p iff that addition carries)
    SPPARK_DEVFN static gl64_dev canon(u64 x)
    {
#if defined(SPPARK_GL64_ASM)
        u32 x0 = (u32)x, x1 = (u32)(x >> 32), u0, u1;
        asm("v_add_co_u32 %0, vcc, -1, %2\n\t"
            "s_nop 1\n\t"
            "v_addc_co_u32 %1, vcc, 0, %3, vcc\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32 %0, %2, %0, vcc\n\t"
            "v_cndmask_b32 %1, %3, %1, vcc"
            : "=&v"(u0), "=&v"(u1) : "v"(x0), "v"(x1) : "vcc");
        return from_raw(((u64)u1 << 32) | u0);
#else
        return from_raw(x >= MOD ? x - MOD : x);
#endif
    }

    // lo + hi*2^64 (mod p), canonical
    SPPARK_DEVFN static gl64_dev reduce_u96(u64 lo, u32 hi)
    {   return canon(lo) + from_raw(((u64)hi << 32) - hi);   }
    // x * 2^e, e in [0, 192): 2 has order 192 in this field (2^96 = -1) and every
    // root of unity of order <= 64 is a power of two -- w_64 = 2^39 in the
    // reference's table (ntt/parameters/goldilocks.h:86-93: 0x8000000000 = 2^39).
    // With e known at compile time (unrolled butterflies) this is a handful of
    // shifts/adds instead of a 64x64 product.
    SPPARK_DEVFN static gl64_dev mul_pow2(gl64_dev x, unsigned e)     // e in [0, 96)
    {
        const unsigned q = e >> 5, sh = e & 31;
        const u32 x0 = (u32)x.v, x1 = (u32)(x.v >> 32);
        const u32 y0 = x0 << sh;
        const u32 y1 = sh ? (x1 << sh) | (x0 >> (32 - sh)) : x1;
        const u32 y2 = sh ? x1 >> (32 - sh) : 0;
        if (q == 0) return fold_u96(y0, y1, y2);
        if (q == 1) {                                   // (x*2^sh) * 2^32: two folds (2 mads + 8) beat reduce128's 1 + 17
            const gl64_dev z = fold_u96(y0, y1, y2);
            return fold_u96(0, (u32)z.v, (u32)(z.v >> 32));
        }
        // y0*2^64 + y1*2^96 + y2*2^128, and 2^96 = -1, 2^128 = -2^32:  y0*E - (y2:y1)
        return fold_hi(y0, y1, y2);
    }
    // (w1:w0) + w2*E, canonical, for w2 < 2^31 (a 64-bit value shifted left by < 32 bits) or for
    // w0 = 0 and any w2 (a canonical value times 2^32).  The multiply-add wraps at most once and the
    // wrapped value + E is below p in both cases (V < 2^64 + 2^63, resp. V <= (2^32-1)(2^33-1)); the
    // two possible corrections -- "it wrapped" and "it is >= p" -- are the same addition of E,
    // exactly as in operator+: one mad, two carry instructions and two selects instead of
    // reduce128's seventeen.
    SPPARK_DEVFN static gl64_dev fold_u96(u32 w0, u32 w1, u32 w2)
    {
#if defined(SPPARK_GL64_ASM)
        u64 t, c, c2;
        const u64 lo64 = ((u64)w1 << 32) | w0;
        asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=&v"(t), "=&s"(c) : "v"(w2), "v"(lo64));
        const u32 t0 = (u32)t, t1 = (u32)(t >> 32);
        u32 x0, x1, r0, r1;
        asm("v_add_co_u32 %[x0], %[c2], -1, %[t0]\n\t"
            "s_nop 1\n\t"
            "v_addc_co_u32 %[x1], %[c2], 0, %[t1], %[c2]\n\t"
            "s_nop 1\n\t"
            "s_or_b64 %[c2], %[c2], %[c]\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32 %[r0], %[t0], %[x0], %[c2]\n\t"
            "v_cndmask_b32 %[r1], %[t1], %[x1], %[c2]"
            : [x0]"=&v"(x0), [x1]"=&v"(x1), [r0]"=&v"(r0), [r1]"=&v"(r1), [c2]"=&s"(c2)
            : [t0]"v"(t0), [t1]"v"(t1), [c]"s"(c) : "scc");
        return from_raw(((u64)r1 << 32) | r0);
#else
        const u64 lo64 = ((u64)w1 << 32) | w0, E = 0xffffffffULL;
        u64 t = lo64 + (u64)w2 * E;
        if (t < lo64) return from_raw(t + E);           // wrapped once: + 2^64 = + E, result < 2^63 + 2^32
        return from_raw(t >= MOD ? t - MOD : t);
#endif
    }
    // w0*E - (w2:w1), canonical, for w2 < 2^31 (the words of a value shifted to 2^64 and above).
    // w0*E <= (2^32-1)^2 < p and (w2:w1) < 2^63: one multiply, one 64-bit subtraction, and + p on
    // borrow (the tail of operator-).
    SPPARK_DEVFN static gl64_dev fold_hi(u32 w0, u32 w1, u32 w2)
    {
#if defined(SPPARK_GL64_ASM)
        u64 t, c;
        const u64 zero = 0;
        asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=&v"(t), "=&s"(c) : "v"(w0), "v"(zero));
        const u32 t0 = (u32)t, t1 = (u32)(t >> 32);
        u32 u0, u1, k;
        asm("v_sub_co_u32 %[u0], vcc, %[t0], %[w1]\n\t"
            "s_nop 1\n\t"
            "v_subb_co_u32 %[u1], vcc, %[t1], %[w2], vcc\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32 %[k], 0, -1, vcc\n\t"
            "v_sub_co_u32 %[u0], vcc, %[u0], %[k]\n\t"
            "s_nop 1\n\t"
            "v_subbrev_co_u32 %[u1], vcc, 0, %[u1], vcc"
            : [u0]"=&v"(u0), [u1]"=&v"(u1), [k]"=&v"(k)
            : [t0]"v"(t0), [t1]"v"(t1), [w1]"v"(w1), [w2]"v"(w2) : "vcc");
        return from_raw(((u64)u1 << 32) | u0);
#else
        const u64 t = (u64)w0 * 0xffffffffULL, s = ((u64)w2 << 32) | w1;
        u64 u = t - s;
        if (t < s) u -= 0xffffffffULL;                  // + p  (mod 2^64)
        return from_raw(u);
#endif
    }
    // w_{2^R}^k (w^-1 for INV) = +-2^e with the reference's root convention:
    // root_neg() says whether the sign is minus, mul_root() multiplies by 2^e only.
    // The butterflies absorb the sign by swapping their operands / outputs.
    template<bool INV>
    SPPARK_DEVFN static constexpr unsigned root_exp(unsigned R, unsigned k)
    {
#ifdef GOLDILOCKS_PLONKY2
        constexpr unsigned ER[7] = {0, 96, 48, 24, 12, 6, 3};       // w_{2^R} = 2^(192 / 2^R): w_64 = 8 (goldilocks.h:13-19)
#else
        constexpr unsigned ER[7] = {0, 96, 48, 120, 156, 78, 39};   // w_{2^R} = 2^ER[R]
#endif
        unsigned e = (ER[R] * k) % 192;
        if (INV) e = (192 - e) % 192;
        return e;
    }
    template<bool INV>
    SPPARK_DEVFN static constexpr bool root_neg(unsigned R, unsigned k) { return root_exp<INV>(R, k) >= 96; }
    template<bool INV>
    SPPARK_DEVFN static gl64_dev mul_root(gl64_dev x, unsigned R, unsigned k, const gl64_dev*)
    {
        const unsigned e = root_exp<INV>(R, k) % 96;
        return e ? mul_pow2(x, e) : x;
    }
    // p - a (0 for a == 0): 0 - a, and - (2^32 - 1) on borrow, i.e. the tail of operator-
    SPPARK_DEVFN static gl64_dev neg(gl64_dev a)
    {
#if defined(SPPARK_GL64_ASM)
        u32 a0 = (u32)a.v, a1 = (u32)(a.v >> 32), lo, hi, t;
        asm("v_sub_co_u32 %0, vcc, 0, %3\n\t"
            "s_nop 1\n\t"
            "v_subb_co_u32 %1, vcc, 0, %4, vcc\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32 %2, 0, -1, vcc\n\t"
            "v_sub_co_u32 %0, vcc, %0, %2\n\t"
            "s_nop 1\n\t"
            "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
            : "=&v"(lo), "=&v"(hi), "=&v"(t) : "v"(a0), "v"(a1) : "vcc");
        return from_raw(((u64)hi << 32) | lo);
#else
        return from_raw(a.v ? MOD - a.v : 0);
#endif
    }
    // x * w_{2^R}^k with the sign included (a twiddle that no butterfly follows directly:
    // the diagonal between the two radix-8 rounds of a radix-64 block, ntt/ntt_r64_kernels.hpp)
    template<bool INV>
    SPPARK_DEVFN static gl64_dev mul_root_full(gl64_dev x, unsigned R, unsigned k, const gl64_dev*)
    {
        const unsigned e = root_exp<INV>(R, k);
        if (e == 0) return x;
        if (e < 96) return mul_pow2(x, e);
        return neg(e == 96 ? x : mul_pow2(x, e - 96));
    }
    static constexpr bool SHIFT_ROOTS = true;
};

struct bb31_dev {
    static constexpr u32 MOD = 0x78000001u, M = 0x77ffffffu, RR = 0x45dddde3u, ONE = 0x0ffffffeu;
    static constexpr unsigned TWO_ADICITY = 27;
    // root convention: ntt/parameters/baby_bear.h:76-175 by default (group_gen 3, e.g. RISC Zero),
    // :7-74 with -DBABY_BEAR_CANONICAL (group_gen 31; libsppark_bb31_canonical.so)
#ifdef BABY_BEAR_CANONICAL
    static constexpr u32 TOP_ROOT = 0x57fab6eeu;                // Montgomery form of 31^((p-1)/2^27)
    static constexpr u32 GROUP_GEN = 31;
#else
    static constexpr u32 TOP_ROOT = 0x1ffffedcu;                // Montgomery form
    static constexpr u32 GROUP_GEN = 3;
#endif
    typedef u32 word_t;
    u32 v;

    SPPARK_DEVFN static bb31_dev from_raw(u32 x) { bb31_dev r; r.v = x; return r; }
    SPPARK_DEVFN static bb31_dev one() { return from_raw(ONE); }
    SPPARK_DEVFN static bb31_dev top_root() { return from_raw(TOP_ROOT); }
    SPPARK_DEVFN static bb31_dev group_gen() { return from_raw(GROUP_GEN) * from_raw(RR); }    // in Montgomery form

    // Conditional corrections as an unsigned MINIMUM: x in [0, 2p) -> min(x, x - p) (x < p: x - p wraps above x), and
    // for a difference d = a - b (mod 2^32) -> min(d, d + p) (a < b: d is huge and d + p wraps to the residue).  Two
    // instructions instead of compare + select + add: an addition is 3 instead of 4, a product 5 instead of 6 --
    // 13 % of the transform's vector instructions (profiles/r04_ntt_bb31_pmc.txt: the passes sit at the issue ceiling).
    SPPARK_DEVFN static u32 umin(u32 x, u32 y) { return x < y ? x : y; }
    SPPARK_DEVFN friend bb31_dev operator+(bb31_dev a, bb31_dev b)
    {   const u32 s = a.v + b.v; return from_raw(umin(s, s - MOD));   }
    SPPARK_DEVFN friend bb31_dev operator-(bb31_dev a, bb31_dev b)
    {   const u32 d = a.v - b.v; return from_raw(umin(d, d + MOD));   }
    SPPARK_DEVFN friend bb31_dev operator*(bb31_dev a, bb31_dev b)
    {
        u64 t = (u64)a.v * b.v;
        u32 m = (u32)t * M;
        const u32 r = (u32)((t + (u64)m * MOD) >> 32);          // < 2p
        return from_raw(umin(r, r - MOD));
    }
    // x * w_{2^R}^k from the per-(size, direction) table inner[(1 << R) + k]
    template<bool INV>
    SPPARK_DEVFN static bb31_dev mul_root(bb31_dev x, unsigned R, unsigned k, const bb31_dev* inner)
    {   return k ? x * inner[(1u << R) + k] : x;   }
    template<bool INV>
    SPPARK_DEVFN static constexpr bool root_neg(unsigned, unsigned) { return false; }
    template<bool INV>
    SPPARK_DEVFN static bb31_dev mul_root_full(bb31_dev x, unsigned R, unsigned k, const bb31_dev* inner)
    {   return k ? x * inner[(1u << R) + k] : x;   }
    SPPARK_DEVFN static void bfly(bb31_dev a, bb31_dev b, bb31_dev& s, bb31_dev& d) { s = a + b; d = a - b; }
    static constexpr bool SHIFT_ROOTS = false;
};

// Mersenne31, p = 2^31 - 1 (ff/mersenne31.hpp:13-60): the reference computes on Montgomery residues
// (R = 2^32 = 2 mod p, so to/from Montgomery is a shift by one) and stores CANONICAL residues
// (mrs31_t::mem_t); here the canonical residue is also the register form -- a product is one
// 31x31 multiply and two folds of 2^31 = 1.
struct mrs31_dev {
    static constexpr u32 MOD = 0x7fffffffu;
    typedef u32 word_t;
    u32 v;                                                      // [0, p)
    SPPARK_DEVFN static mrs31_dev from_raw(u32 x) { mrs31_dev r; r.v = x; return r; }
    SPPARK_DEVFN static mrs31_dev one() { return from_raw(1); }
    SPPARK_DEVFN friend mrs31_dev operator+(mrs31_dev a, mrs31_dev b)
    {   u32 s = a.v + b.v; s -= (s >= MOD) ? MOD : 0; return from_raw(s);   }
    SPPARK_DEVFN friend mrs31_dev operator-(mrs31_dev a, mrs31_dev b)
    {   u32 d = a.v - b.v; d += (a.v < b.v) ? MOD : 0; return from_raw(d);   }
    SPPARK_DEVFN friend mrs31_dev operator*(mrs31_dev a, mrs31_dev b)
    {
        const u64 t = (u64)a.v * b.v;                           // < 2^62
        u32 r = (u32)(t & MOD) + (u32)(t >> 31);                // < 2^32
        r = (r & MOD) + (r >> 31);                              // <= p
        r -= (r >= MOD) ? MOD : 0;
        return from_raw(r);
    }
};

// BabyBear quartic extension F_p[x]/(x^4 - beta) (bb31_4_t, ff/baby_bear.hpp:70-446): four Montgomery
// residues c0 | c1 | c2 | c3, the reference's memory image.  beta = -11 by default (x^4 + 11, as RISC Zero),
// +11 with -DBABY_BEAR_CANONICAL (ff/baby_bear.hpp:75-79).
struct alignas(16) bb31_4_dev {
#ifdef BABY_BEAR_CANONICAL
    static constexpr u32 BETA = 0x37ffffe9u;                    // (11 << 32) % p
#else
    static constexpr u32 BETA = 0x40000018u;                    // (-11 << 32) % p
#endif
    bb31_dev c[4];
    SPPARK_DEVFN static bb31_4_dev one()
    {   bb31_4_dev r; r.c[0] = bb31_dev::one(); r.c[1] = r.c[2] = r.c[3] = bb31_dev::from_raw(0); return r;   }
    SPPARK_DEVFN friend bb31_4_dev operator+(const bb31_4_dev& a, const bb31_4_dev& b)
    {   bb31_4_dev r; for (int i = 0; i < 4; i++) r.c[i] = a.c[i] + b.c[i]; return r;   }
    SPPARK_DEVFN friend bb31_4_dev operator-(const bb31_4_dev& a, const bb31_4_dev& b)
    {   bb31_4_dev r; for (int i = 0; i < 4; i++) r.c[i] = a.c[i] - b.c[i]; return r;   }
    // schoolbook: the degree-4..6 terms fold back with x^4 = beta
    SPPARK_DEVFN friend bb31_4_dev operator*(const bb31_4_dev& a, const bb31_4_dev& b)
    {
        const bb31_dev beta = bb31_dev::from_raw(BETA);
        bb31_dev lo[4], hi[3];
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            lo[k] = a.c[0] * b.c[k];
            #pragma unroll
            for (int i = 1; i <= k; i++) lo[k] = lo[k] + a.c[i] * b.c[k - i];
        }
        #pragma unroll
        for (int k = 4; k < 7; k++) {
            hi[k - 4] = a.c[k - 3] * b.c[3];
            #pragma unroll
            for (int i = k - 2; i < 4; i++) hi[k - 4] = hi[k - 4] + a.c[i] * b.c[k - i];
        }
        bb31_4_dev r;
        r.c[0] = lo[0] + beta * hi[0]; r.c[1] = lo[1] + beta * hi[1]; r.c[2] = lo[2] + beta * hi[2]; r.c[3] = lo[3];
        return r;
    }
};

template<class F> SPPARK_DEVFN F field_pow(F b, u64 e)
{
    F r = F::one();
    while (e) { if (e & 1) r = r * b; b = b * b; e >>= 1; }
    return r;
}

} // namespace sppark_amd
