// Reduced-radix (30-bit limb) Montgomery field for gfx950 — the arithmetic the
// MSM bucket kernels run on.
//
// Why not 32-bit limbs (mont_dev.hpp)?  Measured on MI355X
// (profiles/r01_ubench2_instruction_rates.log): v_mad_u64_u32 issues at half
// rate (~4 cycles per wave64) and so does every carry-consuming VALU op
// (v_addc_co_u32 ~4 cycles); plain 32-bit ops take ~2.  With full 32-bit limbs
// every partial product needs a mad AND an addc (the mad has a carry-out but no
// carry-in), so half of a 12-limb product's ~2400 cycles is carry bookkeeping.
// With 30-bit limbs a product is < 2^60 and SIXTEEN of them fit a 64-bit
// accumulator: a column is a pure chain of v_mad_u64_u32 and the carry between
// columns is one 64-bit shift.  381 bits = 13 limbs: 338 + 13 multiplier ops and
// ~130 cheap ops per product instead of 300 + 288 carry ops.
//
// The wire format stays the reference's (32-bit limbs, Montgomery R = 2^(32N),
// ff/bls12-381.hpp:13-33): elements are converted on load/store, and the
// Montgomery reduction divides by exactly 2^(32N) = 2^(30*(NL-1)) * 2^TOP by
// taking NL-1 full 30-bit steps and one last TOP-bit step.
//
// Values are kept canonical (< p) between operations.
#pragma once
#include "mont_dev.hpp"

namespace sppark_amd {

template<class P> struct mont30_dev {
    static constexpr int N = P::N;                          // 32-bit wire words
    static constexpr int NL = (32 * P::N + 29) / 30;        // 30-bit limbs (13 for 384 bits, 9 for 256)
    static constexpr int TOP = 32 * P::N - 30 * (NL - 1);   // bits of the last reduction step (24 / 16)
    static constexpr u32 M30 = (1u << 30) - 1;
    u32 l[NL];

    // limb j of a little-endian array of 32-bit words
    SPPARK_DEVFN static constexpr u32 limb_of(const u32* w, int j)
    {
        const int bit = 30 * j, wi = bit >> 5, sh = bit & 31;
        u64 two = w[wi];
        if (wi + 1 < N) two |= (u64)w[wi + 1] << 32;
        return (u32)(two >> sh) & M30;
    }
    SPPARK_DEVFN static constexpr u32 mod_limb(int j) { return limb_of(P::MOD, j); }

    SPPARK_DEVFN static mont30_dev from_wire(const u32* w)
    {
        mont30_dev r;
        #pragma unroll
        for (int j = 0; j < NL; j++) r.l[j] = limb_of(w, j);
        return r;
    }
    SPPARK_DEVFN void to_wire(u32* w) const
    {
        #pragma unroll
        for (int i = 0; i < N; i++) {
            const int bit = 32 * i, j = bit / 30, sh = bit % 30;      // word i starts inside limb j
            u64 acc = (u64)l[j] >> sh;
            if (j + 1 < NL) acc |= (u64)l[j + 1] << (30 - sh);
            if (j + 2 < NL && 60 - sh < 32) acc |= (u64)l[j + 2] << (60 - sh);
            w[i] = (u32)acc;
        }
    }

    SPPARK_DEVFN static mont30_dev zero()
    {   mont30_dev r; for (int j = 0; j < NL; j++) r.l[j] = 0; return r;   }
    SPPARK_DEVFN static mont30_dev one()
    {   mont30_dev r; for (int j = 0; j < NL; j++) r.l[j] = limb_of(P::ONE, j); return r;   }

    SPPARK_DEVFN bool is_zero() const
    {
        u32 acc = l[0];
        #pragma unroll
        for (int j = 1; j < NL; j++) acc |= l[j];
        return acc == 0;
    }

    // t (limbs < 2^31, value < 2p) -> canonical: propagate carries, subtract p if >= p
    SPPARK_DEVFN static mont30_dev normalize(const u32 t[NL])
    {
        u32 s[NL], d[NL], c = 0;
        #pragma unroll
        for (int j = 0; j < NL; j++) { u32 v = t[j] + c; s[j] = v & M30; c = v >> 30; }
        // d = s - p with borrow through arithmetic shifts (limbs are < 2^30, so v - p - b fits an int)
        int bw = 0;
        #pragma unroll
        for (int j = 0; j < NL; j++) {
            int v = (int)s[j] - (int)mod_limb(j) - bw;
            d[j] = (u32)v & M30; bw = (v >> 31) & 1;
        }
        mont30_dev r;
        #pragma unroll
        for (int j = 0; j < NL; j++) r.l[j] = bw ? s[j] : d[j];
        return r;
    }

    SPPARK_DEVFN friend mont30_dev operator+(const mont30_dev& a, const mont30_dev& b)
    {
        u32 t[NL];
        #pragma unroll
        for (int j = 0; j < NL; j++) t[j] = a.l[j] + b.l[j];
        return normalize(t);
    }
    SPPARK_DEVFN friend mont30_dev operator-(const mont30_dev& a, const mont30_dev& b)
    {
        // a - b + p, limb-wise with a signed borrow chain, then canonicalise
        u32 t[NL]; int bw = 0;
        #pragma unroll
        for (int j = 0; j < NL; j++) {
            int v = (int)a.l[j] + (int)mod_limb(j) - (int)b.l[j] - bw;      // in (-2^30, 2^31)
            t[j] = (u32)v & M30; bw = (v >> 31) & 1;
            if (v >= (1 << 30)) { t[j] = (u32)v - (1u << 30); bw = -1; }     // carry out = negative borrow
        }
        return normalize(t);
    }
    SPPARK_DEVFN mont30_dev dbl() const { return *this + *this; }
    SPPARK_DEVFN mont30_dev neg() const
    {
        mont30_dev z = zero();
        return is_zero() ? z : z - *this;
    }
    SPPARK_DEVFN mont30_dev cneg(bool flag) const
    {
        mont30_dev n = neg(), r;
        #pragma unroll
        for (int j = 0; j < NL; j++) r.l[j] = flag ? n.l[j] : l[j];
        return r;
    }

    // number of a*b (= m*p) partial products in column k
    SPPARK_DEVFN static constexpr int col_count(int k) { return (k < NL ? k : 2 * NL - 2 - k) + 1; }

    // Montgomery product a*b / 2^(32N) mod p
    SPPARK_DEVFN friend mont30_dev operator*(const mont30_dev& a, const mont30_dev& b)
    {
        constexpr u32 PINV = P::M0 & M30;                   // -1/p mod 2^30
        u32 m[NL], c[NL + 2];
        u64 A = 0;                                          // a*b products (+ carry-in)
        #pragma unroll
        for (int k = 0; k < 2 * NL; k++) {
            const bool split = 2 * col_count(k) > 16;       // > 16 products: second accumulator
            u64 B = 0;
            if (k <= 2 * NL - 2) {
                #pragma unroll
                for (int i = 0; i < NL; i++) {
                    const int j = k - i;
                    if (j < 0 || j >= NL) continue;
                    A += (u64)a.l[i] * b.l[j];
                }
                #pragma unroll
                for (int i = 0; i < NL; i++) {
                    const int j = k - i;
                    if (j < 0 || j >= NL || i >= k) continue;       // i < k: m[i] already computed
                    if (split) B += (u64)m[i] * mod_limb(j);
                    else       A += (u64)m[i] * mod_limb(j);
                }
            }
            if (k < NL) {
                const u32 lo = split ? ((u32)A + (u32)B) : (u32)A;
                const u32 stepmask = k < NL - 1 ? M30 : ((1u << TOP) - 1);
                m[k] = (lo * PINV) & stepmask;
                if (split) B += (u64)m[k] * mod_limb(0);
                else       A += (u64)m[k] * mod_limb(0);
            }
            // column end: limb out (kept only from column NL-1 on), carry to the next column
            u32 limb; u64 carry;
            if (split) {
                const u32 s = ((u32)A & M30) + ((u32)B & M30);
                limb = s & M30;
                carry = (A >> 30) + (B >> 30) + (s >> 30);
            } else {
                limb = (u32)A & M30;
                carry = A >> 30;
            }
            if (k >= NL - 1) c[k - (NL - 1)] = limb;
            A = carry;
        }
        c[NL + 1] = (u32)A;                                 // zero for in-range inputs
        // result bit 0 sits at bit TOP of c[0]
        u32 t[NL];
        #pragma unroll
        for (int j = 0; j < NL; j++) t[j] = (c[j] >> TOP) | ((c[j + 1] << (30 - TOP)) & M30);
        return normalize(t);
    }
    SPPARK_DEVFN mont30_dev sqr() const { return *this * *this; }

    // out of / into Montgomery form (test hooks; the MSM never needs them)
    SPPARK_DEVFN mont30_dev from() const
    {   mont30_dev o = zero(); o.l[0] = 1; return *this * o;   }
    SPPARK_DEVFN mont30_dev to() const
    {   mont30_dev rr; for (int j = 0; j < NL; j++) rr.l[j] = limb_of(P::RR, j); return *this * rr;   }
};

} // namespace sppark_amd
