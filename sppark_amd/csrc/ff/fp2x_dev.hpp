// Quadratic extension Fp2 = Fp[u]/(u^2 + NR) over the loosely-reduced reduced-radix field of montx_dev.hpp: the
// coordinate field of the G2 bucket pipeline (round 3; until then fp2_dev.hpp over the canonical 32-bit-limb class,
// which stays the wire-format type).  The reference splits one Fp2 element over two lanes (ff/bls12-381-fp2.hpp:25-150);
// here one lane owns c0 | c1.
//
// What the representation buys (the instruction COUNT is the currency, montx_dev.hpp):
//   * a product is two sums of two base products, each with ONE Montgomery reduction (montx_dev::mul_add):
//         c1 = a0 b1 + a1 b0            c0 = a0 b0 + (K p - a1) (NR b1)
//     4 NL^2 + 2 NL^2 = 1176 multiply-adds for NL = 14 and no carry instruction, against three canonical products
//     (3 x 680 instructions) and five modular additions (5 x 50) of the Karatsuba form over mont_dev;
//   * a square (NR = 1) is (a0 + a1)(a0 - a1) | 2 a0 a1 as one interleaved pair of base products;
//   * additions and subtractions are limb-wise, without carries or conditional subtractions.
//
// CONTRACT (the caller's, stated at every use in ec/xyzzx2_dev.hpp): every operand handed to mul / sqr is NORMALISED
// (limbs < 2^LB, norm()) and its value bound K (value < K p, both components) is a template argument where a negation
// needs it.  Values stay below 16 p, far inside the 2^(LB*NL) / p head-room of the radix (2^11 for 381 bits).
#pragma once
#include "montx_dev.hpp"

namespace sppark_amd {

template<class P, int LB> struct fp2x_dev {
    typedef montx_dev<P, LB> fp;
    static constexpr int NL = fp::NL;
    static constexpr int N = 2 * fp::NL;                // words of the in-memory image (internal limbs)
    static constexpr int NW = 2 * P::N;                 // 32-bit words of the standard wire form
    static constexpr int FP2_NR = P::FP2_NR;            // u^2 = -FP2_NR
    fp c0, c1;

    SPPARK_DEVFN static fp2x_dev from_wire(const u32* w)
    {   fp2x_dev r; r.c0 = fp::from_wire(w); r.c1 = fp::from_wire(w + NL); return r;   }
    SPPARK_DEVFN void to_wire(u32* w) const { c0.to_wire(w); c1.to_wire(w + NL); }
    // standard wire form (c0 | c1, canonical Montgomery words) <-> internal
    SPPARK_DEVFN static fp2x_dev from_std(const u32* w)
    {   fp2x_dev r; r.c0 = fp::from_std(w); r.c1 = fp::from_std(w + P::N); return r;   }
    SPPARK_DEVFN void to_std(u32* w) const { c0.to_std(w); c1.to_std(w + P::N); }

    SPPARK_DEVFN static fp2x_dev zero() { fp2x_dev r; r.c0 = fp::zero(); r.c1 = fp::zero(); return r; }
    SPPARK_DEVFN static fp2x_dev one()  { fp2x_dev r; r.c0 = fp::one();  r.c1 = fp::zero(); return r; }
    SPPARK_DEVFN bool limbs_all_zero() const { return c0.limbs_all_zero() & c1.limbs_all_zero(); }
    SPPARK_DEVFN fp2x_dev norm() const { fp2x_dev r; r.c0 = c0.norm(); r.c1 = c1.norm(); return r; }

    SPPARK_DEVFN friend fp2x_dev operator+(const fp2x_dev& a, const fp2x_dev& b)
    {   fp2x_dev r; r.c0 = a.c0 + b.c0; r.c1 = a.c1 + b.c1; return r;   }
    // a + K p - b / K p - b, component-wise (montx_dev::sub / neg: b < (K-1) p, b's limbs <= B (2^LB - 1))
    template<int K, int B = 1> SPPARK_DEVFN static fp2x_dev sub(const fp2x_dev& a, const fp2x_dev& b)
    {   fp2x_dev r; r.c0 = fp::template sub<K, B>(a.c0, b.c0); r.c1 = fp::template sub<K, B>(a.c1, b.c1); return r;   }
    template<int K, int B = 1> SPPARK_DEVFN static fp2x_dev neg(const fp2x_dev& b)
    {   fp2x_dev r; r.c0 = fp::template neg<K, B>(b.c0); r.c1 = fp::template neg<K, B>(b.c1); return r;   }

    // NR * x for the right-hand operand of a product (normalised in, normalised out)
    SPPARK_DEVFN static fp mul_nr(const fp& x)
    {
        if constexpr (P::FP2_NR == 1) return x;
        else { static_assert(P::FP2_NR == 5, "non-residue"); return (x + x + x + x + x).norm(); }
    }

    // a * b.  a, b normalised; a < (KA - 1) p.  Result normalised, < 2 p:
    // (a0 b1 + a1 b0) / R + p with a, b < 16 p is < (512 p / R + 1) p.
    template<int KA> SPPARK_DEVFN static fp2x_dev mul(const fp2x_dev& a, const fp2x_dev& b)
    {
        fp2x_dev r;
        r.c1 = fp::mul_add(a.c0, b.c1, a.c1, b.c0);
        const fp na1 = fp::template neg<KA, 1>(a.c1);               // limbs <= 2 * 2^LB: admissible second left operand
        r.c0 = fp::mul_add(a.c0, b.c0, na1, mul_nr(b.c1));
        return r;
    }
    // a^2.  a normalised, a < (KA - 1) p.  Result normalised, < 2 p.
    template<int KA> SPPARK_DEVFN fp2x_dev sqr() const
    {
        if constexpr (P::FP2_NR == 1) {
            fp2x_dev r;
            const fp s = c0 + c1;                                   // limbs < 2 * 2^LB: a left operand
            const fp d = fp::template sub<KA, 1>(c0, c1).norm();    // c0 - c1 + KA p, normalised
            fp::mul2(r.c0, r.c1, s, d, c0 + c0, c1);
            return r;
        } else {
            return mul<KA>(*this, *this);
        }
    }

    // value == 0 in Fp2 for a normalised value < KMAX p
    template<int KMAX> SPPARK_DEVFN bool is_zero_mod() const
    {   return c0.template is_zero_mod<KMAX>() && c1.template is_zero_mod<KMAX>();   }
};

} // namespace sppark_amd
