// Quadratic extension Fp2 = Fp[u]/(u^2 + NR) on the device: the coordinate field of G2.
// NR = P::FP2_NR: 1 for BLS12-381 and alt_bn128 (u^2 = -1), 5 for BLS12-377 (u^2 = -5,
// ff/bls12-377-fp2.hpp).
//
// The reference splits one Fp2 element over two adjacent lanes and exchanges halves
// with warp shuffles (ff/bls12-381-fp2.hpp:25-150, `degree = 2`); here one lane owns
// the whole element: c0 | c1, the same memory image (fp_mont x[2], :33-34), so the
// G1 bucket pipeline is reused unchanged with F = fp2_dev<P>.  Products are
// Karatsuba over Fp (3 base products, 2 for a square).
#pragma once
#include "mont_dev.hpp"

namespace sppark_amd {

template<class P> struct fp2_dev {
    typedef mont_dev<P> fp;
    static constexpr int N = 2 * fp::N;                 // 32-bit wire words
    fp c0, c1;

    SPPARK_DEVFN static fp2_dev from_wire(const u32* w)
    {   fp2_dev r; r.c0 = fp::from_wire(w); r.c1 = fp::from_wire(w + fp::N); return r;   }
    SPPARK_DEVFN void to_wire(u32* w) const { c0.to_wire(w); c1.to_wire(w + fp::N); }

    SPPARK_DEVFN static fp2_dev zero() { fp2_dev r; r.c0 = fp::zero(); r.c1 = fp::zero(); return r; }
    SPPARK_DEVFN static fp2_dev one()  { fp2_dev r; r.c0 = fp::one();  r.c1 = fp::zero(); return r; }
    SPPARK_DEVFN bool is_zero() const { return c0.is_zero() & c1.is_zero(); }
    SPPARK_DEVFN bool equals(const fp2_dev& b) const { return c0.equals(b.c0) & c1.equals(b.c1); }

    SPPARK_DEVFN friend fp2_dev operator+(const fp2_dev& a, const fp2_dev& b)
    {   fp2_dev r; r.c0 = a.c0 + b.c0; r.c1 = a.c1 + b.c1; return r;   }
    SPPARK_DEVFN friend fp2_dev operator-(const fp2_dev& a, const fp2_dev& b)
    {   fp2_dev r; r.c0 = a.c0 - b.c0; r.c1 = a.c1 - b.c1; return r;   }
    SPPARK_DEVFN fp2_dev dbl() const { fp2_dev r; r.c0 = c0.dbl(); r.c1 = c1.dbl(); return r; }
    SPPARK_DEVFN fp2_dev neg() const { fp2_dev r; r.c0 = c0.neg(); r.c1 = c1.neg(); return r; }
    SPPARK_DEVFN fp2_dev cneg(bool flag) const { fp2_dev r; r.c0 = c0.cneg(flag); r.c1 = c1.cneg(flag); return r; }

    // x * NR for the small non-residues in use
    SPPARK_DEVFN static fp mul_nr(const fp& x)
    {
        if constexpr (P::FP2_NR == 1) return x;
        else { static_assert(P::FP2_NR == 5, "non-residue"); return x.dbl().dbl() + x; }
    }
    // (a0 + a1 u)(b0 + b1 u) = (a0 b0 - NR a1 b1) + ((a0 + a1)(b0 + b1) - a0 b0 - a1 b1) u
    SPPARK_DEVFN friend fp2_dev operator*(const fp2_dev& a, const fp2_dev& b)
    {
        fp t0 = a.c0 * b.c0, t1 = a.c1 * b.c1;
        fp2_dev r;
        r.c1 = (a.c0 + a.c1) * (b.c0 + b.c1) - t0 - t1;
        r.c0 = t0 - mul_nr(t1);
        return r;
    }
    // (a0 + a1 u)^2 = (a0^2 - NR a1^2) + 2 a0 a1 u, with
    // a0^2 - NR a1^2 = (a0 + a1)(a0 - NR a1) + (NR - 1) a0 a1: two base products for any NR
    SPPARK_DEVFN fp2_dev sqr() const
    {
        fp2_dev r;
        const fp m = c0 * c1;
        r.c1 = m.dbl();
        if constexpr (P::FP2_NR == 1) r.c0 = (c0 + c1) * (c0 - c1);
        else                          r.c0 = (c0 + c1) * (c0 - mul_nr(c1)) + m.dbl().dbl();     // NR - 1 = 4
        return r;
    }
};

} // namespace sppark_amd
