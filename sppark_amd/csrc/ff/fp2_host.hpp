// Host-side Fp2 = Fp[u]/(u^2 + NR), NR = P::FP2_NR (1, or 5 for BLS12-377), for the G2 MSM tail (window sums -> Horner) and the
// multi-GPU combine; same scope note as mont_host.hpp.  Memory image c0 | c1.
#pragma once
#include "mont_host.hpp"

namespace sppark_amd {

template<class P> struct fp2_host {
    typedef mont_host<P> fp;
    fp c0, c1;

    static fp2_host zero() { return fp2_host{fp::zero(), fp::zero()}; }
    static fp2_host one()  { return fp2_host{fp::one(), fp::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    friend bool operator==(const fp2_host& a, const fp2_host& b) { return a.c0 == b.c0 && a.c1 == b.c1; }

    friend fp2_host operator+(const fp2_host& a, const fp2_host& b) { return fp2_host{a.c0 + b.c0, a.c1 + b.c1}; }
    friend fp2_host operator-(const fp2_host& a, const fp2_host& b) { return fp2_host{a.c0 - b.c0, a.c1 - b.c1}; }
    static fp mul_nr(const fp& x)
    {
        fp r = x;
        for (unsigned k = 1; k < P::FP2_NR; k++) r = r + x;
        return r;
    }
    friend fp2_host operator*(const fp2_host& a, const fp2_host& b)
    {
        fp t0 = a.c0 * b.c0, t1 = a.c1 * b.c1;
        return fp2_host{t0 - mul_nr(t1), (a.c0 + a.c1) * (b.c0 + b.c1) - t0 - t1};
    }
    fp2_host sqr() const { return fp2_host{c0.sqr() - mul_nr(c1.sqr()), (c0 * c1).dbl()}; }
    fp2_host dbl() const { return fp2_host{c0.dbl(), c1.dbl()}; }
    fp2_host neg() const { return fp2_host{c0.neg(), c1.neg()}; }
    // 1/(a0 + a1 u) = (a0 - a1 u) / (a0^2 + NR a1^2); 1/0 = 0
    fp2_host inverse() const
    {
        fp n = (c0.sqr() + mul_nr(c1.sqr())).inverse();
        return fp2_host{c0 * n, (c1 * n).neg()};
    }
};

} // namespace sppark_amd
