// Host-side Jacobian point for the MSM tail (see ff/mont_host.hpp for scope).
// Memory image X|Y|Z == the reference's result type (ec/jacobian_t.hpp:17), i.e.
// what blst_p1 / ark G1Projective callers pass as `out`.
#pragma once
#include "../ff/mont_host.hpp"

namespace sppark_amd {

template<class F> struct jacobian_host {
    F X, Y, Z;

    bool is_inf() const { return Z.is_zero(); }
    void set_inf() { X = F::zero(); Y = F::zero(); Z = F::zero(); }

    // from the device bucket type (X, Y, ZZZ, ZZ): (X*ZZ, Y*ZZZ, ZZ)
    static jacobian_host from_xyzz(const F& X, const F& Y, const F& ZZZ, const F& ZZ)
    {
        jacobian_host r;
        if (ZZZ.is_zero() && ZZ.is_zero()) { r.set_inf(); return r; }
        r.X = X * ZZ; r.Y = Y * ZZZ; r.Z = ZZ;
        return r;
    }

    void dbl()                              // dbl-2009-l, a = 0
    {
        if (is_inf()) return;
        F A = X.sqr(), B = Y.sqr(), C = B.sqr();
        F D = (X + B).sqr() - A - C; D = D.dbl();
        F E = A + A + A;
        F X3 = E.sqr() - D - D;
        F Z3 = (Y * Z).dbl();
        F Y3 = E * (D - X3) - C.dbl().dbl().dbl();
        X = X3; Y = Y3; Z = Z3;
    }

    void add(const jacobian_host& q)        // add-2007-bl with the usual special cases
    {
        if (q.is_inf()) return;
        if (is_inf()) { *this = q; return; }
        F Z1Z1 = Z.sqr(), Z2Z2 = q.Z.sqr();
        F U1 = X * Z2Z2, U2 = q.X * Z1Z1;
        F S1 = Y * q.Z * Z2Z2, S2 = q.Y * Z * Z1Z1;
        F H = U2 - U1, r = S2 - S1;
        if (H.is_zero()) {
            if (r.is_zero()) dbl(); else set_inf();
            return;
        }
        F I = H.dbl().sqr(), J = H * I, V = U1 * I;
        r = r.dbl();
        F X3 = r.sqr() - J - V - V;
        F Y3 = r * (V - X3) - (S1 * J).dbl();
        F Z3 = ((Z + q.Z).sqr() - Z1Z1 - Z2Z2) * H;
        X = X3; Y = Y3; Z = Z3;
    }

    // affine (x, y); infinity -> (0, 0)
    void to_affine(F& x, F& y) const
    {
        if (is_inf()) { x = F::zero(); y = F::zero(); return; }
        F iz = Z.inverse(), iz2 = iz.sqr();
        x = X * iz2; y = Y * iz2 * iz;
    }
};

} // namespace sppark_amd
