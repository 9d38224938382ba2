// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
//
// CPU restatement of the reference's short-Weierstrass point types for a4 = 0
// curves (BLS12-381 G1: y^2 = x^3 + 4, alt_bn128 G1: y^2 = x^3 + 3):
//
//   affine   <- ec/affine_t.hpp:17-62    infinity encoded as X = Y = 0
//   xyzz     <- ec/xyzz_t.hpp:14-548     x = X/ZZ, y = Y/ZZZ, inf: ZZZ = ZZ = 0
//                                        member order X, Y, ZZZ, ZZ (:17)
//   jacobian <- ec/jacobian_t.hpp:14-587 x = X/Z^2, y = Y/Z^3, inf: Z = 0
//
// The formulas are the EFD ones the reference cites (add-2008-s, madd-2008-s,
// mdbl-2008-s-1, dbl-2008-s-1, dbl-2009-l, add-2007-bl, madd-2007-bl) including
// its infinity / doubling / P-P twists; they are restated here, not included.
#pragma once
#include "ff.hpp"

namespace oracle {

template<class F> struct affine {
    F X, Y;
    bool is_inf() const { return X.is_zero() && Y.is_zero(); }
    void set_inf() { X.zero(); Y.zero(); }
};

template<class F> struct jacobian;

template<class F> struct xyzz {
    F X, Y, ZZZ, ZZ;

    bool is_inf() const { return ZZZ.is_zero() && ZZ.is_zero(); }
    void inf() { ZZZ.zero(); ZZ.zero(); }

    void set(const affine<F>& a)                        // ec/xyzz_t.hpp:67-74
    {   X = a.X; Y = a.Y; ZZZ = ZZ = F::one(a.is_inf());   }

    // general addition, ec/xyzz_t.hpp:117-200
    void add(const xyzz& q)
    {
        if (q.is_inf()) return;
        if (is_inf()) { *this = q; return; }

        F U1 = X * q.ZZ, S1 = Y * q.ZZZ;
        F Pd = q.X * ZZ - U1;                           // U2 - U1
        F Rd = q.Y * ZZZ - S1;                          // S2 - S1

        if (!Pd.is_zero()) {
            F PP = Pd ^ 2, PPP = Pd * PP, Q = U1 * PP;
            F X3 = (Rd ^ 2) - PPP - Q - Q;
            F Y3 = Rd * (Q - X3) - S1 * PPP;
            ZZ  = ZZ * PP * q.ZZ;
            ZZZ = ZZZ * PPP * q.ZZZ;
            X = X3; Y = Y3;
        } else if (Rd.is_zero()) {                      // same point: double
            F U = Y + Y, V = U ^ 2, W = U * V, S = X * V;
            F M = X ^ 2; M = M + M + M;
            F X3 = (M ^ 2) - S - S;
            F Y3 = M * (S - X3) - W * Y;
            ZZ = ZZ * V; ZZZ = ZZZ * W;
            X = X3; Y = Y3;
        } else {
            inf();                                      // P + (-P)
        }
    }

    // mixed addition with optional subtraction, ec/xyzz_t.hpp:351-429
    void add(const affine<F>& q, bool subtract = false)
    {
        if (q.is_inf()) return;
        if (is_inf()) {
            set(q);
            ZZZ.cneg(subtract);
            return;
        }
        F Rd = q.Y * ZZZ; Rd.cneg(subtract); Rd -= Y;   // S2 - Y1
        F Pd = q.X * ZZ - X;                            // U2 - X1

        if (!Pd.is_zero()) {
            F PP = Pd ^ 2, PPP = Pd * PP, Q = X * PP;
            F X3 = (Rd ^ 2) - PPP - Q - Q;
            F Y3 = Rd * (Q - X3) - Y * PPP;
            ZZ *= PP; ZZZ *= PPP;
            X = X3; Y = Y3;
        } else if (Rd.is_zero()) {                      // double the affine point
            F U = q.Y + q.Y;
            F V = U ^ 2, W = U * V, S = q.X * V;
            F M = q.X ^ 2; M = M + M + M;
            F X3 = (M ^ 2) - S - S;
            F Y3 = M * (S - X3) - W * q.Y;
            X = X3; Y = Y3; ZZ = V; ZZZ = W;
            ZZZ.cneg(subtract);
        } else {
            inf();
        }
    }

    affine<F> to_affine() const                         // ec/xyzz_t.hpp:77-85
    {
        affine<F> a;
        if (is_inf()) { a.set_inf(); return a; }
        F iy = ZZZ.reciprocal();                        // 1/Z^3
        F iz = iy * ZZ;                                 // 1/Z
        F ix = iz ^ 2;                                  // 1/Z^2
        a.X = X * ix; a.Y = Y * iy;
        return a;
    }

    jacobian<F> to_jacobian() const;                    // ec/xyzz_t.hpp:88-89
};

template<class F> struct jacobian {
    F X, Y, Z;

    bool is_inf() const { return Z.is_zero(); }
    void inf() { Z.zero(); }
    void set(const affine<F>& a) { X = a.X; Y = a.Y; Z = F::one(a.is_inf()); }

    void dbl()                                          // ec/jacobian_t.hpp:349-383
    {
        F A = X ^ 2, B = Y ^ 2, C = B ^ 2;
        F D = ((X + B) ^ 2) - A - C; D += D;
        F E = A + A + A;
        F X3 = (E ^ 2) - D - D;
        F Z3 = Z * Y; Z3 += Z3;
        F Y3 = E * (D - X3) - (C << 3);
        X = X3; Y = Y3; Z = Z3;
    }

    void add(const jacobian& q)                         // ec/jacobian_t.hpp:388-478
    {
        if (q.is_inf()) return;
        if (is_inf()) { *this = q; return; }

        F Z1Z1 = Z ^ 2, Z2Z2 = q.Z ^ 2;
        F S2 = q.Y * Z * Z1Z1, S1 = Y * q.Z * Z2Z2;
        F U1 = X * Z2Z2, H = q.X * Z1Z1 - U1;
        F r = S2 - S1;

        if (H.is_zero() && r.is_zero()) { dbl(); return; }

        F I = (H + H) ^ 2, J = H * I, V = U1 * I;
        r += r;
        F X3 = (r ^ 2) - J - V - V;
        F SJ = S1 * J;
        F Y3 = r * (V - X3) - SJ - SJ;
        F Z3 = (((Z + q.Z) ^ 2) - Z1Z1 - Z2Z2) * H;
        X = X3; Y = Y3; Z = Z3;
    }

    affine<F> to_affine() const                         // ec/jacobian_t.hpp:32-40
    {
        affine<F> a;
        if (is_inf()) { a.set_inf(); return a; }
        F iz = Z.reciprocal(), iz2 = iz ^ 2;
        a.X = X * iz2; a.Y = Y * iz2 * iz;
        return a;
    }

    friend bool operator==(const jacobian& p, const jacobian& q)    // :563-570
    {
        if (p.is_inf() || q.is_inf()) return p.is_inf() == q.is_inf();
        F Z1Z1 = p.Z ^ 2, Z2Z2 = q.Z ^ 2;
        return p.X * Z2Z2 == q.X * Z1Z1 && p.Y * Z2Z2 * q.Z == q.Y * Z1Z1 * p.Z;
    }
};

template<class F> jacobian<F> xyzz<F>::to_jacobian() const
{   jacobian<F> j; j.X = X * ZZ; j.Y = Y * ZZZ; j.Z = ZZ; return j;   }

// y^2 == x^3 + b ?  (b given in Montgomery form)
template<class F> static bool on_curve(const affine<F>& a, const F& b)
{
    if (a.is_inf()) return true;
    return (a.Y ^ 2) == (a.X ^ 2) * a.X + b;
}

} // namespace oracle
