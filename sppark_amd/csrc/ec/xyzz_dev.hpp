// Device-side point types for the MSM bucket pipeline (a4 = 0 curves).
//
//   affine_dev : X, Y  (+ in-register infinity flag).  Memory formats accepted:
//                plain  X|Y, infinity = all-zero          (ec/affine_t.hpp:17-35)
//                flagged X|Y|flag byte, stride = ffi size (ec/affine_t.hpp:64-122,
//                the arkworks G1Affine layout poc/msm-cuda/src/lib.rs:52-58 passes)
//   xyzz_dev   : X, Y, ZZZ, ZZ with x = X/ZZ, y = Y/ZZZ, infinity = (ZZZ = ZZ = 0);
//                same member order and memory image as the reference's bucket
//                type (ec/xyzz_t.hpp:17) so bucket dumps are interchangeable.
//
// Formulas: EFD madd-2008-s / mdbl-2008-s-1 (mixed) and add-2008-s /
// dbl-2008-s-1 (full), the ones ec/xyzz_t.hpp:117-200,351-429 uses.  Control
// flow is written for wave64: the generic-add path is the straight-line fast
// path every lane normally takes; "same point" (doubling), "opposite point"
// and "operand at infinity" are handled by wave-divergent side branches that
// cost nothing unless some lane actually needs them.
#pragma once
#include "../ff/mont_dev.hpp"

namespace sppark_amd {

template<class F> struct affine_dev {
    F X, Y;
    bool inf;
};

// Load one affine point.  |stride| bytes between points; FLAGGED selects the
// Affine_inf_t wire format.  Limbs are fetched as 8-byte words so that the
// 104-byte arkworks stride (8-byte aligned only) is legal.  affine_loader<F> is
// the customisation point (ec/xyzzx_dev.hpp loads pre-converted records instead).
template<class F> struct affine_loader {
    template<bool FLAGGED>
    SPPARK_DEVFN static affine_dev<F> load(const unsigned char* base, size_t idx, unsigned stride)
    {
        constexpr int N = F::N;
        const unsigned char* p = base + idx * (size_t)stride;
        const uint2* q = reinterpret_cast<const uint2*>(p);
        u32 wx[N], wy[N];
        #pragma unroll
        for (int i = 0; i < N / 2; i++) { uint2 w = q[i]; wx[2*i] = w.x; wx[2*i+1] = w.y; }
        #pragma unroll
        for (int i = 0; i < N / 2; i++) { uint2 w = q[N/2 + i]; wy[2*i] = w.x; wy[2*i+1] = w.y; }
        affine_dev<F> a;
        a.X = F::from_wire(wx); a.Y = F::from_wire(wy);
        if (FLAGGED) a.inf = (p[2 * N * 4] & 1) != 0;
        else         a.inf = a.X.is_zero() & a.Y.is_zero();
        return a;
    }
};
template<class F, bool FLAGGED>
SPPARK_DEVFN affine_dev<F> load_affine(const unsigned char* base, size_t idx, unsigned stride)
{   return affine_loader<F>::template load<FLAGGED>(base, idx, stride);   }

// Memory image of a bucket: X | Y | ZZZ | ZZ in the wire format (32-bit limbs),
// i.e. the reference's xyzz_t layout (ec/xyzz_t.hpp:17), whatever limb size the
// register type uses.
template<int N> struct alignas(16) xyzz_mem { u32 w[4 * N]; };

template<class F> struct xyzz_dev {
    F X, Y, ZZZ, ZZ;

    SPPARK_DEVFN bool is_inf() const { return ZZZ.is_zero() & ZZ.is_zero(); }
    SPPARK_DEVFN void set_inf()
    {   X = F::zero(); Y = F::zero(); ZZZ = F::zero(); ZZ = F::zero();   }

    // this = +/- affine point
    SPPARK_DEVFN void set(const affine_dev<F>& p, bool negate)
    {
        if (p.inf) { set_inf(); return; }
        X = p.X; Y = p.Y.cneg(negate); ZZZ = F::one(); ZZ = F::one();
    }

    // this += (negate ? -p : p), mixed addition 8M + 2S on the fast path.
    SPPARK_DEVFN void madd(const affine_dev<F>& p, bool negate)
    {
        if (p.inf) return;
        F y2 = p.Y.cneg(negate);
        if (is_inf()) { X = p.X; Y = y2; ZZZ = F::one(); ZZ = F::one(); return; }

        F Pd = p.X * ZZ - X;                    // U2 - X1
        F Rd = y2 * ZZZ - Y;                    // S2 - Y1

        if (!Pd.is_zero()) {                    // fast path
            F PP  = Pd.sqr();
            F PPP = Pd * PP;
            F Q   = X * PP;
            F X3  = Rd.sqr() - PPP - Q - Q;
            F Y3  = Rd * (Q - X3) - Y * PPP;
            ZZ  = ZZ * PP;
            ZZZ = ZZZ * PPP;
            X = X3; Y = Y3;
        } else if (Rd.is_zero()) {              // same point: 2*p
            F U = y2.dbl();
            F V = U.sqr();
            F W = U * V;
            F S = p.X * V;
            F M = p.X.sqr(); M = M + M + M;
            F X3 = M.sqr() - S - S;
            F Y3 = M * (S - X3) - W * y2;
            X = X3; Y = Y3; ZZ = V; ZZZ = W;
        } else {                                // p + (-p)
            set_inf();
        }
    }

    // this += q, full addition 12M + 2S on the fast path.
    SPPARK_DEVFN void add(const xyzz_dev& q)
    {
        if (q.is_inf()) return;
        if (is_inf()) { *this = q; return; }

        F U1 = X * q.ZZ;
        F S1 = Y * q.ZZZ;
        F Pd = q.X * ZZ - U1;
        F Rd = q.Y * ZZZ - S1;

        if (!Pd.is_zero()) {
            F PP  = Pd.sqr();
            F PPP = Pd * PP;
            F Q   = U1 * PP;
            F X3  = Rd.sqr() - PPP - Q - Q;
            F Y3  = Rd * (Q - X3) - S1 * PPP;
            ZZ  = ZZ * PP * q.ZZ;
            ZZZ = ZZZ * PPP * q.ZZZ;
            X = X3; Y = Y3;
        } else if (Rd.is_zero()) {
            dbl();
        } else {
            set_inf();
        }
    }

    // this = 2*this (dbl-2008-s-1)
    SPPARK_DEVFN void dbl()
    {
        if (is_inf()) return;
        F U = Y.dbl();
        F V = U.sqr();
        F W = U * V;
        F S = X * V;
        F M = X.sqr(); M = M + M + M;
        F X3 = M.sqr() - S - S;
        F Y3 = M * (S - X3) - W * Y;
        ZZ = ZZ * V; ZZZ = ZZZ * W;
        X = X3; Y = Y3;
    }

    typedef xyzz_mem<F::N> mem_t;

    SPPARK_DEVFN void store(mem_t* dst) const
    {
        constexpr int N = F::N;
        u32 s[4 * N];
        X.to_wire(s); Y.to_wire(s + N); ZZZ.to_wire(s + 2 * N); ZZ.to_wire(s + 3 * N);
        uint4* d = reinterpret_cast<uint4*>(dst);
        #pragma unroll
        for (int i = 0; i < N; i++) d[i] = make_uint4(s[4*i], s[4*i+1], s[4*i+2], s[4*i+3]);
    }
    SPPARK_DEVFN static xyzz_dev load(const mem_t* src)
    {
        constexpr int N = F::N;
        u32 d[4 * N];
        const uint4* q = reinterpret_cast<const uint4*>(src);
        #pragma unroll
        for (int i = 0; i < N; i++) { uint4 w = q[i]; d[4*i] = w.x; d[4*i+1] = w.y; d[4*i+2] = w.z; d[4*i+3] = w.w; }
        xyzz_dev r;
        r.X = F::from_wire(d); r.Y = F::from_wire(d + N); r.ZZZ = F::from_wire(d + 2 * N); r.ZZ = F::from_wire(d + 3 * N);
        return r;
    }
};

} // namespace sppark_amd
