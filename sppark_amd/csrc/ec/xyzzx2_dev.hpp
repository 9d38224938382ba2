// XYZZ buckets and affine records of the G2 pipeline over fp2x_dev (ff/fp2x_dev.hpp): the formulas of ec/xyzz_dev.hpp
// (EFD madd-2008-s / add-2008-s / dbl-2008-s-1, the ones ec/xyzz_t.hpp:117-200,351-429 uses) written against the
// loosely-reduced field's contract.
//
// INVARIANT of a bucket between operations: all four coordinates NORMALISED (limbs < 2^LB), X < 9 p, Y < 5 p,
// ZZ, ZZZ < 2 p (products).  Affine records: X, Y normalised, < 2 p.  Every value that becomes an operand of a product
// is normalised first -- norm() is 3 instructions per limb against ~1300 for the product -- and its bound is written at
// the use as the template argument of mul / sqr (value < (K - 1) p) or of sub / neg (subtrahend < (K - 1) p).
#pragma once
#include "xyzzx_dev.hpp"
#include "../ff/fp2_dev.hpp"
#include "../ff/fp2x_dev.hpp"

namespace sppark_amd {

template<class P, int LB> struct field_is_internal<fp2x_dev<P, LB>> { static constexpr bool value = true; };

// Points converted once per MSM into X.c0 | X.c1 | Y.c0 | Y.c1 internal limbs, padded to whole 64-byte sectors; the
// infinity flag rides in bit 31 of X.c0's top limb.
template<class P, int LB> struct affine_loader<fp2x_dev<P, LB>> {
    typedef fp2x_dev<P, LB> F;
    static constexpr unsigned RAW = 2 * F::N * 4;                   // 224 bytes for BLS12-381, 160 for alt_bn128
    static constexpr unsigned STRIDE = ((RAW + 63) / 64) * 64;
    template<bool FLAGGED>
    SPPARK_DEVFN static affine_dev<F> load(const unsigned char* base, size_t idx, unsigned)
    {
        const uint4* q = reinterpret_cast<const uint4*>(base + idx * (size_t)STRIDE);
        u32 w[RAW / 4];
        #pragma unroll
        for (unsigned i = 0; i < RAW / 16; i++) { uint4 v = q[i]; w[4*i] = v.x; w[4*i+1] = v.y; w[4*i+2] = v.z; w[4*i+3] = v.w; }
        affine_dev<F> a;
        a.X = F::from_wire(w); a.Y = F::from_wire(w + F::N);
        a.inf = (a.X.c0.l[F::NL - 1] >> 31) != 0;
        a.X.c0.l[F::NL - 1] &= 0x7fffffffu;
        return a;
    }
    // one work item of the conversion pass: standard wire point -> internal record
    template<bool FLAGGED>
    SPPARK_DEVFN static void convert(unsigned char* dst, const unsigned char* src, size_t idx, unsigned stride)
    {
        typedef fp2_dev<P> S;
        affine_dev<S> p = affine_loader<S>::template load<FLAGGED>(src, idx, stride);
        u32 wx[S::N], wy[S::N];
        p.X.to_wire(wx); p.Y.to_wire(wy);
        F x = F::from_std(wx), y = F::from_std(wy);
        u32 w[STRIDE / 4] = {};
        x.to_wire(w); y.to_wire(w + F::N);
        if (p.inf) w[F::NL - 1] |= 0x80000000u;
        uint4* q = reinterpret_cast<uint4*>(dst + idx * (size_t)STRIDE);
        #pragma unroll
        for (unsigned i = 0; i < STRIDE / 16; i++) q[i] = make_uint4(w[4*i], w[4*i+1], w[4*i+2], w[4*i+3]);
    }
};

template<class P, int LB> struct xyzz_dev<fp2x_dev<P, LB>> {
    typedef fp2x_dev<P, LB> F;
    F X, Y, ZZZ, ZZ;
    static constexpr int KX = 10, KY = 6;                           // X < (KX - 1) p, Y < (KY - 1) p

    SPPARK_DEVFN bool is_inf() const { return ZZ.limbs_all_zero(); }
    SPPARK_DEVFN void set_inf() { X = F::zero(); Y = F::zero(); ZZZ = F::zero(); ZZ = F::zero(); }

    SPPARK_DEVFN void set(const affine_dev<F>& p, bool negate)
    {
        if (p.inf) { set_inf(); return; }
        X = p.X; Y = negate ? F::template neg<3>(p.Y).norm() : p.Y;         // < 3 p, n
        ZZZ = F::one(); ZZ = F::one();
    }

    SPPARK_DEVFN void madd(const affine_dev<F>& p, bool negate)
    {
        if (p.inf) return;
        if (is_inf()) { set(p, negate); return; }

        F U2 = F::template mul<3>(p.X, ZZ);                         // < 2 p, n
        F S2 = F::template mul<3>(p.Y, ZZZ);
        if (negate) S2 = F::template neg<3>(S2).norm();             // < 3 p
        const F Pd = F::template sub<KX>(U2, X).norm();             // U2 - X      < 12 p
        const F Rd = F::template sub<KY>(S2, Y).norm();             // +-S2 - Y    < 9 p

        if (!Pd.template is_zero_mod<12>()) {                       // fast path
            const F PP  = Pd.template sqr<13>();                    // < 2 p
            const F RR  = Rd.template sqr<10>();
            const F PPP = F::template mul<13>(Pd, PP);
            const F Q   = F::template mul<KX>(X, PP);
            const F T   = PPP + Q + Q;                              // < 6 p, limbs < 3 * 2^LB
            const F X3  = F::template sub<7, 3>(RR, T).norm();      // < 9 p
            const F D   = F::template sub<10>(Q, X3).norm();        // Q - X3      < 12 p
            Y   = F::template sub<3>(F::template mul<13>(D, Rd), F::template mul<KY>(Y, PPP)).norm();       // < 5 p
            ZZ  = F::template mul<3>(ZZ, PP);
            ZZZ = F::template mul<3>(ZZZ, PPP);
            X = X3;
        } else if (Rd.template is_zero_mod<9>()) {                  // same point: 2 * p
            dbl_affine(p.X, negate ? F::template neg<3>(p.Y).norm() : p.Y);
        } else {
            set_inf();
        }
    }

    SPPARK_DEVFN void add(const xyzz_dev& q)
    {
        if (q.is_inf()) return;
        if (is_inf()) { *this = q; return; }

        const F U1 = F::template mul<KX>(X, q.ZZ);                  // < 2 p, n
        const F S1 = F::template mul<KY>(Y, q.ZZZ);
        const F U2 = F::template mul<KX>(q.X, ZZ);
        const F S2 = F::template mul<KY>(q.Y, ZZZ);
        const F Pd = F::template sub<3>(U2, U1).norm();             // < 5 p
        const F Rd = F::template sub<3>(S2, S1).norm();

        if (!Pd.template is_zero_mod<5>()) {
            const F PP  = Pd.template sqr<6>();
            const F PPP = F::template mul<6>(Pd, PP);
            const F Q   = F::template mul<3>(U1, PP);
            const F T   = PPP + Q + Q;
            const F X3  = F::template sub<7, 3>(Rd.template sqr<6>(), T).norm();    // < 9 p
            const F D   = F::template sub<10>(Q, X3).norm();                        // < 12 p
            Y   = F::template sub<3>(F::template mul<13>(D, Rd), F::template mul<3>(S1, PPP)).norm();   // < 5 p
            ZZ  = F::template mul<3>(F::template mul<3>(ZZ, PP), q.ZZ);
            ZZZ = F::template mul<3>(F::template mul<3>(ZZZ, PPP), q.ZZZ);
            X = X3;
        } else if (Rd.template is_zero_mod<5>()) {
            xyzz_dev t = *this;                                     // rare (equal points): one out-of-line copy
            dbl_outlined(t);
            *this = t;
        } else {
            set_inf();
        }
    }
    // (the interleaved-pair forms of the G1 class are what the low-latency kernels are built on; G2 runs the plain ones)
    SPPARK_DEVFN void add_pairs(const xyzz_dev& q) { add(q); }
    SPPARK_DEVFN void dbl_pairs() { dbl(); }

#if defined(SPPARK_HOST_EMULATION)
    static void dbl_outlined(xyzz_dev& t) { t.dbl(); }
#else
    __device__ __noinline__ static void dbl_outlined(xyzz_dev& t) { t.dbl(); }
#endif

    SPPARK_DEVFN void dbl()
    {
        if (is_inf()) return;
        const F U  = (Y + Y).norm();                                // < 10 p
        const F V  = U.template sqr<11>();                          // < 2 p
        const F W  = F::template mul<11>(U, V);
        const F S  = F::template mul<KX>(X, V);
        const F M  = X.template sqr<KX>();
        const F M3 = (M + M + M).norm();                            // < 6 p
        const F X3 = F::template sub<5, 2>(M3.template sqr<7>(), S + S).norm();     // < 7 p
        const F D  = F::template sub<8>(S, X3).norm();              // < 10 p
        Y = F::template sub<3>(F::template mul<11>(D, M3), F::template mul<3>(W, Y)).norm();     // < 5 p
        ZZ = F::template mul<3>(ZZ, V); ZZZ = F::template mul<3>(ZZZ, W);
        X = X3;
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

    // internal XYZZ -> the reference's image in the standard wire form (ec/xyzz_t.hpp:17 over fp2), canonical
    // coordinates; infinity stays all-zero
    SPPARK_DEVFN void store_std(xyzz_mem<F::NW>* dst) const
    {
        constexpr int N = F::NW;
        u32 s[4 * N];
        if (is_inf()) { for (int i = 0; i < 4 * N; i++) s[i] = 0; }
        else { X.to_std(s); Y.to_std(s + N); ZZZ.to_std(s + 2 * N); ZZ.to_std(s + 3 * N); }
        uint4* d = reinterpret_cast<uint4*>(dst);
        #pragma unroll
        for (int i = 0; i < N; i++) d[i] = make_uint4(s[4*i], s[4*i+1], s[4*i+2], s[4*i+3]);
    }

private:
    // this = 2 * (x, y)   (mdbl-2008-s-1); x, y n with x < 2 p, y < 3 p
    SPPARK_DEVFN void dbl_affine(const F& x, const F& y)
    {
        const F U  = (y + y).norm();                                // < 6 p
        const F V  = U.template sqr<7>();
        const F W  = F::template mul<7>(U, V);
        const F S  = F::template mul<3>(x, V);
        const F M  = x.template sqr<3>();
        const F M3 = (M + M + M).norm();                            // < 6 p
        const F X3 = F::template sub<5, 2>(M3.template sqr<7>(), S + S).norm();     // < 7 p
        const F D  = F::template sub<8>(S, X3).norm();              // < 10 p
        Y = F::template sub<3>(F::template mul<11>(D, M3), F::template mul<3>(W, y)).norm();     // < 5 p
        X = X3; ZZ = V; ZZZ = W;
    }
};

} // namespace sppark_amd
