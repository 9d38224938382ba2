// xyzz_dev / affine loading specialised for the loosely-reduced field (ff/montx_dev.hpp).
// Same formulas as ec/xyzz_dev.hpp (madd-2008-s, add-2008-s, dbl-2008-s-1, the ones
// ec/xyzz_t.hpp:117-200,351-429 uses); what changes is the bookkeeping:
//
//   value bounds, in multiples of p (rho = 2^RBITS / p ~ 2500; a product of a < ka*p and
//   b < kb*p is < (ka*kb/rho + 1)*p < 2p for every pair below), and limb sizes
//   ("n" = normalised, limbs < 2^LB; otherwise the stated multiple of 2^LB):
//
//     X   < 10p, limbs <= 5*2^LB        Y   < 5p, limbs <= 3*2^LB
//     ZZ, ZZZ  < 2p, n                  infinity: every limb of ZZ (and ZZZ) is zero
//     affine input coordinates: < 2p, n
//
//   29-bit limbs (F::TIGHT: alt_bn128 and Pasta G1 on NINE limbs, rho = 169 / 128): a lazy sum wraps at 8*2^LB instead of 16, the left
//   operand of a product may reach 4*2^LB instead of 8 and mul_add takes two left operands of <= 3*2^LB each.  X is
//   therefore kept NORMALISED (one norm() per new X3; in return it is subtracted with B = 1) and Y tighter:
//
//     X   < 10p, n                      Y   < 3p, limbs <= 2*2^LB          ZZ, ZZZ  < 2p, n
//
//   and every product of the formulas is still < 2p (the largest: P^2 with P < 13p, 169/169.3 + 1; Pasta: < 2.33p, which
//   only ever multiplies from the right).  Both sets of
//   bounds are machine-checked from the loosest admissible operands by tests/emu/emu_bounds.cpp.
//
//   These hold for every value written to memory, so any kernel can load any bucket.
//   sub<K, B>(a, b) = a + K*p - b needs b < (K-1)*p and b's limbs <= B*(2^LB - 1);
//   operator* needs its right operand n and its left operand's limbs < 2^31; sqr() needs n.
#pragma once
#include "../ff/montx_dev.hpp"
#include "xyzz_dev.hpp"

namespace sppark_amd {

// coordinate fields with their own point/bucket records (converted on the way in and out)
template<class FP> struct field_is_internal { static constexpr bool value = false; };
template<class P, int LB> struct field_is_internal<montx_dev<P, LB>> { static constexpr bool value = true; };
// ... of which the base field itself (G1): interleaved-pair point operations, gather prefetch, fixed-base tables
template<class FP> struct field_is_montx { static constexpr bool value = false; };
template<class P, int LB> struct field_is_montx<montx_dev<P, LB>> { static constexpr bool value = true; };

// Points converted once per MSM into X | Y internal limbs (2*NL words, 16-byte aligned
// stride); the infinity flag rides in bit 31 of X's top limb (a value < 2p leaves it free).
template<class P, int LB> struct affine_loader<montx_dev<P, LB>> {
    typedef montx_dev<P, LB> F;
    // record = X | Y limbs (2*NL words = 112 bytes for BLS12-381) padded to a power of two: a 128-byte
    // record at a 128-byte-aligned offset is exactly one L2 line / two 64-byte sectors per gather; at
    // its natural 112-byte stride a record straddles 2.5 sectors on average (measured:
    // profiles/r02_pmc_traffic.json)
    static constexpr unsigned RAW = 2 * F::NL * 4;
    static constexpr unsigned STRIDE = RAW <= 64 ? 64 : RAW <= 128 ? 128 : ((RAW + 15) / 16) * 16;
    template<bool FLAGGED>
    SPPARK_DEVFN static affine_dev<F> load(const unsigned char* base, size_t idx, unsigned)
    {
        const uint4* q = reinterpret_cast<const uint4*>(base + idx * (size_t)STRIDE);
        u32 w[(RAW + 15) / 16 * 4];
        #pragma unroll
        for (unsigned i = 0; i < (RAW + 15) / 16; i++) { uint4 v = q[i]; w[4*i] = v.x; w[4*i+1] = v.y; w[4*i+2] = v.z; w[4*i+3] = v.w; }
        affine_dev<F> a;
        a.X = F::from_wire(w); a.Y = F::from_wire(w + F::NL);
        a.inf = (a.X.l[F::NL - 1] >> 31) != 0;
        a.X.l[F::NL - 1] &= 0x7fffffffu;
        return a;
    }
    // one work item of the conversion pass: standard wire point -> internal record
    template<bool FLAGGED>
    SPPARK_DEVFN static void convert(unsigned char* dst, const unsigned char* src, size_t idx, unsigned stride)
    {
        typedef mont_dev<P> S;
        u32 wx[S::N], wy[S::N];
        bool inf;
        // The pass is bound by its memory accesses, not by the conversion (round 6: with the conversion a shift instead of two
        // products it ran exactly as long, 4.1 ms for 15 GB at 2^26 points).  A lane reads its point in 16-byte pieces where the
        // layout allows (plain points at a 16-byte-aligned base and stride: half the requests of the 8-byte loads that the
        // 104-byte flagged stride needs)
        if (!FLAGGED && (S::N % 2) == 0 && (stride & 15) == 0 && ((size_t)src & 15) == 0) {
            const uint4* q = reinterpret_cast<const uint4*>(src + idx * (size_t)stride);
            u32 w[2 * S::N];
            #pragma unroll
            for (int i = 0; i < S::N / 2; i++) { uint4 v = q[i]; w[4*i] = v.x; w[4*i+1] = v.y; w[4*i+2] = v.z; w[4*i+3] = v.w; }
            u32 any = 0;
            #pragma unroll
            for (int i = 0; i < S::N; i++) { wx[i] = w[i]; wy[i] = w[S::N + i]; any |= w[i] | w[S::N + i]; }
            inf = any == 0;
        } else {
            affine_dev<S> p = affine_loader<S>::template load<FLAGGED>(src, idx, stride);
            p.X.to_wire(wx); p.Y.to_wire(wy);
            inf = p.inf;
        }
        F x = F::from_std(wx), y = F::from_std(wy);
        u32 w[STRIDE / 4] = {};
        x.to_wire(w); y.to_wire(w + F::NL);
        if (inf) w[F::NL - 1] |= 0x80000000u;
        uint4* q = reinterpret_cast<uint4*>(dst + idx * (size_t)STRIDE);
        #pragma unroll
        for (unsigned i = 0; i < STRIDE / 16; i++) q[i] = make_uint4(w[4*i], w[4*i+1], w[4*i+2], w[4*i+3]);
    }
};

template<class P, int LB> struct xyzz_dev<montx_dev<P, LB>> {
    typedef montx_dev<P, LB> F;
    F X, Y, ZZZ, ZZ;
    // see the table at the top: with 29-bit limbs a fresh X3 is normalised and a stored X / Y has tighter limbs
    static constexpr bool TIGHT = F::TIGHT;
    static constexpr int BX = TIGHT ? 1 : 6;                // limb bound of X (and of a fresh X3) as a subtrahend
    static constexpr int BX2 = TIGHT ? 1 : 4;               // ... of the X3 of a doubling
    SPPARK_DEVFN static F keep_x(const F& x3) { if constexpr (TIGHT) return x3.norm(); else return x3; }
    SPPARK_DEVFN static F minus_y(const F& y) { if constexpr (TIGHT) return F::template neg<4, 2>(y); else return F::template neg<6, 4>(y); }

    SPPARK_DEVFN bool is_inf() const { return ZZ.limbs_all_zero(); }
    SPPARK_DEVFN void set_inf() { X = F::zero(); Y = F::zero(); ZZZ = F::zero(); ZZ = F::zero(); }

    SPPARK_DEVFN void set(const affine_dev<F>& p, bool negate)
    {
        if (p.inf) { set_inf(); return; }
        X = p.X; Y = negate ? F::template neg<3>(p.Y) : p.Y;           // < 3p, limbs <= 2*2^LB
        ZZZ = F::one(); ZZ = F::one();
    }

    SPPARK_DEVFN void madd(const affine_dev<F>& p, bool negate)
    {
        if (p.inf) return;
        if (is_inf()) { set(p, negate); return; }

        F U2, S2;
        F::template mul2<true, true>(U2, S2, p.X, ZZ, p.Y, ZZZ);   // all four operands n: < 2p, n (pairs of products are interleaved)
        if (negate) S2 = F::template neg<3>(S2);            // < 3p, limbs <= 2*2^LB
        F Pd = F::template sub<11, BX>(U2, X).norm();       // U2 - X      < 13p, n
        F Rd;                                               // +-S2 - Y    < 9p, n (TIGHT: < 7p)
        if constexpr (TIGHT) Rd = F::template sub<4, 2>(S2, Y).norm(); else Rd = F::template sub<6, 4>(S2, Y).norm();

        if (!Pd.template is_zero_mod<13>()) {               // fast path
            F PP, RR, PPP, Q;
            F::sqr2(PP, RR, Pd, Rd);                        // n, < 2p
            F::template mul2<true, false>(PPP, Q, Pd, PP, X, PP);  // Pd n; the left operand of the second one fat: allowed
            F T   = PPP + Q + Q;                            // < 6p, limbs <= 3*(2^LB - 1)
            F X3  = keep_x(F::template sub<8, 3>(RR, T));   // < 10p, limbs <= 5*2^LB (TIGHT: n)
            F D   = F::template sub<11, BX>(Q, X3);         // Q - X3      < 13p, limbs < 2^31 (TIGHT: <= 3*2^LB)
            // Y3 = R*(Q - X3) - Y1*PPP as ONE reduced sum of two products: D*Rd + (6p - Y)*PPP
            // (6p - Y: Y < 5p with limbs < 3*2^LB, negated against the fat 6p whose limbs are >= 4*2^LB - 4;
            // the result's limbs are < 5*2^LB)
            F nY  = minus_y(Y);                             // < 6p (TIGHT: 4p - Y < 4p, limbs <= 3*2^LB)
            Y   = F::mul_add(D, Rd, nY, PPP);               // < (13*9 + 6*2)p/rho + p < 2p, n
            F::template mul2<true, true>(ZZ, ZZZ, ZZ, PP, ZZZ, PPP);       // n x n
            X = X3;
        } else if (Rd.template is_zero_mod<9>()) {          // same point: 2*p
            F y2 = negate ? F::template neg<3>(p.Y).norm() : p.Y;       // n, < 3p
            dbl_affine(p.X, y2);
        } else {
            set_inf();
        }
    }

    SPPARK_DEVFN void add(const xyzz_dev& q)
    {
        if (q.is_inf()) return;
        if (is_inf()) { *this = q; return; }

        // (single product chains here: the full addition lives in the cold kernels, where the
        // register footprint of interleaved pairs costs an occupancy step)
        F U1 = X * q.ZZ;                                    // n, < 2p
        F S1 = Y * q.ZZZ;
        F U2 = q.X * ZZ;
        F S2 = q.Y * ZZZ;
        F Pd = F::template sub<3>(U2, U1).norm();           // < 5p, n
        F Rd = F::template sub<3>(S2, S1).norm();

        if (!Pd.template is_zero_mod<5>()) {
            F PP  = Pd.sqr();
            F PPP = Pd * PP;
            F Q   = U1 * PP;
            F T   = PPP + Q + Q;
            F X3  = keep_x(F::template sub<8, 3>(Rd.sqr(), T));
            F D   = F::template sub<11, BX>(Q, X3);
            Y   = F::mul_add(D, Rd, F::template neg<3>(S1), PPP);      // one reduction (see madd); n, < 2p
            ZZ  = (ZZ * PP) * q.ZZ;
            ZZZ = (ZZZ * PPP) * q.ZZZ;
            X = X3;
        } else if (Rd.template is_zero_mod<5>()) {
            // rare (equal points): one out-of-line copy, so that the nine products of a doubling
            // are not inlined into every addition of the cold kernels
            xyzz_dev t = *this;
            dbl_outlined(t);
            *this = t;
        } else {
            set_inf();
        }
    }

    // The same addition with its products in interleaved PAIRS (two multiply-add chains in flight, quotient
    // digits without masks where both operands are normalised): for the kernels at the top of the bucket sums,
    // where a handful of waves -- one per SIMD at most -- run chains of dependent additions and the latency of
    // one addition, not the register footprint, is what counts.
    SPPARK_DEVFN void add_pairs(const xyzz_dev& q)
    {
        if (q.is_inf()) return;
        if (is_inf()) { *this = q; return; }
        F U1, S1, U2, S2;
        F::mul2(U1, S1, X, q.ZZ, Y, q.ZZZ);                 // fat left operands; n, < 2p
        F::mul2(U2, S2, q.X, ZZ, q.Y, ZZZ);
        F Pd = F::template sub<3>(U2, U1).norm();           // < 5p, n
        F Rd = F::template sub<3>(S2, S1).norm();
        if (!Pd.template is_zero_mod<5>()) {
            F PP, RR, PPP, Q, t0, t1;
            F::sqr2(PP, RR, Pd, Rd);
            F::template mul2<true, true>(PPP, Q, Pd, PP, U1, PP);
            F T   = PPP + Q + Q;
            F X3  = keep_x(F::template sub<8, 3>(RR, T));
            F D   = F::template sub<11, BX>(Q, X3);
            Y   = F::mul_add(D, Rd, F::template neg<3>(S1), PPP);
            F::template mul2<true, true>(t0, t1, ZZ, PP, ZZZ, PPP);
            F::template mul2<true, true>(ZZ, ZZZ, t0, q.ZZ, t1, q.ZZZ);
            X = X3;
        } else if (Rd.template is_zero_mod<5>()) {
            xyzz_dev t = *this;
            dbl_outlined(t);
            *this = t;
        } else {
            set_inf();
        }
    }
    // ... and the doubling: (V, M) and (W, S) as pairs
    SPPARK_DEVFN void dbl_pairs()
    {
        if (is_inf()) return;
        F Yn = Y.norm(), Xn = X.norm();                     // n, < 5p / < 10p
        F U = (Yn + Yn).norm();                             // < 10p
        F V, M, W, S, M3s;
        F::sqr2(V, M, U, Xn);                               // < 2p
        F::template mul2<true, true>(W, S, U, V, Xn, V);
        F M3 = (M + M + M).norm();                          // < 6p
        M3s = M3.sqr();
        F X3 = keep_x(F::template sub<5, 2>(M3s, S + S));   // < 7p, limbs <= 4*2^LB (TIGHT: n)
        F D  = F::template sub<8, BX2>(S, X3);              // < 10p
        Y = F::mul_add(D, M3, F::template neg<3>(W), Yn);
        F::template mul2<true, true>(ZZ, ZZZ, ZZ, V, ZZZ, W);
        X = X3;
    }

#if defined(SPPARK_HOST_EMULATION)
    static void dbl_outlined(xyzz_dev& t) { t.dbl(); }
#else
    __device__ __noinline__ static void dbl_outlined(xyzz_dev& t) { t.dbl(); }
#endif

    SPPARK_DEVFN void dbl()
    {
        if (is_inf()) return;
        F Yn = Y.norm(), Xn = X.norm();                     // n, < 5p / < 10p
        F U = (Yn + Yn).norm();                             // < 10p
        F V = U.sqr();                                      // < 2p
        F W = U * V;
        F S = Xn * V;
        F M = Xn.sqr();
        F M3 = (M + M + M).norm();                          // < 6p
        F X3 = keep_x(F::template sub<5, 2>(M3.sqr(), S + S));     // < 7p, limbs <= 4*2^LB (TIGHT: n)
        F D  = F::template sub<8, BX2>(S, X3);              // < 10p
        Y = F::mul_add(D, M3, F::template neg<3>(W), Yn);
        ZZ = ZZ * V; ZZZ = ZZZ * W;
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

    // internal XYZZ -> the reference's image in the standard wire form (ec/xyzz_t.hpp:17),
    // canonical coordinates; infinity stays all-zero
    SPPARK_DEVFN void store_std(xyzz_mem<P::N>* dst) const
    {
        constexpr int N = P::N;
        u32 s[4 * N];
        if (is_inf()) { for (int i = 0; i < 4 * N; i++) s[i] = 0; }
        else { X.to_std(s); Y.to_std(s + N); ZZZ.to_std(s + 2 * N); ZZ.to_std(s + 3 * N); }
        uint4* d = reinterpret_cast<uint4*>(dst);
        #pragma unroll
        for (int i = 0; i < N; i++) d[i] = make_uint4(s[4*i], s[4*i+1], s[4*i+2], s[4*i+3]);
    }

private:
    // this = 2 * (x, y)   (mdbl-2008-s-1); x, y n with x < 2p, y < 3p
    SPPARK_DEVFN void dbl_affine(const F& x, const F& y)
    {
        F U = (y + y).norm();                               // < 6p
        F V = U.sqr();
        F W = U * V;
        F S = x * V;
        F M = x.sqr();
        F M3 = (M + M + M).norm();
        F X3 = keep_x(F::template sub<5, 2>(M3.sqr(), S + S));
        F D  = F::template sub<8, BX2>(S, X3);
        Y = F::mul_add(D, M3, F::template neg<3>(W), y);
        X = X3; ZZ = V; ZZZ = W;
    }
};

} // namespace sppark_amd
