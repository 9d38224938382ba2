// G2 buckets with ONE Fp2 COMPONENT PER WAVE: two waves of a 128-lane work-group carry one vector of 64 mixed additions.
// (Written in round 4 on the host emulation, measured and adopted in round 5 for the 14-limb base fields:
// BLS12-381 G2 2^22 47.8 -> 40.2 ms, 2^20 15.3 -> 14.0 ms; the 10-limb alt_bn128 loses 9 % and keeps the one-lane kernel;
// profiles/r05_g2_coop_ab.log.)
//
// Why.  The G2 accumulation over fp2x_dev holds a whole Fp2 bucket, an Fp2 point and the temporaries of a mixed addition
// in one lane: 414 registers, ONE wave per SIMD, 0.164 vector instructions per clock where the two-wave G1 kernel issues
// 0.25 (profiles/r04_msm_g2.log; two attempts at two waves by register caps spilled their gain away, DESIGN_HISTORY.md
// section G).  An Fp2 product is two INDEPENDENT sums of two base products with one reduction each --
//     c0 = a0 b0 + (K p - a1)(NR b1)        c1 = a0 b1 + a1 b0        (ff/fp2x_dev.hpp: montx_dev::mul_add)
// -- exactly one per component.  So wave r of a pair holds component r of every coordinate (the register set of the G1
// kernel: half the state), computes component r of every product, and reads the other wave's operand components from LDS.
// No arithmetic is duplicated; what is added is 13 + 15 fourteen-limb vectors through LDS and eight two-wave barriers per
// mixed addition of ~7500 vector instructions per wave.  The reference splits an Fp2 element over two LANES with shuffles
// (ff/bls12-381-fp2.hpp:25-150): the same cut, one level up, because a wave -- not a lane -- is what gets an issue slot.
//
// Contract: all 128 lanes of the work-group call every operation (barriers inside); role = tid >> 6 is the component,
// lane = tid & 63 the addition; both waves see the same control flow (they walk the same entries).  Value / limb bounds
// are those of ec/xyzzx2_dev.hpp, component by component.
#pragma once
#include "xyzzx2_dev.hpp"
#include "xyzz_coop.hpp"      // coop_barrier / coop_any and their host-emulation hooks

namespace sppark_amd {

// exchange area of one wave pair: five slots of [component][limb][lane] words (limb-major, lane-contiguous: conflict-free)
// and one flag word per component and lane
template<class F2> struct g2c_lds {
    u32 w[5][2][F2::NL][64]; u32 flag[2][64];
    SPPARK_BND(double bv[5][2][64]; double bl[5][2][64];)       // (host bound tracking: the claims travel with the limbs)
};

template<class F2> struct g2c_ctx {
    typedef typename F2::fp fp;
    g2c_lds<F2>* ex;
    unsigned role, lane;
    SPPARK_DEVFN void put(unsigned slot, const fp& v) const
    {
        #pragma unroll
        for (int j = 0; j < F2::NL; j++) ex->w[slot][role][j][lane] = v.l[j];
        SPPARK_BND(ex->bv[slot][role][lane] = v.bv; ex->bl[slot][role][lane] = v.bl;)
    }
    SPPARK_DEVFN fp other(unsigned slot) const
    {
        fp r;
        #pragma unroll
        for (int j = 0; j < F2::NL; j++) r.l[j] = ex->w[slot][role ^ 1][j][lane];
        SPPARK_BND(r.bnd_set(ex->bv[slot][role ^ 1][lane], ex->bl[slot][role ^ 1][lane]);)
        return r;
    }
    SPPARK_DEVFN void put_flag(u32 f) const { ex->flag[role][lane] = f; }
    SPPARK_DEVFN u32 other_flag() const { return ex->flag[role ^ 1][lane]; }
    // every write phase starts after the previous read phase of BOTH waves has ended, every read phase after the write
    // phase: open() ... put() ... shut() ... other()
    SPPARK_DEVFN void open() const { coop_barrier(); }
    SPPARK_DEVFN void shut() const { coop_barrier(); }
};

// component |role| of a * b from this wave's components (a, b) and the other wave's (ao, bo); contract of
// fp2x_dev::mul<KA>: operands normalised, a < (KA - 1) p
template<class F2, int KA, unsigned ROLE>
SPPARK_DEVFN typename F2::fp g2c_mul(const typename F2::fp& a, const typename F2::fp& ao,
                                     const typename F2::fp& b, const typename F2::fp& bo)
{
    typedef typename F2::fp fp;
    if constexpr (ROLE == 0) return fp::mul_add(a, b, fp::template neg<KA, 1>(ao), F2::mul_nr(bo));    // a0 b0 - NR a1 b1
    else                     return fp::mul_add(ao, b, a, bo);                                          // a0 b1 + a1 b0
}
// component |role| of c^2 (fp2x_dev::sqr<KA>: (c0 + c1)(c0 - c1) | 2 c0 c1 when u^2 = -1)
template<class F2, int KA, unsigned ROLE>
SPPARK_DEVFN typename F2::fp g2c_sqr(const typename F2::fp& c, const typename F2::fp& co)
{
    typedef typename F2::fp fp;
    if constexpr (F2::FP2_NR == 1) {
        if constexpr (ROLE == 0) return (c + co) * fp::template sub<KA, 1>(c, co).norm();
        else                     return (co + co) * c;
    } else {
        return g2c_mul<F2, KA, ROLE>(c, co, c, co);
    }
}

// this wave's components of an affine point of the converted records (ec/xyzzx2_dev.hpp: X.c0 | X.c1 | Y.c0 | Y.c1
// internal limbs, the infinity flag in bit 31 of X.c0's top limb)
template<class F2> struct g2c_affine {
    typedef typename F2::fp fp;
    fp X, Y;
    bool inf;
    SPPARK_DEVFN static g2c_affine load(const unsigned char* base, size_t idx, unsigned role)
    {
        constexpr int NL = F2::NL;
        static_assert(NL % 2 == 0, "components are read in 8-byte pieces");
        const u32* rec = reinterpret_cast<const u32*>(base + idx * (size_t)affine_loader<F2>::STRIDE);
        g2c_affine a;
        u32 wx[NL], wy[NL];
        const uint2* qx = reinterpret_cast<const uint2*>(rec + role * NL);
        const uint2* qy = reinterpret_cast<const uint2*>(rec + 2 * NL + role * NL);
        #pragma unroll
        for (int i = 0; i < NL / 2; i++) { uint2 v = qx[i]; wx[2*i] = v.x; wx[2*i+1] = v.y; uint2 u = qy[i]; wy[2*i] = u.x; wy[2*i+1] = u.y; }
        const u32 top0 = role == 0 ? wx[NL - 1] : rec[NL - 1];          // X.c0's top limb carries the flag
        a.inf = (top0 >> 31) != 0;
        if (role == 0) wx[NL - 1] &= 0x7fffffffu;
        a.X = fp::from_wire(wx); a.Y = fp::from_wire(wy);
        return a;
    }
    SPPARK_DEVFN static g2c_affine infinity() { g2c_affine a; a.X = fp::zero(); a.Y = fp::zero(); a.inf = true; return a; }
};

// this wave's components of an XYZZ bucket
template<class F2> struct g2c_bucket {
    typedef typename F2::fp fp;
    typedef xyzz_mem<F2::N> mem_t;
    fp X, Y, ZZZ, ZZ;
    static constexpr int KX = 10, KY = 6;                           // as xyzz_dev<fp2x_dev>: X < (KX - 1) p, Y < (KY - 1) p

    SPPARK_DEVFN void set_inf() { X = fp::zero(); Y = fp::zero(); ZZZ = fp::zero(); ZZ = fp::zero(); }
    SPPARK_DEVFN static fp one_component(unsigned role) { return role == 0 ? fp::one() : fp::zero(); }

    // memory image: X | Y | ZZZ | ZZ, each c0 | c1 limbs -- this wave's NL words of every coordinate
    SPPARK_DEVFN void store(mem_t* dst, unsigned role) const
    {
        constexpr int NL = F2::NL;
        u32* d = reinterpret_cast<u32*>(dst);
        const fp* co[4] = {&X, &Y, &ZZZ, &ZZ};
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            uint2* q = reinterpret_cast<uint2*>(d + k * 2 * NL + role * NL);
            #pragma unroll
            for (int i = 0; i < NL / 2; i++) q[i] = make_uint2(co[k]->l[2*i], co[k]->l[2*i+1]);
        }
    }

    // 2 * (x, y) for an affine point, cooperatively (xyzz_dev<fp2x_dev>::dbl_affine: mdbl-2008-s-1); x < 2 p, y < 3 p
    template<unsigned ROLE>
    SPPARK_DEVFN void dbl_affine(const fp& x, const fp& y, const g2c_ctx<F2>& c)
    {
        const fp U = (y + y).norm();                                // < 6 p
        c.open(); c.put(0, U); c.put(1, x); c.shut();
        const fp Uo = c.other(0), xo = c.other(1);
        const fp V  = g2c_sqr<F2, 7, ROLE>(U, Uo);
        const fp M  = g2c_sqr<F2, 3, ROLE>(x, xo);
        const fp M3 = (M + M + M).norm();                           // < 6 p
        c.open(); c.put(2, V); c.put(3, M3); c.shut();              // (slots 0, 1 still hold U, x)
        const fp Vo = c.other(2), M3o = c.other(3);
        const fp W  = g2c_mul<F2, 7, ROLE>(U, Uo, V, Vo);
        const fp S  = g2c_mul<F2, 3, ROLE>(x, xo, V, Vo);
        const fp X3 = fp::template sub<5, 2>(g2c_sqr<F2, 7, ROLE>(M3, M3o), S + S).norm();    // < 7 p
        const fp D  = fp::template sub<8>(S, X3).norm();            // < 10 p
        c.open(); c.put(0, D); c.put(1, W); c.put(4, y); c.shut();  // (slot 3 still holds M3)
        const fp Do = c.other(0), Wo = c.other(1), yo = c.other(4);
        Y = fp::template sub<3>(g2c_mul<F2, 11, ROLE>(D, Do, M3, M3o), g2c_mul<F2, 3, ROLE>(W, Wo, y, yo)).norm();    // < 5 p
        X = X3; ZZ = V; ZZZ = W;
    }

    // this += +-p, or this = +-p when |restart| (the first entry of a bucket); madd-2008-s as xyzz_dev<fp2x_dev>::madd.
    // Lanes whose point is at infinity keep their bucket.  Every lane of both waves runs every step.
    // ROLE = the component this wave owns (wave-uniform: the kernel branches on it once, at its top)
    template<unsigned ROLE>
    SPPARK_DEVFN void madd(const g2c_affine<F2>& p, bool negate, bool restart, const g2c_ctx<F2>& c)
    {
        // -- exchange 1: the operands of U2 = x ZZ, S2 = y ZZZ; "this bucket is at infinity"
        c.open();
        c.put(0, p.X); c.put(1, p.Y); c.put(2, ZZ); c.put(3, ZZZ);
        c.put_flag(ZZ.limbs_all_zero() ? 1u : 0u);
        c.shut();
        const bool from_point = restart || (ZZ.limbs_all_zero() && c.other_flag() != 0);
        const fp U2 = g2c_mul<F2, 3, ROLE>(p.X, c.other(0), ZZ, c.other(2));           // < 2 p, n
        fp S2 = g2c_mul<F2, 3, ROLE>(p.Y, c.other(1), ZZZ, c.other(3));
        if (negate) S2 = fp::template neg<3>(S2).norm();                            // < 3 p
        const fp Pd = fp::template sub<KX>(U2, X).norm();                           // < 12 p
        const fp Rd = fp::template sub<KY>(S2, Y).norm();                           // < 9 p
        // -- exchange 2: P, R and whether they vanish
        c.open();
        c.put(0, Pd); c.put(1, Rd);
        c.put_flag((Pd.template is_zero_mod<12>() ? 1u : 0u) | (Rd.template is_zero_mod<9>() ? 2u : 0u));
        c.shut();
        const u32 zf = c.ex->flag[0][c.lane] & c.ex->flag[1][c.lane];               // both components vanish
        const bool p_zero = (zf & 1u) != 0, r_zero = (zf & 2u) != 0;
        const fp PP = g2c_sqr<F2, 13, ROLE>(Pd, c.other(0));                           // < 2 p
        const fp RR = g2c_sqr<F2, 10, ROLE>(Rd, c.other(1));
        // -- exchange 3: PP and X (slot 0 still holds P, slot 2 ZZ: re-read, not kept in registers)
        c.open(); c.put(1, PP); c.put(3, X); c.shut();
        const fp PPo = c.other(1);
        const fp PPP = g2c_mul<F2, 13, ROLE>(Pd, c.other(0), PP, PPo);
        const fp Q   = g2c_mul<F2, KX, ROLE>(X, c.other(3), PP, PPo);
        const fp ZZn = g2c_mul<F2, 3, ROLE>(ZZ, c.other(2), PP, PPo);
        const fp T   = PPP + Q + Q;                                                 // < 6 p, limbs < 3 * 2^LB
        const fp X3  = fp::template sub<7, 3>(RR, T).norm();                        // < 9 p
        const fp D   = fp::template sub<10>(Q, X3).norm();                          // < 12 p
        // -- exchange 4: the operands of Y3 = D R - Y PPP and ZZZ3 = ZZZ PPP
        c.open(); c.put(0, D); c.put(1, Rd); c.put(2, Y); c.put(3, PPP); c.put(4, ZZZ); c.shut();
        const fp PPPo = c.other(3);
        const fp Yn   = fp::template sub<3>(g2c_mul<F2, 13, ROLE>(D, c.other(0), Rd, c.other(1)),
                                            g2c_mul<F2, KY, ROLE>(Y, c.other(2), PPP, PPPo)).norm();     // < 5 p
        const fp ZZZn = g2c_mul<F2, 3, ROLE>(ZZZ, c.other(4), PPP, PPPo);
        // -- the lanes' cases (identical in both waves); the doubling last, so that its operands are the only extra state
        const bool twice = !p.inf && !from_point && p_zero && r_zero;               // the same point: 2 p
        if (!p.inf) {
            if (from_point)   { X = p.X; Y = negate ? fp::template neg<3>(p.Y).norm() : p.Y; ZZZ = one_component(ROLE); ZZ = one_component(ROLE); }
            else if (!p_zero) { X = X3; Y = Yn; ZZZ = ZZZn; ZZ = ZZn; }
            else if (!r_zero) set_inf();
        }
        if (coop_any(twice)) {                                                      // (rare; a vote of the pair)
            g2c_bucket d;
            d.template dbl_affine<ROLE>(p.X, negate ? fp::template neg<3>(p.Y).norm() : p.Y, c);
            if (twice) *this = d;
        }
    }
};

} // namespace sppark_amd
