// Cooperative XYZZ addition / doubling over the loosely-reduced field (ff/montx_dev.hpp): FOUR waves of a work-group
// compute ONE vector of 64 point operations.
//
// Why.  The tail of an MSM (record tree, bucket-sum levels, the subset-sum top) is chains of dependent point
// operations run by a handful of waves.  A wave of a chain issues one instruction every few clocks whatever its
// neighbours do (profiles/r02_ubench_instruction_rates.log: 0.08 per clock and SIMD with one wave, 0.17 with two),
// so the time of a chain is its INSTRUCTION COUNT: ~6800 vector instructions for a full addition with its products
// in interleaved pairs (xyzz_dev::add_pairs), ~3500 for a doubling.  The products of one addition are mostly
// independent, though: U1, S1, U2, S2 | PP, RR, ZZ*ZZ', ZZZ*ZZZ' | PPP, Q, ZZ'' | Y3, ZZZ'' -- four levels.  Here the four
// waves hold IDENTICAL copies of the 64 operand pairs, each computes ONE product per level, and the results go round
// through LDS (limb-major, lane-contiguous: conflict-free).  The critical path is 4 products + the lazy additions
// (~2500 instructions) instead of 14 products; the price is 4x the waves, which these kernels have to spare -- they
// never fill the SIMDs.  The reference spreads ONE field element over two lanes for Fp2 with shuffles
// (ff/bls12-381-fp2.hpp:25-150) and reduces a bucket's partial sums with a warp-shuffle tree
// (msm/batch_addition.cuh:134-181): the same intent, lane-level; wave-level here because a wave is the unit that
// gets an issue slot.
//
// Contract: every lane of all four waves calls the operation (barriers inside); role = wave index in the work-group
// (tid >> 6), lane = tid & 63; the four copies of a lane's operands are equal on entry and the four copies of the
// result are equal on exit.  Operand / result bounds are those of ec/xyzzx_dev.hpp (X < 10p, Y < 5p, ZZ, ZZZ < 2p
// normalised).  Exceptional lanes (an operand at infinity; equal or opposite points) are resolved per lane after the
// cooperative part, by the serial formulas, identically in the four copies.
#pragma once
#include "xyzzx_dev.hpp"

namespace sppark_amd {

// exchange area of one work-group: two sets (alternating per level, so that a level's writes never meet the previous
// level's reads) of four slots of NL x 64 words
template<class F> struct coop_lds {
    u32 w[2][4][F::NL][64];
    SPPARK_BND(double bv[2][4][64]; double bl[2][4][64];)      // (host bound tracking: the claims travel with the limbs)
};

// Work-group barrier / barrier with an OR-vote.  Host emulation (tests/emu/emu_coop.cpp) runs a work-group as 256 host
// threads and supplies the two hooks.
#if defined(SPPARK_HOST_EMULATION)
extern "C" void sppark_emu_barrier();
extern "C" int sppark_emu_barrier_or(int);
#endif
SPPARK_DEVFN void coop_barrier()
{
#if defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
#elif defined(SPPARK_HOST_EMULATION)
    sppark_emu_barrier();
#endif
}
SPPARK_DEVFN bool coop_any(bool p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __syncthreads_or(p) != 0;
#elif defined(SPPARK_HOST_EMULATION)
    return sppark_emu_barrier_or(p) != 0;
#else
    return p;
#endif
}

template<class F> struct coop_ctx {
    coop_lds<F>* ex;
    unsigned role, lane, par;           // par: the set the NEXT level writes
    SPPARK_DEVFN void put(unsigned slot, const F& v) const
    {
        #pragma unroll
        for (int j = 0; j < F::NL; j++) ex->w[par][slot][j][lane] = v.l[j];
        SPPARK_BND(ex->bv[par][slot][lane] = v.bv; ex->bl[par][slot][lane] = v.bl;)
    }
    // (reads the set written by the level that just ended: call after next_level())
    SPPARK_DEVFN F get(unsigned slot) const
    {
        F r;
        #pragma unroll
        for (int j = 0; j < F::NL; j++) r.l[j] = ex->w[par ^ 1][slot][j][lane];
        SPPARK_BND(r.bnd_set(ex->bv[par ^ 1][slot][lane], ex->bl[par ^ 1][slot][lane]);)
        return r;
    }
    SPPARK_DEVFN void next_level() { coop_barrier(); par ^= 1; }
};

// the serial addition of the exceptional lanes (equal or opposite points): ONE out-of-line copy per kernel image, not one
// inlined 7000-instruction body per cooperative call site
#if defined(SPPARK_HOST_EMULATION)
template<class F> inline void coop_serial_add(xyzz_dev<F>& a, const xyzz_dev<F>& b) { a.add(b); }
#else
template<class F> __device__ __noinline__ void coop_serial_add(xyzz_dev<F>& a, const xyzz_dev<F>& b) { a.add(b); }
#endif

// a += b (full addition, add-2008-s).  All four waves return the same |a|.
template<class F>
SPPARK_DEVFN void coop_add(xyzz_dev<F>& a, const xyzz_dev<F>& b, coop_ctx<F>& c)
{
    const bool a_inf = a.is_inf(), b_inf = b.is_inf();
    const unsigned role = c.role;
    // level 1: U1 = X1*ZZ2, S1 = Y1*ZZZ2, U2 = X2*ZZ1, S2 = Y2*ZZZ1 (fat left operands, normalised right ones)
    {
        F t;
        if (role == 0)      t = a.X * b.ZZ;
        else if (role == 1) t = a.Y * b.ZZZ;
        else if (role == 2) t = b.X * a.ZZ;
        else                t = b.Y * a.ZZZ;
        c.put(role, t);
    }
    c.next_level();
    const F U1 = c.get(0), S1 = c.get(1);
    const F Pd = F::template sub<3>(c.get(2), U1).norm();      // < 5p, n
    const F Rd = F::template sub<3>(c.get(3), S1).norm();
    const bool special = !a_inf && !b_inf && Pd.template is_zero_mod<5>();
    // level 2: PP = P^2, RR = R^2, and the two products that do not depend on P: ZZ1*ZZ2, ZZZ1*ZZZ2
    F keep = F::zero();
    if (role == 0)      c.put(0, Pd.sqr());
    else if (role == 1) c.put(1, Rd.sqr());
    else if (role == 2) keep = a.ZZ * b.ZZ;                     // n x n, < 2p
    else                keep = a.ZZZ * b.ZZZ;
    c.next_level();
    const F PP = c.get(0), RR = c.get(1);
    // level 3: PPP = P*PP, Q = U1*PP, ZZ3 = (ZZ1*ZZ2)*PP
    if (role == 0)      c.put(0, Pd * PP);
    else if (role == 1) c.put(1, U1 * PP);
    else if (role == 2) c.put(2, keep * PP);
    c.next_level();
    const F PPP = c.get(0), Q = c.get(1), ZZ3 = c.get(2);
    const F T  = PPP + Q + Q;                                   // < 6p
    const F X3 = xyzz_dev<F>::keep_x(F::template sub<8, 3>(RR, T));     // < 10p, limbs <= 5*2^LB (29-bit limbs: n)
    // level 4: Y3 = R*(Q - X3) - S1*PPP as one reduced sum of two products; ZZZ3 = (ZZZ1*ZZZ2)*PPP
    if (role == 0) {
        const F D = F::template sub<11, xyzz_dev<F>::BX>(Q, X3);
        c.put(0, F::mul_add(D, Rd, F::template neg<3>(S1), PPP));
    } else if (role == 3) {
        c.put(3, keep * PPP);
    }
    c.next_level();
    xyzz_dev<F> r;
    r.X = X3; r.Y = c.get(0); r.ZZZ = c.get(3); r.ZZ = ZZ3;
    // exceptional lanes, identically in the four copies (no barrier below this line)
    if (b_inf) r = a;
    else if (a_inf) r = b;
    else if (special) {                                         // equal points (doubling) or opposite ones (infinity)
        // (copies: only THEY have their address taken by the out-of-line call -- with |a| and |b| themselves passed by
        // reference the operands lived in scratch memory on the hot path too: every cooperative kernel 25 % slower)
        xyzz_dev<F> ta = a, tb = b;
        coop_serial_add<F>(ta, tb);
        r = ta;
    }
    a = r;
}

// a = 2a (dbl-2008-s-1)
template<class F>
SPPARK_DEVFN void coop_dbl(xyzz_dev<F>& a, coop_ctx<F>& c)
{
    const bool a_inf = a.is_inf();
    const unsigned role = c.role;
    const F Yn = a.Y.norm(), Xn = a.X.norm();                   // n, < 5p / < 10p
    const F U = (Yn + Yn).norm();                               // < 10p
    // level 1: V = U^2, M = X^2
    if (role == 0)      c.put(0, U.sqr());
    else if (role == 1) c.put(1, Xn.sqr());
    c.next_level();
    const F V = c.get(0), M = c.get(1);
    const F M3 = (M + M + M).norm();                            // < 6p
    // level 2: W = U*V, S = X*V, ZZ3 = ZZ*V, M3^2
    {
        F t;
        if (role == 0)      t = U * V;
        else if (role == 1) t = Xn * V;
        else if (role == 2) t = a.ZZ * V;
        else                t = M3.sqr();
        c.put(role, t);
    }
    c.next_level();
    const F W = c.get(0), S = c.get(1), ZZ3 = c.get(2);
    const F X3 = xyzz_dev<F>::keep_x(F::template sub<5, 2>(c.get(3), S + S));   // < 7p, limbs <= 4*2^LB (29-bit limbs: n)
    // level 3: Y3 = M3*(S - X3) - W*Y as one reduced sum; ZZZ3 = ZZZ*W
    if (role == 0) {
        const F D = F::template sub<8, xyzz_dev<F>::BX2>(S, X3);       // < 10p
        c.put(0, F::mul_add(D, M3, F::template neg<3>(W), Yn));
    } else if (role == 1) {
        c.put(1, a.ZZZ * W);
    }
    c.next_level();
    xyzz_dev<F> r;
    r.X = X3; r.Y = c.get(0); r.ZZZ = c.get(1); r.ZZ = ZZ3;
    if (!a_inf) a = r;
}

} // namespace sppark_amd
