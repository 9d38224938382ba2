// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
//
// The reference has no CPU NTT.  This is a definition-level restatement of what
// its GPU driver computes, following the dispatch in ntt/ntt.cuh:161-213
// (NTT_internal), the coset scaling of ntt/kernels.cu:131-153
// (LDE_distribute_powers) and the root conventions of ntt/parameters.cuh:222-337
// + ntt/parameters/{goldilocks,baby_bear}.h:
//
//   omega      = forward_roots_of_unity[lg]   (2^lg-th primitive root; every
//                table entry is a repeated square of the table's last entry)
//   inverse    : omega^-1 and a final multiplication by domain_size_inverse[lg]
//   NN : bit_rev, CT      NR : GS      RN : CT      RR : GS, bit_rev
//   coset fwd : a[idx] *= g^(bitrev ? rev(idx) : idx) before the transform
//   coset inv : a[idx] *= g^-(!bitrev ? rev(idx) : idx) after it
//   where bitrev = (order != NR), exactly as ntt.cuh:174-209 sets it.
//
//   GS = decimation-in-frequency, natural in  -> bit-reversed out
//   CT = decimation-in-time,      bit-reversed in -> natural out
//
// ntt_naive() is the O(n^2) textbook DFT used to pin the fast restatement.
// PINNED by the reference itself: live on the GPU box (tests/test_ntt_vs_reference_gpu.py holds the product against the reference's own
// HIP build and against this file) and, in the CPU suite, against recorded outputs of that build (tests/golden/ref_ntt_golden.json).
#pragma once
#include "ff.hpp"
#include <vector>

namespace oracle {

enum ntt_order { NN = 0, NR = 1, RN = 2, RR = 3 };         // ntt/ntt.cuh:33
enum ntt_direction { forward = 0, inverse = 1 };           // ntt/ntt.cuh:34
enum ntt_type { standard = 0, coset = 1 };                 // ntt/ntt.cuh:35

static inline size_t bit_rev(size_t i, unsigned lg)
{
    size_t r = 0;
    for (unsigned k = 0; k < lg; k++) r |= ((i >> k) & 1) << (lg - 1 - k);
    return r;
}

template<class F> static F root_of_unity(unsigned lg)
{
    F w = F::top_root();
    for (unsigned k = F::TWO_ADICITY; k > lg; k--) w *= w;
    return w;
}

template<class F> static F finv(F a, unsigned lg_order)   // a has order 2^lg_order
{   return fpow(a, ((uint64_t)1 << lg_order) - 1);   }

template<class F> static F field_inverse(F a);             // a^(p-2)
template<> inline gl64 field_inverse<gl64>(gl64 a) { return fpow(a, gl64::MOD - 2); }
template<> inline bb31 field_inverse<bb31>(bb31 a) { return fpow(a, (uint64_t)bb31::MOD - 2); }
template<class P> static mont_t<P> field_inverse(mont_t<P> a) { return a.reciprocal(); }

template<class F> static void bit_rev_permute(F* a, unsigned lg)
{
    size_t n = (size_t)1 << lg;
    for (size_t i = 0; i < n; i++) {
        size_t r = bit_rev(i, lg);
        if (i < r) { F t = a[i]; a[i] = a[r]; a[r] = t; }
    }
}

template<class F> static void gs_dif(F* a, unsigned lg, F omega)
{
    size_t n = (size_t)1 << lg;
    for (size_t half = n >> 1; half >= 1; half >>= 1) {
        // twiddle step for this stage: omega^(n / (2*half))
        F wstep = omega;
        for (size_t m = 2 * half; m < n; m <<= 1) wstep *= wstep;
        for (size_t blk = 0; blk < n; blk += 2 * half) {
            F w = F::one();
            for (size_t j = 0; j < half; j++) {
                F u = a[blk + j], v = a[blk + j + half];
                a[blk + j] = u + v;
                a[blk + j + half] = (u - v) * w;
                w *= wstep;
            }
        }
    }
}

template<class F> static void ct_dit(F* a, unsigned lg, F omega)
{
    size_t n = (size_t)1 << lg;
    for (size_t half = 1; half < n; half <<= 1) {
        F wstep = omega;
        for (size_t m = 2 * half; m < n; m <<= 1) wstep *= wstep;
        for (size_t blk = 0; blk < n; blk += 2 * half) {
            F w = F::one();
            for (size_t j = 0; j < half; j++) {
                F u = a[blk + j], v = a[blk + j + half] * w;
                a[blk + j] = u + v;
                a[blk + j + half] = u - v;
                w *= wstep;
            }
        }
    }
}

template<class F> static void lde_powers(F* a, unsigned lg, bool bitrev, F g)
{
    size_t n = (size_t)1 << lg;
    std::vector<F> pw(n);
    F x = F::one();
    for (size_t i = 0; i < n; i++) { pw[i] = x; x *= g; }
    for (size_t i = 0; i < n; i++) a[i] *= pw[bitrev ? bit_rev(i, lg) : i];
}

template<class F>
static void ntt(F* a, unsigned lg, int order, int direction, int type)
{
    if (lg == 0) return;                                   // ntt/ntt.cuh:220-221
    const bool intt = direction == inverse;
    size_t n = (size_t)1 << lg;

    F omega = root_of_unity<F>(lg);
    if (intt) omega = finv(omega, lg);
    F g = F::group_gen();
    if (intt) g = field_inverse(g);

    bool bitrev, use_gs;
    switch (order) {
        case NN: bit_rev_permute(a, lg); bitrev = true;  use_gs = false; break;
        case NR:                         bitrev = false; use_gs = true;  break;
        case RN:                         bitrev = true;  use_gs = false; break;
        default:                         bitrev = true;  use_gs = true;  break;  // RR
    }

    if (!intt && type == coset) lde_powers(a, lg, bitrev, g);

    if (use_gs) gs_dif(a, lg, omega); else ct_dit(a, lg, omega);

    if (intt) {
        F ninv = field_inverse(fpow(F::one() + F::one(), lg));
        for (size_t i = 0; i < n; i++) a[i] *= ninv;
    }

    if (intt && type == coset) lde_powers(a, lg, !bitrev, g);

    if (order == RR) bit_rev_permute(a, lg);
}

// NTT::LDE_aux (ntt/ntt.cuh:283-336).  inout: 2^(lg_domain+lg_blowup) elements, the first
// 2^lg_domain hold the input; aux (nullable): 2^lg_domain elements.
template<class F>
static void lde(F* inout, unsigned lg_domain, unsigned lg_blowup, F* aux)
{
    const size_t dom = (size_t)1 << lg_domain, ext = dom << lg_blowup;
    std::vector<F> d(inout, inout + dom);
    ntt(d.data(), lg_domain, NR, inverse, standard);                   // :301-303
    if (aux) {                                                         // :312-315 bit_rev(aux, domain)
        for (size_t i = 0; i < dom; i++) aux[bit_rev(i, lg_domain)] = d[i];
    }
    // LDE_spread_distribute_powers, perform_shift = true, ext_pow = false
    // (ntt/kernels.cu:155-237): r = in[idx] * g^bit_rev(idx, lg_domain) lands at
    // out[idx << lg_blowup], the other 2^lg_blowup - 1 slots are zero
    std::vector<F> pw(dom);
    F x = F::one(), g = F::group_gen();
    for (size_t i = 0; i < dom; i++) { pw[i] = x; x *= g; }
    const F zero = F::one() - F::one();
    for (size_t o = 0; o < ext; o++) inout[o] = zero;
    for (size_t idx = 0; idx < dom; idx++) inout[idx << lg_blowup] = d[idx] * pw[bit_rev(idx, lg_domain)];
    ntt(inout, lg_domain + lg_blowup, RN, forward, standard);          // :321-323
}

// NTT::LDE_powers(stream, d_inout, lg) (ntt/ntt.cuh:352-356) = LDE_distribute_powers with
// bitrev = true, lg_blowup = 0, forward generator (ntt/kernels.cu:131-153)
template<class F>
static void lde_powers_bitrev(F* inout, unsigned lg) { lde_powers(inout, lg, true, F::group_gen()); }

// NTT::LDE_expand (ntt/ntt.cuh:358-365): spread without the shift
template<class F>
static void lde_expand(F* out, const F* in, unsigned lg_domain, unsigned lg_blowup)
{
    const size_t dom = (size_t)1 << lg_domain, ext = dom << lg_blowup;
    const F zero = F::one() - F::one();
    for (size_t o = 0; o < ext; o++) out[o] = zero;
    for (size_t idx = 0; idx < dom; idx++) out[idx << lg_blowup] = in[idx];
}

// textbook DFT, natural in / natural out: X[k] = sum_j x[j] w^(jk)
template<class F>
static void ntt_naive(F* out, const F* in, unsigned lg, bool inv)
{
    size_t n = (size_t)1 << lg;
    F omega = root_of_unity<F>(lg);
    if (inv) omega = finv(omega, lg);
    F ninv = field_inverse(fpow(F::one() + F::one(), lg));
    F wk = F::one();
    for (size_t k = 0; k < n; k++) {
        F acc = in[0] - in[0], w = F::one();
        for (size_t j = 0; j < n; j++) { acc = acc + in[j] * w; w *= wk; }
        out[k] = inv ? acc * ninv : acc;
        wk *= omega;
    }
}

} // namespace oracle
