// The G2 accumulation with one Fp2 component per wave (ec/xyzz2_coop.hpp): k_accumulate's walk over the sorted entries
// of a chunk, by a PAIR of waves per 64 chunks.  The default accumulation of G2 over the 14-limb base fields (BLS12-381,
// BLS12-377; msm_driver.hpp, msm_tunables::g2_coop): BLS12-381 G2 2^22 47.8 -> 40.2 ms, profiles/r05_g2_coop_ab.log.
// Covered on the GPU by every G2 test (tests/test_msm_gpu.py runs them over both accumulation paths) and on the host
// emulation (tests/emu/emu_msm.cpp, SPPARK_G2).
//
// Same inputs, same outputs (records, keys, buckets) as accumulate_chunk<fp2x_dev> in msm_kernels.hpp.  The control flow
// is made UNIFORM over the work-group, because every mixed addition contains barriers: each lane runs exactly L
// iterations (a chunk with fewer entries -- the last of a window, or none -- adds points at infinity), and the flush at a
// bucket boundary is a predicated store followed by the same madd call with |restart| set.
#pragma once
#include "msm_kernels.hpp"
#include "../ec/xyzz2_coop.hpp"

namespace sppark_amd {

template<class F2, unsigned ROLE>
SPPARK_DEVFN void accumulate_chunk_g2c(xyzz_mem<F2::N>* buckets, u32* rec_key, xyzz_mem<F2::N>* rec_pt,
                                       const unsigned char* points,
                                       const u32* sorted, const u32* off,
                                       unsigned n, unsigned NB, unsigned L, unsigned chunks_per_win,
                                       unsigned chunk, unsigned w_local, unsigned w_base, const g2c_ctx<F2>& c)
{
    constexpr unsigned role = ROLE;
    const unsigned w = w_base + w_local;
    const bool live = chunk < chunks_per_win;
    const size_t rec0 = ((size_t)w * chunks_per_win + (live ? chunk : 0)) * 2;
    const u32* o = off + (size_t)w_local * (NB + 1);
    const unsigned total = o[NB];
    unsigned p = live ? chunk * L : total;
    const bool any = live && p < total;
    const unsigned end = any ? (total < p + L ? total : p + L) : p;
    if (live && !any && role == 0) { rec_key[rec0] = KEY_NONE; rec_key[rec0 + 1] = KEY_NONE; }

    unsigned b = 0, next = 0;
    if (any) {                                      // bucket holding position p: o[b] <= p < o[b+1]
        unsigned lo = 0, hi = NB;
        while (hi - lo > 1) {
            unsigned mid = (lo + hi) >> 1;
            if (o[mid] <= p) lo = mid; else hi = mid;
        }
        b = lo; next = o[b + 1];
    }
    const u32* src = sorted + (size_t)w_local * n;
    g2c_bucket<F2> acc; acc.set_inf();
    bool first_run = true;
    u32 slot0_key = KEY_NONE;

    for (unsigned i = 0; i < L; i++, p++) {         // uniform trip count: barriers inside madd
        const bool act = p < end;
        u32 e = 0;
        g2c_affine<F2> pt = g2c_affine<F2>::infinity();
        if (act) { e = src[p]; pt = g2c_affine<F2>::load(points, e & 0x7fffffffu, role); }
        bool restart = i == 0;
        if (act && i > 0 && p == next) {            // bucket boundary: flush this wave's components
            const u32 key = w * NB + b;
            if (first_run) { acc.store(&rec_pt[rec0], role); slot0_key = key; first_run = false; }
            else           acc.store(&buckets[key], role);
            do { b++; next = o[b + 1]; } while (p == next);
            restart = true;
        }
        if (act && pt.inf && restart) { acc.set_inf(); restart = false; }      // (a bucket that starts with the point at infinity)
        acc.template madd<ROLE>(pt, (e >> 31) != 0, restart, c);
    }
    if (!any) return;
    const u32 key = w * NB + b;
    if (first_run) { acc.store(&rec_pt[rec0], role); if (role == 0) { rec_key[rec0] = key; rec_key[rec0 + 1] = KEY_NONE; } }
    else           { acc.store(&rec_pt[rec0 + 1], role); if (role == 0) { rec_key[rec0] = slot0_key; rec_key[rec0 + 1] = key; } }
}

static constexpr unsigned G2C_NT = 128;             // a pair of waves
// (the converted records carry their own infinity flag: one instantiation serves both wire layouts)
template<class F2>
__global__ __launch_bounds__(128, 2)
void k_accumulate_g2c(xyzz_mem<F2::N>* __restrict__ buckets,
                      u32* __restrict__ rec_key, xyzz_mem<F2::N>* __restrict__ rec_pt,
                      const unsigned char* __restrict__ points, unsigned stride,
                      const u32* __restrict__ sorted, const u32* __restrict__ off,
                      unsigned n, unsigned NB, unsigned L, unsigned chunks_per_win, unsigned w_base)
{
    (void)stride;                                   // (the converted records have their own stride)
    __shared__ g2c_lds<F2> ex;
    const g2c_ctx<F2> c{&ex, threadIdx.x >> 6, threadIdx.x & 63};
    // The component is WAVE-uniform: one branch here, two specialised walks (each wave computes only its component's
    // formula of every Fp2 product).  The two walks execute the same sequence of work-group barriers -- every barrier sits
    // in madd / dbl_affine / coop_any, which both call at the same points of the same trip counts -- and no barrier is
    // under a condition that differs between the LANES of a wave.  A work-group barrier on CDNA is s_barrier, which
    // counts arriving WAVES, not program addresses (and the fences around it are per wave), so the two copies of each
    // barrier pair up; the vote (coop_any -> __ockl_wgred_or_i32) is one out-of-line routine with one LDS cell whatever
    // the call site.  Exercised on hardware by every G2 GPU test.
    if (c.role == 0) accumulate_chunk_g2c<F2, 0>(buckets, rec_key, rec_pt, points, sorted, off, n, NB, L, chunks_per_win,
                                                          blockIdx.x * 64 + c.lane, blockIdx.y, w_base, c);
    else             accumulate_chunk_g2c<F2, 1>(buckets, rec_key, rec_pt, points, sorted, off, n, NB, L, chunks_per_win,
                                                          blockIdx.x * 64 + c.lane, blockIdx.y, w_base, c);
}

} // namespace sppark_amd
