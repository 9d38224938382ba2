// The record list of a SMALL MSM: the pieces of a bucket summed by a tree over the bucket's OWN pieces.
//
// Below ~2^17 points a window has few buckets (2^3 .. 2^7) of hundreds of entries, so a bucket is cut by
// k_accumulate's fixed-length runs into tens of pieces (2^16 points: 32, 2^12: 64) instead of the two that
// k_join_runs' walk is made for.  The fan-in tree (reduce_runs_chunk) halves the WHOLE record list per level
// whatever the segments look like: 2^16 points are 262 144 records = 18 levels of up to three dependent additions,
// eleven launches, 0.29 ms of a 0.87 ms MSM, although every segment is done after six
// (profiles/r06_msm_timeline_2p16_before.txt).
//
// The pieces of a bucket need no search: bucket b of window w holds the entries [o_b, e_b) of the window's grouped
// list (the sort's offsets), the runs are the fixed ranges [rho L, (rho + 1) L), so b touches the runs
// rho0 = o_b / L .. rho1 = (e_b - 1) / L and its pieces are
//     piece 0    the LAST-run record (slot 1) of run rho0 when b starts inside that run, its first-run record (slot 0)
//                when b starts exactly at rho0 L;
//     piece k    the first-run record of run rho0 + k (b is the first bucket of every further run it touches).
// (A bucket strictly inside one run has no record: k_accumulate stored it.)  Level t of the tree adds piece
// (2m + 1) 2^t into piece 2m 2^t -- one launch per level, every addition of a level independent, one dependent
// addition per level -- and the last level stores piece 0 into buckets[].  T = log2(CMAX) levels for buckets of up to
// CMAX pieces, chosen from the AVERAGE bucket with head-room for the denser top window (piece_cmax).  A bucket
// with more pieces (skewed scalars) is left alone: its records keep their keys, the others' keys are cleared at
// level 0, a device flag says that such a bucket exists, and the driver then runs the fan-in tree over what is left
// (msm_driver.hpp: after the fact -- the flag comes back with the window sums, and the tail is redone in the rare
// case it is set).
#pragma once
#include "msm_kernels.hpp"

namespace sppark_amd {

struct piece_geom {
    unsigned cnt;       // pieces (0: the bucket has no record)
    size_t root;        // record slot of piece 0
    size_t run0;        // record slot of the first-run record of run rho0 (piece k > 0: run0 + 2 k)
};

SPPARK_DEVFN piece_geom piece_geometry(const u32* off, unsigned NB, unsigned L, unsigned chunks_per_win, unsigned w, unsigned b)
{
    piece_geom g; g.cnt = 0; g.root = 0; g.run0 = 0;
    const u32* o = off + (size_t)w * (NB + 1);
    const unsigned ob = o[b], eb = o[b + 1];
    if (eb == ob) return g;
    const unsigned r0 = ob / L, r1 = (eb - 1) / L;
    const bool aligned = ob % L == 0;
    if (r1 == r0 && !aligned) {
        const unsigned total = o[NB];
        const unsigned run_end = total < (r0 + 1) * L ? total : (r0 + 1) * L;
        if (eb < run_end) return g;                                 // strictly inside one run: stored by k_accumulate
    }
    g.cnt = r1 - r0 + 1;
    g.run0 = ((size_t)w * chunks_per_win + r0) * 2;
    g.root = aligned ? g.run0 : g.run0 + 1;
    return g;
}
SPPARK_DEVFN size_t piece_slot(const piece_geom& g, unsigned k) { return k == 0 ? g.root : g.run0 + 2 * (size_t)k; }

// work item (bucket B = w NB + b, pair m) of level t; pairs per bucket at this level: cmax >> (t + 1)
struct piece_job { bool live, add, finish; size_t dst, src; u32 B; };
// pair m of bucket B (< nwins NB) at level t
SPPARK_DEVFN piece_job piece_job_bm(u32* rec_key, const u32* off, unsigned NB, unsigned L, unsigned chunks_per_win,
                                    unsigned cmax, unsigned t, unsigned last, u32* any_long, size_t B, unsigned m)
{
    piece_job j; j.live = j.add = j.finish = false; j.dst = j.src = 0; j.B = 0;
    const unsigned pm = cmax >> (t + 1);
    if (m >= pm) return j;
    const piece_geom g = piece_geometry(off, NB, L, chunks_per_win, (unsigned)(B / NB), (unsigned)(B % NB));
    if (g.cnt == 0) return j;
    if (g.cnt > cmax) { if (t == 0 && m == 0) *any_long = 1; return j; }
    const unsigned k0 = m << (t + 1), k1 = k0 + (1u << t);
    if (k0 >= g.cnt) return j;
    j.live = true; j.B = (u32)B;
    j.dst = piece_slot(g, k0);
    if (k1 < g.cnt) { j.add = true; j.src = piece_slot(g, k1); }
    if (t == 0) {                                                   // the fan-in tree must not see these records
        rec_key[j.dst] = KEY_NONE;
        if (j.add) rec_key[j.src] = KEY_NONE;
    }
    j.finish = last && m == 0;
    return j;
}
SPPARK_DEVFN piece_job piece_job_of(u32* rec_key, const u32* off, unsigned NB, unsigned L, unsigned chunks_per_win, unsigned nwins,
                                    unsigned cmax, unsigned t, unsigned last, u32* any_long, size_t id)
{
    // pair-major: lane l of a wave is bucket B0 + l of ONE pair index m, so the waves of the pair indices beyond the
    // average bucket's pieces (cmax has 3 x head-room) hold no work at all and leave at once, and those that stay are full
    // (bucket-major, a wave was 64 pair indices of one bucket: a quarter of its lanes busy, four times the waves)
    const size_t nb = (size_t)nwins * NB;
    return piece_job_bm(rec_key, off, NB, L, chunks_per_win, cmax, t, last, any_long, id % nb, (unsigned)(id / nb));
}

// The narrow end of the tree in ONE launch (k_piece_tail_coop): every level from t0 on, a work-group owning 2^lgGB buckets
// with ALL their pairs, so that a level only waits for the work-group's own stores.  Work item |idx| of work-group |wg| at
// level t: bucket (wg << lgGB) + idx % 2^lgGB, pair idx >> lgGB.  lgGB fills the 64 lanes of a cooperative addition at
// level t0: 2^lgGB (cmax >> (t0 + 1)) >= 64 where the buckets allow.
static inline unsigned piece_tail_lgGB(unsigned cmax, unsigned t0)
{
    const unsigned pm = cmax >> (t0 + 1);
    unsigned lg = 0;
    while ((pm << lg) < 64) lg++;
    return lg;
}
// first level of the one-launch end: the first whose work items (buckets x pair slots) are at most |fuse_max|; the levels
// before it are launches of their own (lane-per-addition kernels: throughput, not latency)
static inline unsigned piece_tail_t0(size_t nbuckets, unsigned cmax, size_t fuse_max)
{
    unsigned t = 0;
    while ((cmax >> (t + 1)) >= 1 && nbuckets * (cmax >> (t + 1)) > fuse_max) t++;
    return t;
}

template<class FP>
SPPARK_DEVFN void piece_apply(xyzz_mem<FP::N>* buckets, xyzz_mem<FP::N>* rec_pt, const piece_job& j)
{
    if (!j.live || !(j.add || j.finish)) return;
    xyzz_dev<FP> acc = xyzz_dev<FP>::load(&rec_pt[j.dst]);
    if (j.add) bucket_add_fast<FP>(acc, xyzz_dev<FP>::load(&rec_pt[j.src]));
    acc.store(j.finish ? &buckets[j.B] : &rec_pt[j.dst]);
}

template<class FP>
SPPARK_DEVFN void piece_level_item(xyzz_mem<FP::N>* buckets, u32* rec_key, xyzz_mem<FP::N>* rec_pt, const u32* off,
                                   unsigned NB, unsigned L, unsigned chunks_per_win, unsigned nwins,
                                   unsigned cmax, unsigned t, unsigned last, u32* any_long, size_t id)
{
    piece_apply<FP>(buckets, rec_pt, piece_job_of(rec_key, off, NB, L, chunks_per_win, nwins, cmax, t, last, any_long, id));
}

template<class FP>
__global__ __launch_bounds__(256)
void k_piece_level(xyzz_mem<FP::N>* __restrict__ buckets, u32* __restrict__ rec_key, xyzz_mem<FP::N>* rec_pt,
                   const u32* __restrict__ off, unsigned NB, unsigned L, unsigned chunks_per_win, unsigned nwins,
                   unsigned cmax, unsigned t, unsigned last, u32* __restrict__ any_long)
{
    piece_level_item<FP>(buckets, rec_key, rec_pt, off, NB, L, chunks_per_win, nwins, cmax, t, last, any_long,
                         (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// CMAX for an average bucket of |avg_pieces| pieces: a power of two >= 3 x + 4.  (Uniform scalars are NOT uniform digits in
// the top window: it is a bit shorter than the others when the scalar bits do not divide evenly, and the modulus cuts its
// range -- BLS12-381's r = 0x73ed... leaves 115 of the 128 values of a 7-bit top window, all of magnitude <= 64: 2.2 x the
// entries per bucket.  The extra levels are launches of a few lanes that find nothing to add.)
static inline unsigned piece_cmax_exact(size_t want)            // the power of two >= want
{
    unsigned c = 2;
    while (c < want && c < 4096) c <<= 1;
    return c;
}
static inline unsigned piece_cmax(size_t avg_pieces) { return piece_cmax_exact(3 * avg_pieces + 4); }

} // namespace sppark_amd
