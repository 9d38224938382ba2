// The latency-bound end of the bucket sums with COOPERATIVE point operations (ec/xyzz_coop.hpp: four waves per 64
// operations): the subset-sum top (k_bucket_top_bits / k_bucket_top_sum of msm_kernels.hpp) and the narrow end of the
// record tree.  G1 bucket fields with their own records only.
//
// What the counters and timelines say about these kernels (profiles/r03_msm_timeline_2p*.txt, r04_msm_timeline_2p16_before.txt):
// at every size from 2^18 points on, k_bucket_top_bits + k_bucket_top_sum are ~0.55 ms of chains -- per work-group a
// gather of nitems/512 additions per lane, an 8-step tree over 256 lanes and b + lgG <= 16..20 doublings by ONE lane --
// run by (m + 1) x windows ~ 200 work-groups, one per CU: three quarters of every SIMD's issue slots are empty.  The tree
// and the doublings are where the cooperative forms apply directly: a tree level over s <= 64 pairs is one cooperative
// addition, and the doubling chain of lane 0 is a chain of cooperative doublings -- ~2.7x fewer instructions on the
// critical path each.
#pragma once
#include "msm_kernels.hpp"
#include "msm_piece_kernels.hpp"
#include "../ec/xyzz_coop.hpp"

namespace sppark_amd {

static constexpr unsigned COOP_NT = 256;        // four waves

// CAP points in LDS, word-major (lane-contiguous words: conflict-free 4-byte accesses)
template<class FP, unsigned CAP> struct coop_img {
    u32 w[4 * FP::N][CAP];
    SPPARK_DEVFN xyzz_dev<FP> load(unsigned i) const
    {
        constexpr int N = FP::N;
        xyzz_dev<FP> r;
        #pragma unroll
        for (int j = 0; j < N; j++) { r.X.l[j] = w[j][i]; r.Y.l[j] = w[N + j][i]; r.ZZZ.l[j] = w[2 * N + j][i]; r.ZZ.l[j] = w[3 * N + j][i]; }
        return r;
    }
    SPPARK_DEVFN void store(unsigned i, const xyzz_dev<FP>& v)
    {
        constexpr int N = FP::N;
        #pragma unroll
        for (int j = 0; j < N; j++) { w[j][i] = v.X.l[j]; w[N + j][i] = v.Y.l[j]; w[2 * N + j][i] = v.ZZZ.l[j]; w[3 * N + j][i] = v.ZZ.l[j]; }
    }
    // the four waves hold the same value: wave |role| writes coordinate |role|
    SPPARK_DEVFN void store_coord(unsigned i, unsigned role, const xyzz_dev<FP>& v)
    {
        constexpr int N = FP::N;
        const FP& f = role == 0 ? v.X : role == 1 ? v.Y : role == 2 ? v.ZZZ : v.ZZ;
        #pragma unroll
        for (int j = 0; j < N; j++) w[role * N + j][i] = f.l[j];
    }
};

// img[0] = sum of img[0 .. 2*s0): pairwise tree, one cooperative addition per 64 pairs and level.  All COOP_NT lanes call.
template<class FP, unsigned CAP>
SPPARK_DEVFN void coop_tree_sum(coop_img<FP, CAP>* img, unsigned s0, coop_ctx<FP>& c)
{
    for (unsigned s = s0; s >= 1; s >>= 1) {
        for (unsigned base = 0; base < s; base += 64) {
            const unsigned i = base + c.lane;
            const bool on = i < s;
            xyzz_dev<FP> x, y;
            if (on) { x = img->load(i); y = img->load(i + s); } else { x.set_inf(); y.set_inf(); }
            coop_add<FP>(x, y, c);                  // (barriers inside: every wave has loaded before any wave stores)
            if (on) img->store_coord(i, c.role, x);
        }
        coop_barrier();
    }
}

// (the exchange area of the cooperative operations + an image of COOP_NT / 2 points: 56 KB on fourteen limbs, two work-groups per CU)
static inline size_t top_bits_coop_lds(size_t nl) { return 2 * 4 * nl * 64 * 4 + 4 * nl * (COOP_NT / 2) * 4; }

// k_bucket_top_bits with the tree and the doubling chain in cooperative form (same per-lane gather, same sums), one work-group
// per PIECE of a sum (msm_kernels.hpp bucket_top_piece; sb = sp = 1: per sum, as k_bucket_top_bits): part q of window w.
// The first level of the tree is one-wave additions by the lower half of the lanes (the upper half hands over through the
// image, which then holds COOP_NT / 2 points: two work-groups fit a CU); the rest is cooperative.
// 2^20 points (16 windows of 4096 items, the plain sum in two pieces): 0.43 -> 0.36 ms; profiles/r06_msm_top_cut_sweep.log.
template<class FP>
__global__ __launch_bounds__(COOP_NT)
void k_bucket_top_bits_coop(xyzz_mem<FP::N>* __restrict__ parts, const xyzz_mem<FP::N>* __restrict__ A,
                            const xyzz_mem<FP::N>* __restrict__ Wt, unsigned nitems, unsigned m, unsigned lgG,
                            unsigned sb, unsigned sp)
{
    extern __shared__ unsigned char top_lds[];
    coop_lds<FP>* ex = reinterpret_cast<coop_lds<FP>*>(top_lds);
    coop_img<FP, COOP_NT / 2>* img = reinterpret_cast<coop_img<FP, COOP_NT / 2>*>(top_lds + sizeof(coop_lds<FP>));
    const unsigned q = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
    const top_piece pc = bucket_top_piece(q, m, sb, sp);
    {
        xyzz_dev<FP> acc = bucket_top_gather<FP>(A, Wt, nitems, m, pc.b, w, pc.sub * COOP_NT + tid, pc.nsub * COOP_NT);
        if (tid >= COOP_NT / 2) img->store(tid - COOP_NT / 2, acc);
        coop_barrier();
        if (tid < COOP_NT / 2) {                                    // (slot tid is read and rewritten by lane tid alone)
            bucket_add_fast<FP>(acc, img->load(tid));
            img->store(tid, acc);
        }
    }
    coop_barrier();
    coop_ctx<FP> c{ex, tid >> 6, tid & 63, 0};
    // (with fewer items than lanes the upper lanes hold infinity: the tree starts where there is something to add)
    const unsigned live = nitems / pc.nsub;
    coop_tree_sum<FP, COOP_NT / 2>(img, live >= COOP_NT / 2 ? COOP_NT / 4 : live / 2, c);
    xyzz_dev<FP> x;
    if (c.lane == 0) x = img->load(0); else x.set_inf();
    if (pc.b < m) {
        #pragma unroll 1
        for (unsigned k = 0; k < pc.b + lgG; k++) coop_dbl<FP>(x, c);
    }
    if (tid == 0) x.store(&parts[(size_t)w * (m * sb + sp) + q]);
}

// k_bucket_top_sum: the nparts <= 32 parts of a window (m + 1 sums, or their pieces), one work-group of four waves per window
// |fin| (nullable): the window sum also in the reference's wire image (what k_finalize would write), coordinate r by wave r --
// the small MSMs save that launch
// |flag_src| (nullable): the piece tree's "a bucket was beyond the tree" word is handed over with the sums -- copied to
// |flag_dst| (the word behind fin[nwins - 1]: one device-to-host copy brings both) and CLEARED, so that the next MSM finds it
// zero without a memset of its own (msm_driver.hpp)
template<class FP>
__global__ __launch_bounds__(COOP_NT)
void k_bucket_top_sum_coop(xyzz_mem<FP::N>* __restrict__ out, const xyzz_mem<FP::N>* __restrict__ parts, unsigned nparts,
                           xyzz_mem<FP::NW>* __restrict__ fin, u32* __restrict__ flag_src, u32* __restrict__ flag_dst)
{
    if (flag_src != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { *flag_dst = *flag_src; *flag_src = 0; }
    __shared__ coop_lds<FP> ex;
    __shared__ coop_img<FP, 32> img;
    const unsigned w = blockIdx.x, tid = threadIdx.x;
    if (tid < 32) img.store(tid, bucket_top_sum_gather<FP>(parts, nparts - 1, w, tid));
    coop_barrier();
    coop_ctx<FP> c{&ex, tid >> 6, tid & 63, 0};
    unsigned s0 = 1;
    while (2 * s0 < nparts) s0 <<= 1;
    coop_tree_sum<FP, 32>(&img, s0, c);
    if (tid == 0) img.load(0).store(&out[w]);
    if (fin != nullptr && c.lane == 0) {                            // (xyzz_dev::store_std, a coordinate per wave)
        const xyzz_dev<FP> r = img.load(0);
        const FP& f = c.role == 0 ? r.X : c.role == 1 ? r.Y : c.role == 2 ? r.ZZZ : r.ZZ;
        u32 s[FP::NW];
        if (r.is_inf()) { for (int i = 0; i < FP::NW; i++) s[i] = 0; }
        else f.to_std(s);
        u32* d = reinterpret_cast<u32*>(&fin[w]) + c.role * FP::NW;
        for (int i = 0; i < FP::NW; i++) d[i] = s[i];
    }
}

} // namespace sppark_amd

namespace sppark_amd {

// ---------------------------------------------------------------------------
// The record tree (msm_kernels.hpp reduce_runs_chunk) with its additions in cooperative form: a work-group is 64 work
// items x four waves.  Below ~2^19 points the buckets are longer than k_join_runs' walk and the tree does the work: at
// 2^16 points eight levels of ONE addition each (every level halves the list), 27-35 us per level with one wave per
// addition (profiles/r04_msm_timeline_2p16_before.txt).  From the level with <= COOP_TREE_MAX work items on -- one
// work-group per CU -- the cooperative addition (9 us against 16, profiles/r04_chain_bench.log) sets the pace.
// Same record semantics as reduce_runs_chunk; the four copies of a work item take the same decisions, wave 0 stores.
// The addition at step r is skipped by the whole work-group when no work item has one (__syncthreads_or): with the
// [run, NONE, run, NONE] pattern k_accumulate leaves, every second step.
// ---------------------------------------------------------------------------
static constexpr unsigned COOP_TREE_MAX = 16384;

template<class FP>
SPPARK_DEVFN void reduce_runs_coop_item(xyzz_mem<FP::N>* buckets, u32* out_key, xyzz_mem<FP::N>* out_pt,
                                        const u32* in_key, const xyzz_mem<FP::N>* in_pt,
                                        unsigned nrec, unsigned F, unsigned nthreads, int last, unsigned t, coop_ctx<FP>& c)
{
    const bool live = t < nthreads, writer = c.role == 0;
    const unsigned lo = t * F, hi = !live ? 0 : (nrec < lo + F ? nrec : lo + F);
    const size_t rec0 = (size_t)t * 2;
    xyzz_dev<FP> acc; acc.set_inf();
    u32 cur = KEY_NONE, slot0_key = KEY_NONE;
    bool first_run = true;
    for (unsigned s = 0; s < F; s++) {                              // uniform trip count: barriers inside
        const unsigned r = lo + s;
        const u32 k = live && r < hi ? in_key[r] : KEY_NONE;
        const bool add = k != KEY_NONE && k == cur;
        if (k != KEY_NONE && !add) {                                // a new run: the finished one leaves
            if (cur != KEY_NONE) {
                if (writer) {
                    if (first_run && !last) acc.store(&out_pt[rec0]);
                    else                    acc.store(&buckets[cur]);
                }
                if (first_run && !last) slot0_key = cur;
                first_run = false;
            }
            cur = k;
            acc = xyzz_dev<FP>::load(&in_pt[r]);
        }
        if (coop_any(add)) {
            xyzz_dev<FP> y;
            if (add) y = xyzz_dev<FP>::load(&in_pt[r]); else y.set_inf();
            coop_add<FP>(acc, y, c);                                // (an operand at infinity leaves acc as it is)
        }
    }
    if (!live || !writer) return;
    if (last) {
        if (cur != KEY_NONE) acc.store(&buckets[cur]);
        return;
    }
    if (cur == KEY_NONE)      { out_key[rec0] = KEY_NONE; out_key[rec0 + 1] = KEY_NONE; }
    else if (first_run)       { acc.store(&out_pt[rec0]); out_key[rec0] = cur; out_key[rec0 + 1] = KEY_NONE; }
    else                      { acc.store(&out_pt[rec0 + 1]); out_key[rec0] = slot0_key; out_key[rec0 + 1] = cur; }
}

template<class FP>
__global__ __launch_bounds__(COOP_NT)
void k_reduce_runs_coop(xyzz_mem<FP::N>* __restrict__ buckets,
                        u32* __restrict__ out_key, xyzz_mem<FP::N>* __restrict__ out_pt,
                        const u32* __restrict__ in_key, const xyzz_mem<FP::N>* __restrict__ in_pt,
                        unsigned nrec, unsigned F, unsigned nthreads, int last, const u32* __restrict__ skip)
{
    if (skip != nullptr && *skip == 0) return;
    __shared__ coop_lds<FP> ex;
    coop_ctx<FP> c{&ex, threadIdx.x >> 6, threadIdx.x & 63, 0};
    reduce_runs_coop_item<FP>(buckets, out_key, out_pt, in_key, in_pt, nrec, F, nthreads, last, blockIdx.x * 64 + c.lane, c);
}

// the narrow end (<= 64 work items): every remaining level in one launch, as k_reduce_tail
template<class FP>
__global__ __launch_bounds__(COOP_NT)
void k_reduce_tail_coop(xyzz_mem<FP::N>* __restrict__ buckets, u32* key0, xyzz_mem<FP::N>* pt0, u32* key1, xyzz_mem<FP::N>* pt1,
                        unsigned nrec, unsigned F, const u32* __restrict__ skip)
{
    if (skip != nullptr && *skip == 0) return;
    __shared__ coop_lds<FP> ex;
    coop_ctx<FP> c{&ex, threadIdx.x >> 6, threadIdx.x & 63, 0};
    u32 *ik = key0, *ok = key1; xyzz_mem<FP::N> *ip = pt0, *op = pt1;
    for (;;) {
        const unsigned nthreads = (nrec + F - 1) / F;               // <= 64
        const int last = nthreads == 1;
        reduce_runs_coop_item<FP>(buckets, ok, op, ik, ip, nrec, F, nthreads, last, c.lane, c);
        if (last) break;
#if defined(__HIP_DEVICE_COMPILE__)
        __threadfence_block();                                      // (see k_reduce_tail)
#endif
        coop_barrier();
        nrec = 2 * nthreads;
        u32* tk = ik; ik = ok; ok = tk;
        xyzz_mem<FP::N>* tp = ip; ip = op; op = tp;
    }
}

} // namespace sppark_amd

namespace sppark_amd {

// ---------------------------------------------------------------------------
// The chunked bucket-sum levels (msm_kernels.hpp bucket_level1_item / bucket_levelN_item) for grids of at most one
// work-group of four waves per CU (<= COOP_LEVEL_MAX work items: MSMs of <= 2^16 points, and the last chunked level of
// larger ones): the same running sums, every addition and doubling by four waves.  64 work items per work-group.
// ---------------------------------------------------------------------------
// (24576 or 32768 -- 1.5 or 2 rounds of work-groups -- change nothing measurable: profiles/r04_msm_coop_level_max.log)
static constexpr unsigned COOP_LEVEL_MAX = 16384;

template<class FP>
__global__ __launch_bounds__(COOP_NT)
void k_bucket_level1_coop(xyzz_mem<FP::N>* __restrict__ A, xyzz_mem<FP::N>* __restrict__ Wt,
                          const xyzz_mem<FP::N>* __restrict__ buckets, unsigned NB, unsigned K, unsigned nwins,
                          const u32* __restrict__ off)
{
    __shared__ coop_lds<FP> ex;
    coop_ctx<FP> c{&ex, threadIdx.x >> 6, threadIdx.x & 63, 0};
    const unsigned nchunks = NB / K;
    const size_t id = (size_t)blockIdx.x * 64 + c.lane;
    const bool live = id < (size_t)nwins * nchunks;
    const unsigned w = live ? (unsigned)(id / nchunks) : 0, u = live ? (unsigned)(id % nchunks) : 0;
    const xyzz_mem<FP::N>* row = buckets + (size_t)w * NB + (size_t)u * K;
    const u32* o = off ? off + (size_t)w * (NB + 1) + (size_t)u * K : nullptr;
    xyzz_dev<FP> acc, ret;
    if (live) acc = bucket_load<FP>(row, o, K - 1); else acc.set_inf();
    ret = acc;
    #pragma unroll 1
    for (unsigned j = K - 1; j--;) {
        xyzz_dev<FP> y;
        if (live) y = bucket_load<FP>(row, o, j); else y.set_inf();
        coop_add<FP>(acc, y, c);
        coop_add<FP>(ret, acc, c);
    }
    if (live && c.role == 0) { acc.store(&A[id]); ret.store(&Wt[id]); }
}

// The first level for grids between the cooperative form and one resident round of waves (COOP_LEVEL_MAX < work items <=
// LAT_LANES: 2^17 ... 2^21 points), its two chains on TWO WAVES: acc_j = acc_(j+1) + B_j and ret = sum_j acc_j are 2 (K - 1)
// dependent additions for one lane (k_bucket_level1_lat: 0.27 ms of a 1.4 ms MSM at 2^18 points with 544 waves on 1024 SIMDs),
// but ret only ever needs the acc of the step before: wave 0 of a 128-lane work-group runs the acc chain of 64 work items and
// leaves every acc in LDS (two images, alternating), wave 1 adds the previous one to ret in the same step -- K + 1 additions deep.
template<class FP>
__global__ __launch_bounds__(128, 2)
void k_bucket_level1_pipe(xyzz_mem<FP::N>* __restrict__ A, xyzz_mem<FP::N>* __restrict__ Wt,
                          const xyzz_mem<FP::N>* __restrict__ buckets, unsigned NB, unsigned K, unsigned nwins,
                          const u32* __restrict__ off)
{
    __shared__ coop_img<FP, 64> img[2];
    const unsigned role = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned nchunks = NB / K;
    const size_t id = (size_t)blockIdx.x * 64 + lane;
    const bool live = id < (size_t)nwins * nchunks;
    const unsigned w = live ? (unsigned)(id / nchunks) : 0, u = live ? (unsigned)(id % nchunks) : 0;
    const xyzz_mem<FP::N>* row = buckets + (size_t)w * NB + (size_t)u * K;
    const u32* o = off ? off + (size_t)w * (NB + 1) + (size_t)u * K : nullptr;
    xyzz_dev<FP> v; v.set_inf();                                    // wave 0: acc, wave 1: ret
    // step s: wave 0 brings acc to acc_(K-1-s) and leaves it in img[s & 1]; wave 1 takes acc_(K-s) from img[(s - 1) & 1]
    #pragma unroll 1
    for (unsigned s = 0; s < K; s++) {
        if (role == 0) {
            xyzz_dev<FP> y;
            if (live) y = bucket_load<FP>(row, o, K - 1 - s); else y.set_inf();
            if (s == 0) v = y; else bucket_add_fast<FP>(v, y);
            img[s & 1].store(lane, v);
        } else if (s >= 1) {
            const xyzz_dev<FP> y = img[(s - 1) & 1].load(lane);
            if (s == 1) v = y; else bucket_add_fast<FP>(v, y);
        }
        coop_barrier();
    }
    if (role == 1) {
        const xyzz_dev<FP> y = img[(K - 1) & 1].load(lane);         // acc_0
        if (K == 1) v = y; else bucket_add_fast<FP>(v, y);
    }
    if (live) v.store(role == 0 ? &A[id] : &Wt[id]);
}

// ... and the later chunked levels likewise on THREE waves (k_bucket_levelN_lat is 3 (K - 1) + lgG + 2 operations deep: 2^23
// points 0.53 ms on 57 344 lanes): wave 0 runs acc_j = acc_(j+1) + A_j and leaves every acc_j (j >= 1) in LDS, wave 1 adds the
// one of the step before to r and then doubles r lgG times, wave 2 sums the Wt's meanwhile and takes r at the end --
// K + lgG + 1 operations deep.
template<class FP>
__global__ __launch_bounds__(192, 2)
void k_bucket_levelN_pipe(xyzz_mem<FP::N>* __restrict__ A2, xyzz_mem<FP::N>* __restrict__ Wt2,
                          const xyzz_mem<FP::N>* __restrict__ A1, const xyzz_mem<FP::N>* __restrict__ Wt1,
                          unsigned nitems, unsigned K, unsigned lgG, unsigned nwins)
{
    __shared__ coop_img<FP, 64> img[2];
    const unsigned role = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned nchunks = nitems / K;
    const size_t id = (size_t)blockIdx.x * 64 + lane;
    const bool live = id < (size_t)nwins * nchunks;
    const unsigned w = live ? (unsigned)(id / nchunks) : 0, u = live ? (unsigned)(id % nchunks) : 0;
    const size_t base = (size_t)w * nitems + (size_t)u * K;
    xyzz_dev<FP> v; v.set_inf();                                    // wave 0: acc, wave 1: r, wave 2: sw
    #pragma unroll 1
    for (unsigned s = 0; s < K; s++) {
        if (role == 0) {                                            // acc_(K-1-s)
            xyzz_dev<FP> y;
            if (live) y = xyzz_dev<FP>::load(&A1[base + K - 1 - s]); else y.set_inf();
            if (s == 0) v = y; else bucket_add_fast<FP>(v, y);
            if (s + 1 < K) img[s & 1].store(lane, v);               // (acc_0 is not part of r)
        } else if (role == 1) {                                     // r += acc_(K-s), left by the step before
            if (s >= 1) {
                const xyzz_dev<FP> y = img[(s - 1) & 1].load(lane);
                if (s == 1) v = y; else bucket_add_fast<FP>(v, y);
            }
        } else {                                                    // sw += Wt_s
            xyzz_dev<FP> y;
            if (live) y = xyzz_dev<FP>::load(&Wt1[base + s]); else y.set_inf();
            if (s == 0) v = y; else bucket_add_fast<FP>(v, y);
        }
        coop_barrier();
    }
    if (role == 1) {
        #pragma unroll 1
        for (unsigned k = 0; k < lgG; k++) bucket_dbl_fast<FP>(v);
        img[0].store(lane, v);
    }
    coop_barrier();
    if (role == 2) bucket_add_fast<FP>(v, img[0].load(lane));
    if (live && role == 0) v.store(&A2[id]);
    if (live && role == 2) v.store(&Wt2[id]);
}

template<class FP>
__global__ __launch_bounds__(COOP_NT)
void k_bucket_levelN_coop(xyzz_mem<FP::N>* __restrict__ A2, xyzz_mem<FP::N>* __restrict__ Wt2,
                          const xyzz_mem<FP::N>* __restrict__ A1, const xyzz_mem<FP::N>* __restrict__ Wt1,
                          unsigned nitems, unsigned K, unsigned lgG, unsigned nwins)
{
    __shared__ coop_lds<FP> ex;
    coop_ctx<FP> c{&ex, threadIdx.x >> 6, threadIdx.x & 63, 0};
    const unsigned nchunks = nitems / K;
    const size_t id = (size_t)blockIdx.x * 64 + c.lane;
    const bool live = id < (size_t)nwins * nchunks;
    const unsigned w = live ? (unsigned)(id / nchunks) : 0, u = live ? (unsigned)(id % nchunks) : 0;
    const size_t base = (size_t)w * nitems + (size_t)u * K;
    xyzz_dev<FP> acc, r, sw, y;
    acc.set_inf(); r.set_inf();
    if (live) sw = xyzz_dev<FP>::load(&Wt1[base]); else sw.set_inf();
    #pragma unroll 1
    for (unsigned j = K - 1; j >= 1; j--) {
        if (live) y = xyzz_dev<FP>::load(&A1[base + j]); else y.set_inf();
        coop_add<FP>(acc, y, c);
        coop_add<FP>(r, acc, c);
        if (live) y = xyzz_dev<FP>::load(&Wt1[base + j]); else y.set_inf();
        coop_add<FP>(sw, y, c);
    }
    if (live) y = xyzz_dev<FP>::load(&A1[base]); else y.set_inf();
    coop_add<FP>(acc, y, c);
    #pragma unroll 1
    for (unsigned k = 0; k < lgG; k++) coop_dbl<FP>(r, c);
    coop_add<FP>(sw, r, c);
    if (live && c.role == 0) { acc.store(&A2[id]); sw.store(&Wt2[id]); }
}

} // namespace sppark_amd

namespace sppark_amd {

// A level of the piece tree (msm_piece_kernels.hpp) with four waves per addition: the levels of at most COOP_LEVEL_MAX work
// items (one work-group per CU).  The four copies of a work item take the same decisions and write the same keys; wave 0
// stores the point.  (The in-place update is safe: every wave has loaded both operands before coop_add's first barrier.)
template<class FP>
__global__ __launch_bounds__(COOP_NT)
void k_piece_level_coop(xyzz_mem<FP::N>* __restrict__ buckets, u32* __restrict__ rec_key, xyzz_mem<FP::N>* rec_pt,
                        const u32* __restrict__ off, unsigned NB, unsigned L, unsigned chunks_per_win, unsigned nwins,
                        unsigned cmax, unsigned t, unsigned last, u32* __restrict__ any_long)
{
    __shared__ coop_lds<FP> ex;
    coop_ctx<FP> c{&ex, threadIdx.x >> 6, threadIdx.x & 63, 0};
    const piece_job j = piece_job_of(rec_key, off, NB, L, chunks_per_win, nwins, cmax, t, last, any_long, (size_t)blockIdx.x * 64 + c.lane);
    const bool work = j.live && (j.add || j.finish), add = j.live && j.add;
    xyzz_dev<FP> x, y;
    if (work) x = xyzz_dev<FP>::load(&rec_pt[j.dst]); else x.set_inf();
    if (add)  y = xyzz_dev<FP>::load(&rec_pt[j.src]); else y.set_inf();
    if (coop_any(add)) coop_add<FP>(x, y, c);
    if (work && c.role == 0) x.store(j.finish ? &buckets[j.B] : &rec_pt[j.dst]);
}


// Every level of the piece tree from t0 on in ONE launch: the levels of <= PIECE_FUSE_MAX work items are ~15 us each as
// launches of their own (2^12 points: eight of them, 0.14 ms of a 0.36 ms MSM) although the cooperative addition of a level
// is ~4 us -- the rest is the launch, the offsets' dependent loads and the round trip of the operands.  Here a work-group
// owns 2^lgGB buckets with all their pairs (piece_tail_lgGB), so a level waits for nothing but the work-group's own stores
// (the barrier's work-group-scope release / acquire: the records stay in global memory, same slots as the per-level form).
// Same box, wall (profiles/r06_msm_piece_tail_ab.log): 2^10 0.426 -> 0.406 ms, 2^12 0.482 -> 0.463, 2^14 0.550 -> 0.541, 2^16 0.800 -> 0.788;
// from 2^16 work items on the lane-per-addition launches are faster (2^14: 0.585 with the levels of 2^16 items in here).
static constexpr size_t PIECE_FUSE_MAX = 32768;
template<class FP>
__global__ __launch_bounds__(COOP_NT, 2)         // (two work-groups per CU: 2^12 points are 512 work-groups)
void k_piece_tail_coop(xyzz_mem<FP::N>* __restrict__ buckets, u32* rec_key, xyzz_mem<FP::N>* rec_pt,
                       const u32* __restrict__ off, unsigned NB, unsigned L, unsigned chunks_per_win, unsigned nwins,
                       unsigned cmax, unsigned t0, unsigned lgGB, u32* __restrict__ any_long)
{
    __shared__ coop_lds<FP> ex;
    coop_ctx<FP> c{&ex, threadIdx.x >> 6, threadIdx.x & 63, 0};
    const size_t nb = (size_t)nwins * NB, B0 = (size_t)blockIdx.x << lgGB;
    #pragma unroll 1
    for (unsigned t = t0; (cmax >> (t + 1)) >= 1; t++) {
        const unsigned last = (cmax >> (t + 2)) == 0, njobs = (cmax >> (t + 1)) << lgGB;
        #pragma unroll 1
        for (unsigned base = 0; base < njobs; base += 64) {
            const unsigned idx = base + c.lane;
            const size_t B = B0 + (idx & ((1u << lgGB) - 1));
            piece_job j; j.live = j.add = j.finish = false; j.dst = j.src = 0; j.B = 0;
            if (idx < njobs && B < nb) j = piece_job_bm(rec_key, off, NB, L, chunks_per_win, cmax, t, last, any_long, B, idx >> lgGB);
            const bool work = j.live && (j.add || j.finish), add = j.live && j.add;
            xyzz_dev<FP> x, y;
            if (work) x = xyzz_dev<FP>::load(&rec_pt[j.dst]); else x.set_inf();
            if (add)  y = xyzz_dev<FP>::load(&rec_pt[j.src]); else y.set_inf();
            if (coop_any(add)) coop_add<FP>(x, y, c);
            if (work && c.role == 0) x.store(j.finish ? &buckets[j.B] : &rec_pt[j.dst]);
        }
        coop_barrier();
    }
}

} // namespace sppark_amd

namespace sppark_amd {

// ---------------------------------------------------------------------------
// The bucket sums of a SMALL window (NB = 2^m <= 256 buckets: MSMs of up to 2^16 points) straight from the buckets:
//     sum_{j = 1 .. NB} j B_{j-1}  =  sum_{b = 0 .. m} 2^b S_b,      S_b = sum of the buckets whose 1-BASED number has bit b set.
// Work-group (b, w) gathers the NB / 2 buckets of S_b (b = m: the single bucket number NB) -- the buckets without entries are
// taken as infinity from the sort's offsets, which is all the first chunked level did for such windows -- sums them by the
// cooperative tree, doubles b times and leaves part b; k_bucket_top_sum_coop adds the m + 1 parts of a window as before.
// Against k_bucket_level1_coop + k_bucket_top_bits_coop: one launch less, no plain-sum part of twice the items (the 0-based
// form carries sum_j B_j as a part of its own), a tree of log2(NB / 2) levels instead of log2(NB).
// 2^12 points (8 buckets): 0.142 -> 0.07 ms of bucket sums; 2^16 (128 buckets): 0.19 -> 0.17 (profiles/r06_msm_small_sums_ab.log).
// ---------------------------------------------------------------------------
template<class FP>
__global__ __launch_bounds__(COOP_NT)
void k_bucket_small_bits_coop(xyzz_mem<FP::N>* __restrict__ parts, const xyzz_mem<FP::N>* __restrict__ buckets,
                              const u32* __restrict__ off, unsigned NB, unsigned m)
{
    __shared__ coop_lds<FP> ex;
    __shared__ coop_img<FP, SMALL_SUMS_MAX_NB / 2> img;
    const unsigned b = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
    const unsigned cnt = b >= m ? 1u : NB / 2;
    if (tid < SMALL_SUMS_MAX_NB / 2) img.store(tid, small_sums_gather<FP>(buckets, off, NB, m, b, w, tid));
    coop_barrier();
    coop_ctx<FP> c{&ex, tid >> 6, tid & 63, 0};
    if (cnt > 1) coop_tree_sum<FP, SMALL_SUMS_MAX_NB / 2>(&img, cnt / 2, c);
    xyzz_dev<FP> x;
    if (c.lane == 0) x = img.load(0); else x.set_inf();
    #pragma unroll 1
    for (unsigned k = 0; k < b; k++) coop_dbl<FP>(x, c);
    if (tid == 0) x.store(&parts[(size_t)w * (m + 1) + b]);
}

} // namespace sppark_amd

