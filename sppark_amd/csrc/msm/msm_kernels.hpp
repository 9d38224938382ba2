// MI355X Pippenger MSM kernels (bucket method with signed windows).
//
// Same mathematical pipeline as the reference's GPU path
// (msm/pippenger.cuh:72-296 breakdown/accumulate/integrate, msm/sort.cuh,
// msm/batch_addition.cuh) but laid out for CDNA4 rather than translated:
//
//   breakdown     scalars -> signed digits, one u32 per (window, point)
//   hist          per (window, point-slab) bucket histogram built ENTIRELY in
//                 LDS (2^(c-1) counters = 128 KB at c = 16 fit the 160 KB LDS),
//                 stored with plain coalesced writes -- no global atomics
//   scan_*        slab-exclusive and bucket-exclusive prefix sums
//   scatter       counting-sort scatter with LDS cursors (ds_add_rtn), output
//                 = point indices grouped by bucket, sign in bit 31
//   accumulate    the hot kernel: every lane walks a FIXED-length run of the
//                 grouped index list (so all 64 lanes of a wave do the same
//                 number of mixed additions no matter how skewed the scalars
//                 are), gathers the affine points, keeps the running XYZZ sum
//                 in registers and flushes at bucket boundaries
//   reduce_runs   the same walk over the (key, partial sum) records the
//                 previous level left at its chunk boundaries, until one
//                 record is left: a segmented tree reduction whose cost is
//                 independent of the bucket-size distribution
//   bucket_*      per-window weighted bucket sum  sum_b (b+1)*B_b  by chunked
//                 running sums, log-depth
//
// The reference instead assigns one thread per bucket with dynamic work
// stealing through a device-global atomic counter and a cooperative grid sync
// (pippenger.cuh:157-223), sized for 32-lane warps and <= 100 SMs (sort.cuh:312).
#pragma once
#include "../ec/xyzz_dev.hpp"
#include "../ec/xyzzx_dev.hpp"

namespace sppark_amd {

static constexpr u32 KEY_NONE = 0xffffffffu;

// ---------------------------------------------------------------------------
// Every kernel is a thin __global__ wrapper around a per-work-item body
// (SPPARK_DEVFN = __device__ in the shipped build).  The bodies are also what
// the host-emulation test harness (tests/emu/, -DSPPARK_HOST_EMULATION) calls.
// ---------------------------------------------------------------------------

// Digit recoding (any recoding with the same sum is valid; only the group
// element is observable -- SURVEY Appendix A.6):
//   s > (r-1)/2  ->  s = r - s, all signs flipped   (cf. pippenger.cuh:96-99)
//   d_w = bits of window w + carry;  d_w > 2^(len_w-1)  ->  d_w -= 2^len_w, carry = 1
// so |d_w| <= 2^(len_w-1) and bucket index = |d_w| - 1  (cf. sort.cuh:92).
// The nbits scalar bits are split EVENLY over the windows (len_w = base or
// base+1): no window is left with only a few significant bits, which would put
// every point of that window into a handful of buckets (the reference spreads
// its short top window with an lshift trick instead, sort.cuh:92,111,339).
// Output word: bit31 = sign, low bits = |digit|, 0 = no contribution.
// |limb(k)| returns 32-bit limb k of the reduced magnitude (k <= N, limb N = 0).
__host__ __device__ inline unsigned window_len(unsigned w, unsigned nwins, unsigned nbits)   // also used by the host Horner
{   return nbits / nwins + (w < nbits % nwins ? 1u : 0u);   }

// Only the digits of windows [w_begin, w_begin + w_count) are stored (a window group of the
// pipelined driver), at digits[(w - w_begin) * n + i]; the carry still runs through all
// lower windows.
template<class LimbFn>
SPPARK_DEVFN void recode_digits(u32* digits, size_t n, size_t i, LimbFn limb, bool flip,
                                unsigned nwins, unsigned nbits, unsigned w_begin = 0, unsigned w_count = ~0u)
{
    u32 carry = 0;
    const unsigned w_end = w_count > nwins - w_begin ? nwins : w_begin + w_count;
    for (unsigned w = 0, bit = 0; w < w_end; w++) {
        const unsigned len = window_len(w, nwins, nbits);
        const u32 half = 1u << (len - 1), full = 1u << len, mask = full - 1;
        const unsigned li = bit >> 5, sh = bit & 31;
        bit += len;
        u64 two = limb(li) | ((u64)limb(li + 1) << 32);
        u32 d = ((u32)(two >> sh) & mask) + carry;
        bool minus = d > half;
        carry = minus;
        d = minus ? full - d : d;
        if (w >= w_begin) digits[(size_t)(w - w_begin) * n + i] = d ? (d | ((u32)(minus != flip) << 31)) : 0;
    }
}

// load scalar i, leave Montgomery form if asked, fold into [0, (r-1)/2]
template<class FR>
SPPARK_DEVFN FR load_scalar_abs(const u32* scalars, size_t i, int mont, bool& flip)
{
    constexpr int N = FR::N;
    FR s;
    const uint4* src = reinterpret_cast<const uint4*>(scalars + i * N);
    #pragma unroll
    for (int k = 0; k < N / 4; k++) {
        uint4 q = src[k];
        s.v[4*k] = q.x; s.v[4*k+1] = q.y; s.v[4*k+2] = q.z; s.v[4*k+3] = q.w;
    }
    if (mont) s = s.from();
    FR neg = FR::modulus_minus(s);                      // r - s
    u32 bw = 0;
    #pragma unroll
    for (int k = 0; k < N; k++) (void)__builtin_subc(neg.v[k], s.v[k], bw, &bw);
    flip = bw != 0;                                     // r - s < s  <=>  s > (r-1)/2
    return FR::select(flip, neg, s);
}

template<class FR>
__global__ __launch_bounds__(256)
void k_breakdown(u32* __restrict__ digits, const u32* __restrict__ scalars,
                 unsigned n, unsigned nwins, unsigned nbits, int mont, unsigned w_begin, unsigned w_count)
{
    constexpr int N = FR::N;
    __shared__ u32 limbs[N + 2][256];                   // transposed: dynamic limb index without scratch
    const unsigned tid = threadIdx.x;
    for (unsigned i = blockIdx.x * 256 + tid; i < n; i += gridDim.x * 256) {
        bool flip;
        FR s = load_scalar_abs<FR>(scalars, i, mont, flip);
        #pragma unroll
        for (int k = 0; k < N; k++) limbs[k][tid] = s.v[k];
        limbs[N][tid] = 0; limbs[N + 1][tid] = 0;
        recode_digits(digits, n, i, [&](unsigned k) { return limbs[k][tid]; }, flip, nwins, nbits, w_begin, w_count);
    }
}

// ---------------------------------------------------------------------------
// accumulate (level 0).  Work item (chunk, window) owns entries
// [chunk*L, chunk*L + L) of window w's grouped list.  Runs that touch the
// chunk's ends are handed to the next level as (key, sum) records (slot 0 =
// first run, slot 1 = last run); runs strictly inside are complete buckets and
// are stored straight into buckets[key].
// ---------------------------------------------------------------------------
template<class FP, bool FLAGGED>
SPPARK_DEVFN void accumulate_chunk(xyzz_mem<FP::N>* buckets, u32* rec_key, xyzz_mem<FP::N>* rec_pt,
                                   const unsigned char* points, unsigned stride,
                                   const u32* sorted, const u32* off,
                                   unsigned n, unsigned NB, unsigned L, unsigned chunks_per_win,
                                   unsigned chunk, unsigned w_local, unsigned w_base = 0)
{
    // |sorted| / |off| hold the windows of one window group (index w_local); buckets, records and
    // keys are indexed by the window's number in the whole MSM (w)
    if (chunk >= chunks_per_win) return;
    const unsigned w = w_base + w_local;
    const size_t rec0 = ((size_t)w * chunks_per_win + chunk) * 2;
    const u32* o = off + (size_t)w_local * (NB + 1);
    const unsigned total = o[NB];
    unsigned p = chunk * L;
    if (p >= total) { rec_key[rec0] = KEY_NONE; rec_key[rec0 + 1] = KEY_NONE; return; }
    const unsigned end = total < p + L ? total : p + L;

    // bucket holding position p: o[b] <= p < o[b+1]
    unsigned lo = 0, hi = NB;                       // invariant o[lo] <= p < o[hi]
    while (hi - lo > 1) {
        unsigned mid = (lo + hi) >> 1;
        if (o[mid] <= p) lo = mid; else hi = mid;
    }
    unsigned b = lo, next = o[b + 1];

    const u32* src = sorted + (size_t)w_local * n;
    xyzz_dev<FP> acc;
    bool first_run = true;
    u32 slot0_key = KEY_NONE;

    u32 e = src[p];
    affine_dev<FP> pt = load_affine<FP, FLAGGED>(points, e & 0x7fffffffu, stride);
    acc.set(pt, e >> 31);
    // PREFETCH: the gather of entry p+1 is issued before the addition of entry p, so its latency
    // hides behind ~20k cycles of arithmetic.  It costs a second point in registers: worth it for
    // the reduced-radix field (2 waves/SIMD either way, +2 %), not for alt_bn128 (would drop from
    // 4 to 3 waves/SIMD: 53 -> 71 ms) or Fp2.
    // (Three waves per SIMD instead of two -- __launch_bounds__(256, 3), 168 registers -- were measured
    // with the compiler's allocation: 42 spilled registers without this prefetch, 93 with it, 29 with
    // single-chain products on top; 174 / 200 ms against 112 ms.  The prefetch itself is worth 1 %:
    // 113.4 vs 112.4 ms.  profiles/r02_msm_accumulate_waves_ab.log)
    constexpr bool PREFETCH = field_is_montx<FP>::value;
    u32 e_next = 0, e_next2 = 0;                    // the index list runs two entries ahead, so that the
    affine_dev<FP> pt_next = pt;                    // gather's address never waits for its own load
    if (PREFETCH && p + 1 < end) { e_next = src[p + 1]; pt_next = load_affine<FP, FLAGGED>(points, e_next & 0x7fffffffu, stride); }
    if (PREFETCH && p + 2 < end) e_next2 = src[p + 2];
    for (p++; p < end; p++) {
        if (PREFETCH) {
            e = e_next; pt = pt_next;
            if (p + 1 < end) { e_next = e_next2; pt_next = load_affine<FP, FLAGGED>(points, e_next & 0x7fffffffu, stride); }
            if (p + 2 < end) e_next2 = src[p + 2];
        } else {
            e = src[p];
            pt = load_affine<FP, FLAGGED>(points, e & 0x7fffffffu, stride);
        }
        if (p == next) {                            // bucket boundary: flush
            const u32 key = w * NB + b;
            if (first_run) { acc.store(&rec_pt[rec0]); slot0_key = key; first_run = false; }
            else           acc.store(&buckets[key]);
            do { b++; next = o[b + 1]; } while (p == next);
            acc.set(pt, e >> 31);
        } else {
            acc.madd(pt, e >> 31);
        }
    }
    const u32 key = w * NB + b;
    if (first_run) { acc.store(&rec_pt[rec0]); rec_key[rec0] = key; rec_key[rec0 + 1] = KEY_NONE; }
    else           { acc.store(&rec_pt[rec0 + 1]); rec_key[rec0] = slot0_key; rec_key[rec0 + 1] = key; }
}

// (Fp2: 256 VGPRs + 106 AGPRs, one wave per SIMD; forcing two spills 121 registers and gains nothing:
// 25.2 vs 25.6 ms for 2^22 G2 points)
// The 10-limb fields (alt_bn128, Pasta) need 172 registers unconstrained, four more than three waves
// per SIMD leave each: capped (4 spilled registers), alt_bn128 2^26: 68.4 -> 65.1 ms (A/B on one box,
// tools/gpu_r2_job23.sh).  The 14-limb fields stay at two waves: 230 registers, and every attempt at
// 168 spilled enough to lose (profiles/r02_msm_accumulate_waves_ab.log).
#ifndef SPPARK_G2_TWO_WAVES_MAX_N
// Fp2 over ten limbs (alt_bn128 G2): capped at 256 registers, 99 spilled, 184 B of scratch per lane, 26.0 -> 21.8 ms.
// Over fourteen limbs (BLS12-381 / -377 G2, -DSPPARK_G2_TWO_WAVES_MAX_N=28) the cap spills 688 registers into 852 B of
// scratch per lane: the 2^22 MSM gains (47.3 -> 42.1 ms) but 852 B x every resident lane is above the runtime's resident
// scratch limit, scratch is then set up per dispatch and every small MSM pays milliseconds (the G2 test suite: 8 s -> 123 s).
# define SPPARK_G2_TWO_WAVES_MAX_N 20
#endif
template<class FP, bool FLAGGED>
__global__ __launch_bounds__(256, (field_is_internal<FP>::value && FP::N <= 10) ? 3 : (field_is_internal<FP>::value && FP::N <= SPPARK_G2_TWO_WAVES_MAX_N) ? 2 : 1)
void k_accumulate(xyzz_mem<FP::N>* __restrict__ buckets,
                  u32* __restrict__ rec_key, xyzz_mem<FP::N>* __restrict__ rec_pt,
                  const unsigned char* __restrict__ points, unsigned stride,
                  const u32* __restrict__ sorted, const u32* __restrict__ off,
                  unsigned n, unsigned NB, unsigned L, unsigned chunks_per_win, unsigned w_base)
{
    accumulate_chunk<FP, FLAGGED>(buckets, rec_key, rec_pt, points, stride, sorted, off, n, NB, L,
                                  chunks_per_win, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y, w_base);
}

// ---------------------------------------------------------------------------
// Bitmap batch addition (msm/batch_addition.cuh:25-132): sum of the points whose bit is set.
// With a reference map the selection is the symmetric difference and a point that is only in
// the reference map is SUBTRACTED (bits ^= refs; refs &= bits, batch_addition.cuh:60-61), i.e.
//     sum_{i in bitmap \ refmap} P_i  -  sum_{i in refmap \ bitmap} P_i.
// The reference hands 32x32-bit chunks of the map to warps through a device-global atomic
// counter and shuffles the selected indices to the lanes; here every lane owns a FIXED span of
// |span_words| 32-bit words (so the number of additions per lane is bounded whatever the
// density), and leaves one (key 0, partial sum) record for the same segmented record tree the
// MSM uses (k_reduce_runs), whose last level stores the total into buckets[0].
// ---------------------------------------------------------------------------
template<class FP, bool FLAGGED>
SPPARK_DEVFN void bitmap_accumulate_item(u32* rec_key, xyzz_mem<FP::N>* rec_pt,
                                         const unsigned char* points, unsigned stride, unsigned npoints,
                                         const u32* bitmap, const u32* refmap, unsigned span_words, size_t t)
{
    const size_t nwords = ((size_t)npoints + 31) / 32;
    const size_t w0 = t * span_words;
    if (w0 >= nwords) return;
    xyzz_dev<FP> acc; acc.set_inf();
    for (size_t w = w0; w < w0 + span_words && w < nwords; w++) {
        u32 bits = bitmap[w], refs = refmap ? refmap[w] : 0;
        bits ^= refs; refs &= bits;
        if (w == nwords - 1 && (npoints & 31)) bits &= (1u << (npoints & 31)) - 1;     // bits past the last point
        while (bits) {
            const unsigned k = __builtin_ctz(bits);
            bits &= bits - 1;
            affine_dev<FP> p = load_affine<FP, FLAGGED>(points, w * 32 + k, stride);
            acc.madd(p, (refs >> k) & 1);
        }
    }
    acc.store(&rec_pt[t]); rec_key[t] = 0;
}
template<class FP, bool FLAGGED>
__global__ __launch_bounds__(256)
void k_bitmap_accumulate(u32* __restrict__ rec_key, xyzz_mem<FP::N>* __restrict__ rec_pt,
                         const unsigned char* __restrict__ points, unsigned stride, unsigned npoints,
                         const u32* __restrict__ bitmap, const u32* __restrict__ refmap, unsigned span_words)
{
    bitmap_accumulate_item<FP, FLAGGED>(rec_key, rec_pt, points, stride, npoints, bitmap, refmap, span_words,
                                        (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// ---------------------------------------------------------------------------
// join_runs: the SHORT segments of the record list, in one launch.
// The records k_accumulate leaves are the partial sums of the buckets that touch a chunk
// boundary: [first run | last run or NONE] per chunk.  Consecutive records with one key form a
// segment (the pieces of one bucket); with ~2^4..2^5 entries per bucket and 64..128 per chunk
// almost every segment is two records long -- the end of chunk t and the start of chunk t+1 --
// but the fan-in tree below resolves a segment only at the level where both pieces meet in one
// work item: ~12 dependent launches of a handful of latency-bound waves (1.2 ms of a 2^23-point
// MSM).  Here work item t looks at records 2t-1 and 2t: a record that STARTS a segment which ends
// within WALK records adds the segment up and stores the bucket; every record of such a segment
// gets key NONE in |out_key|; records of longer segments (skewed scalars) keep their key and go
// through the tree, which for uniform scalars then finds nothing to add.  Every record decides
// "short or long" for itself from the keys alone (at most 2*WALK reads of 4 bytes), so there is
// no ordering between work items.
//   two NONE records in a row only occur at the tail of a window (an empty chunk), and the next
//   window's keys are different: they end a segment like a different key does.
// ---------------------------------------------------------------------------
static constexpr unsigned JOIN_WALK = 8;
template<class FP> SPPARK_DEVFN void bucket_add(xyzz_dev<FP>& a, const xyzz_dev<FP>& b);
template<class FP> SPPARK_DEVFN void bucket_add_fast(xyzz_dev<FP>& a, const xyzz_dev<FP>& b);

template<class FP>
SPPARK_DEVFN void join_runs_item(xyzz_mem<FP::N>* buckets, u32* out_key, const u32* in_key,
                                 const xyzz_mem<FP::N>* in_pt, unsigned nrec, u32* any_long, size_t t)
{
    // pass 1 (keys only): classify the two records; a record that starts a short segment is a job [h, e]
    size_t job_h[2], job_e[2]; u32 job_k[2]; unsigned njobs = 0;
    for (unsigned s = 0; s < 2; s++) {
        if (t == 0 && s == 0) continue;
        const size_t q = 2 * t - 1 + s;
        if (q >= nrec) break;
        const u32 k = in_key[q];
        if (k == KEY_NONE) { out_key[q] = KEY_NONE; continue; }
        // the start of the segment: scan back
        size_t h = q; bool is_long = false;
        for (size_t i = q, none = 0; i-- > 0;) {
            const u32 ki = in_key[i];
            if (ki == KEY_NONE) { if (++none == 2) break; continue; }
            none = 0;
            if (ki != k) break;
            h = i;
            if (q - h > JOIN_WALK) { is_long = true; break; }
        }
        // its end: a different key (or two NONEs, or the end of the list) within WALK records of the start
        size_t e = h;
        if (!is_long) {
            bool ended = false;
            size_t i = h + 1;
            for (unsigned none = 0; i < nrec && i <= h + JOIN_WALK; i++) {
                const u32 ki = in_key[i];
                if (ki == KEY_NONE) { if (++none == 2) { ended = true; break; } continue; }
                none = 0;
                if (ki != k) { ended = true; break; }
                e = i;
            }
            if (i >= nrec) ended = true;
            is_long = !ended;
        }
        if (is_long) { out_key[q] = k; *any_long = 1; continue; }
        out_key[q] = KEY_NONE;
        if (q == h) { job_h[njobs] = h; job_e[njobs] = e; job_k[njobs] = k; njobs++; }
    }
    // pass 2: the sums.  ONE site for the addition, whichever of the two records is the start: the lanes of
    // a wave whose segment starts at the odd record and those whose segment starts at the even one run it
    // together (as two separate sites every wave paid for both: 1.1 ms instead of 0.5 at 2^23 points)
    for (unsigned j = 0; j < njobs; j++) {
        const size_t h = job_h[j], e = job_e[j]; const u32 k = job_k[j];
        xyzz_dev<FP> acc = xyzz_dev<FP>::load(&in_pt[h]);
        for (size_t i = h + 1; i <= e; i++)
            if (in_key[i] == k) bucket_add<FP>(acc, xyzz_dev<FP>::load(&in_pt[i]));
        acc.store(&buckets[k]);
    }
}

// ---------------------------------------------------------------------------
// reduce_runs (levels >= 1): same walk over F consecutive records.
// |last| = this is the final level (single work item): every run is complete.
// ---------------------------------------------------------------------------
template<class FP>
SPPARK_DEVFN void reduce_runs_chunk(xyzz_mem<FP::N>* buckets, u32* out_key, xyzz_mem<FP::N>* out_pt,
                                    const u32* in_key, const xyzz_mem<FP::N>* in_pt,
                                    unsigned nrec, unsigned F, unsigned nthreads, int last, unsigned t)
{
    if (t >= nthreads) return;
    const unsigned lo = t * F, hi = nrec < lo + F ? nrec : lo + F;
    const size_t rec0 = (size_t)t * 2;

    xyzz_dev<FP> acc;
    u32 cur = KEY_NONE, slot0_key = KEY_NONE;
    bool first_run = true;

    for (unsigned r = lo; r < hi; r++) {
        const u32 k = in_key[r];
        if (k == KEY_NONE) continue;
        if (k == cur) {
            bucket_add_fast<FP>(acc, xyzz_dev<FP>::load(&in_pt[r]));    // a handful of waves per level: latency, not registers
        } else {
            if (cur != KEY_NONE) {
                if (first_run && !last) { acc.store(&out_pt[rec0]); slot0_key = cur; }
                else                    acc.store(&buckets[cur]);
                first_run = false;
            }
            cur = k;
            acc = xyzz_dev<FP>::load(&in_pt[r]);
        }
    }
    if (last) {
        if (cur != KEY_NONE) acc.store(&buckets[cur]);
        return;
    }
    if (cur == KEY_NONE)      { out_key[rec0] = KEY_NONE; out_key[rec0 + 1] = KEY_NONE; }
    else if (first_run)       { acc.store(&out_pt[rec0]); out_key[rec0] = cur; out_key[rec0 + 1] = KEY_NONE; }
    else                      { acc.store(&out_pt[rec0 + 1]); out_key[rec0] = slot0_key; out_key[rec0 + 1] = cur; }
}

// |skip|: when non-null and *skip == 0 (k_join_runs found no long segment) the level has nothing to do
template<class FP>
__global__ __launch_bounds__(256)
void k_reduce_runs(xyzz_mem<FP::N>* __restrict__ buckets,
                   u32* __restrict__ out_key, xyzz_mem<FP::N>* __restrict__ out_pt,
                   const u32* __restrict__ in_key, const xyzz_mem<FP::N>* __restrict__ in_pt,
                   unsigned nrec, unsigned F, unsigned nthreads, int last, const u32* __restrict__ skip)
{
    if (skip != nullptr && *skip == 0) return;
    reduce_runs_chunk<FP>(buckets, out_key, out_pt, in_key, in_pt, nrec, F, nthreads, last,
                          blockIdx.x * blockDim.x + threadIdx.x);
}

// The narrow end of the tree in ONE launch: from the level with <= REDUCE_TAIL_NT work items on, one work-group runs
// the remaining levels with a barrier between them (a launch boundary costs ~6 us, and below 2^18 points, where the
// buckets are longer than the join's walk, nine of these levels are real work of one addition each).  Record buffers
// ping-pong as in the launch-per-level loop; |buf0| holds the input of the first level run here.
// (256 lanes = one wave per SIMD, like the other cold kernels: a 1024-lane work-group would cap the kernel at 128
// registers -- 187 spilled for the 14-limb field, 3x slower per addition -- and, over Fp2, call the outlined addition,
// which is compiled for up to 512, from a kernel that owns 128: that build hung the G2 tests)
static constexpr unsigned REDUCE_TAIL_NT = 256;
template<class FP>
__global__ __launch_bounds__(REDUCE_TAIL_NT)
void k_reduce_tail(xyzz_mem<FP::N>* __restrict__ buckets, u32* key0, xyzz_mem<FP::N>* pt0, u32* key1, xyzz_mem<FP::N>* pt1,
                   unsigned nrec, unsigned F, const u32* __restrict__ skip)
{
    if (skip != nullptr && *skip == 0) return;
    u32 *ik = key0, *ok = key1; xyzz_mem<FP::N> *ip = pt0, *op = pt1;
    for (;;) {
        const unsigned nthreads = (nrec + F - 1) / F;
        const int last = nthreads == 1;
        reduce_runs_chunk<FP>(buckets, ok, op, ik, ip, nrec, F, nthreads, last, threadIdx.x);
        if (last) break;
        // this level's records are read back by OTHER lanes of the work-group: make the global stores visible at
        // work-group scope before the barrier (a barrier alone orders them only while all waves share one L1,
        // i.e. not under the tgsplit / CU-split launch modes)
        __threadfence_block();
        __syncthreads();
        nrec = 2 * nthreads;
        u32* tk = ik; ik = ok; ok = tk;
        xyzz_mem<FP::N>* tp = ip; ip = op; op = tp;
    }
}

template<class FP>
__global__ __launch_bounds__(256, 2)
void k_join_runs(xyzz_mem<FP::N>* __restrict__ buckets, u32* __restrict__ out_key, const u32* __restrict__ in_key,
                 const xyzz_mem<FP::N>* __restrict__ in_pt, unsigned nrec, u32* __restrict__ any_long)
{
    join_runs_item<FP>(buckets, out_key, in_key, in_pt, nrec, any_long, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// ---------------------------------------------------------------------------
// bucket reduction, level 1: work item = (window, chunk of K buckets)
//   A[u]  = sum_j B[uK+j]          Wt[u] = sum_j (j+1) * B[uK+j]
// (running-sum trick of msm/pippenger.hpp:40-56 / pippenger.cuh:225-296)
// Full additions in the bucket-sum levels.  Over Fp2 (G2) one inlined add is ~70 base-field
// product bodies; the five call sites of bucket_levelN_item would be a 0.9 MB kernel that
// takes minutes to compile, for a phase that is a few percent of an MSM.  Wide coordinate
// fields therefore call ONE outlined copy (operands travel through scratch memory).
#if defined(SPPARK_HOST_EMULATION)
# define SPPARK_OUTLINED inline
#else
# define SPPARK_OUTLINED __device__ __noinline__
#endif
template<class FP> SPPARK_OUTLINED void xyzz_add_outlined(xyzz_dev<FP>& a, const xyzz_dev<FP>& b) { a.add(b); }
template<class FP> SPPARK_OUTLINED void xyzz_dbl_outlined(xyzz_dev<FP>& a) { a.dbl(); }
template<class FP> SPPARK_DEVFN void bucket_add(xyzz_dev<FP>& a, const xyzz_dev<FP>& b)
{   if constexpr (FP::N > 16) xyzz_add_outlined<FP>(a, b); else a.add(b);   }
template<class FP> SPPARK_DEVFN void bucket_dbl(xyzz_dev<FP>& a)
{   if constexpr (FP::N > 16) xyzz_dbl_outlined<FP>(a); else a.dbl();   }
// the low-latency forms (products in interleaved pairs) where the field has them: the top of the bucket sums
template<class FP> SPPARK_DEVFN void bucket_add_fast(xyzz_dev<FP>& a, const xyzz_dev<FP>& b)
{   if constexpr (field_is_montx<FP>::value) a.add_pairs(b); else bucket_add<FP>(a, b);   }
template<class FP> SPPARK_DEVFN void bucket_dbl_fast(xyzz_dev<FP>& a)
{   if constexpr (field_is_montx<FP>::value) a.dbl_pairs(); else bucket_dbl<FP>(a);   }

// ---------------------------------------------------------------------------
// |off| (nullable): the bucket offsets of the sort, off[w * (NB + 1) + b].  A bucket is written by exactly
// one owner iff it holds entries (off[b + 1] > off[b]); with the offsets at hand the empty ones are taken
// as infinity without being read, so the bucket array needs no memset before the accumulation (0.9 ms of a
// 2^26-point MSM: 5.6 GB of zeros).
template<class FP>
SPPARK_DEVFN xyzz_dev<FP> bucket_load(const xyzz_mem<FP::N>* row, const u32* o, unsigned b)
{
    if (o != nullptr && o[b + 1] == o[b]) { xyzz_dev<FP> z; z.set_inf(); return z; }
    return xyzz_dev<FP>::load(&row[b]);
}
// LAT: the low-latency additions (products in pairs): for grids of at most one resident round of waves, where the
// chain of 2(K-1) dependent additions, not the number of additions, is the time (k_bucket_level1_lat / _levelN_lat)
template<class FP, bool LAT = false>
SPPARK_DEVFN void bucket_level1_item(xyzz_mem<FP::N>* A, xyzz_mem<FP::N>* Wt, const xyzz_mem<FP::N>* buckets,
                                     unsigned NB, unsigned K, unsigned nwins, size_t id, const u32* off = nullptr)
{
    const unsigned nchunks = NB / K;
    if (id >= (size_t)nwins * nchunks) return;
    const unsigned w = id / nchunks, u = id % nchunks;
    const xyzz_mem<FP::N>* row = buckets + (size_t)w * NB + (size_t)u * K;
    const u32* o = off ? off + (size_t)w * (NB + 1) + (size_t)u * K : nullptr;
    xyzz_dev<FP> acc = bucket_load<FP>(row, o, K - 1), ret = acc;
    for (unsigned j = K - 1; j--;) {
        if (LAT) { bucket_add_fast<FP>(acc, bucket_load<FP>(row, o, j)); bucket_add_fast<FP>(ret, acc); }
        else     { bucket_add<FP>(acc, bucket_load<FP>(row, o, j)); bucket_add<FP>(ret, acc); }
    }
    acc.store(&A[id]); ret.store(&Wt[id]);
}

// (two waves per SIMD: the 28-bit-limb instantiation needs 263 registers unconstrained, seven more
// than two waves leave each -- one wave per SIMD ran the bucket sums at half speed)
template<class FP>
__global__ __launch_bounds__(256, 2)
void k_bucket_level1(xyzz_mem<FP::N>* __restrict__ A, xyzz_mem<FP::N>* __restrict__ Wt,
                     const xyzz_mem<FP::N>* __restrict__ buckets, unsigned NB, unsigned K, unsigned nwins,
                     const u32* __restrict__ off)
{   bucket_level1_item<FP>(A, Wt, buckets, NB, K, nwins, (size_t)blockIdx.x * blockDim.x + threadIdx.x, off);   }
// the same for small grids: one wave per SIMD (no register cap, no spills), products in pairs
template<class FP>
__global__ __launch_bounds__(256)
void k_bucket_level1_lat(xyzz_mem<FP::N>* __restrict__ A, xyzz_mem<FP::N>* __restrict__ Wt,
                         const xyzz_mem<FP::N>* __restrict__ buckets, unsigned NB, unsigned K, unsigned nwins,
                         const u32* __restrict__ off)
{   bucket_level1_item<FP, true>(A, Wt, buckets, NB, K, nwins, (size_t)blockIdx.x * blockDim.x + threadIdx.x, off);   }

// level >= 2: chunk u of K items (A_j, Wt_j), each item spanning 2^lgG buckets:
//   A'[u] = sum_j A_j      Wt'[u] = sum_j Wt_j + 2^lgG * sum_j j*A_j
template<class FP, bool LAT = false>
SPPARK_DEVFN void bucket_levelN_item(xyzz_mem<FP::N>* A2, xyzz_mem<FP::N>* Wt2,
                                     const xyzz_mem<FP::N>* A1, const xyzz_mem<FP::N>* Wt1,
                                     unsigned nitems, unsigned K, unsigned lgG, unsigned nwins, size_t id)
{
    const unsigned nchunks = nitems / K;
    if (id >= (size_t)nwins * nchunks) return;
    const unsigned w = id / nchunks, u = id % nchunks;
    const size_t base = (size_t)w * nitems + (size_t)u * K;
    xyzz_dev<FP> acc, r, sw;
    acc.set_inf(); r.set_inf();
    sw = xyzz_dev<FP>::load(&Wt1[base]);
    auto add = [](xyzz_dev<FP>& a, const xyzz_dev<FP>& b) { if (LAT) bucket_add_fast<FP>(a, b); else bucket_add<FP>(a, b); };
    for (unsigned j = K - 1; j >= 1; j--) {
        add(acc, xyzz_dev<FP>::load(&A1[base + j]));
        add(r, acc);
        add(sw, xyzz_dev<FP>::load(&Wt1[base + j]));
    }
    add(acc, xyzz_dev<FP>::load(&A1[base]));
    for (unsigned k = 0; k < lgG; k++) { if (LAT) bucket_dbl_fast<FP>(r); else bucket_dbl<FP>(r); }
    add(sw, r);
    acc.store(&A2[id]); sw.store(&Wt2[id]);
}

template<class FP>
__global__ __launch_bounds__(256, 2)
void k_bucket_levelN(xyzz_mem<FP::N>* __restrict__ A2, xyzz_mem<FP::N>* __restrict__ Wt2,
                     const xyzz_mem<FP::N>* __restrict__ A1, const xyzz_mem<FP::N>* __restrict__ Wt1,
                     unsigned nitems, unsigned K, unsigned lgG, unsigned nwins)
{   bucket_levelN_item<FP>(A2, Wt2, A1, Wt1, nitems, K, lgG, nwins, (size_t)blockIdx.x * blockDim.x + threadIdx.x);   }
template<class FP>
__global__ __launch_bounds__(256)
void k_bucket_levelN_lat(xyzz_mem<FP::N>* __restrict__ A2, xyzz_mem<FP::N>* __restrict__ Wt2,
                         const xyzz_mem<FP::N>* __restrict__ A1, const xyzz_mem<FP::N>* __restrict__ Wt1,
                         unsigned nitems, unsigned K, unsigned lgG, unsigned nwins)
{   bucket_levelN_item<FP, true>(A2, Wt2, A1, Wt1, nitems, K, lgG, nwins, (size_t)blockIdx.x * blockDim.x + threadIdx.x);   }

// ---------------------------------------------------------------------------
// Top of the bucket sums.  Once a window is down to M = 2^m <= BUCKET_TOP_MAX items (A_j, Wt_j) of
// 2^lgG buckets each, the chunked levels above are chains of dependent additions run by a handful of
// waves (0.6 ms per level whatever its size: one wave gets an issue slot every ~12 cycles).  The
// weighted sum is re-associated so that its DEPTH, not its work, is small:
//     sum_j Wt_j + 2^lgG * sum_j j*A_j  =  sum_j Wt_j + sum_{b<m} 2^(lgG+b) * S_b,   S_b = sum_{j: bit b of j} A_j
// k_bucket_top_bits: work-group (b, w) reduces S_b (b < m) or sum_j Wt_j (b == m) -- M/512 additions per
// lane, an 8-step tree through LDS, then b + lgG doublings by one lane; k_bucket_top_sum adds the m + 1
// parts of a window with a 4..5-step tree.  m*M/2 + M additions instead of ~3M, at a depth of
// ~M/512 + 8 + (m + lgG) doublings + 5 instead of (m/3) * (23 additions + doublings).
// ---------------------------------------------------------------------------
static constexpr unsigned BUCKET_TOP_MAX = 4096, BUCKET_TOP_NT = 256;

template<class FP>
SPPARK_DEVFN void lds_tree_sum(xyzz_dev<FP>& acc, xyzz_mem<FP::N>* img, unsigned tid, unsigned nt)
{
    for (unsigned s = nt >> 1; s >= 1; s >>= 1) {
        if (tid >= s && tid < 2 * s) acc.store(&img[tid]);
        __syncthreads();
        if (tid < s) bucket_add_fast<FP>(acc, xyzz_dev<FP>::load(&img[tid + s]));
        __syncthreads();
    }
}

// The per-lane parts of the two kernels as functions of (b, w, tid), so that tests/emu runs the same index math on
// the host (the LDS tree between them is a plain pairwise reduction).
// items of a lane: j = lo | i << p | hi << (p + lgI), bit b inside the i field, so that every lane
// owns as many selected items as any other
template<class FP>
SPPARK_DEVFN xyzz_dev<FP> bucket_top_gather(const xyzz_mem<FP::N>* A, const xyzz_mem<FP::N>* Wt,
                                            unsigned nitems, unsigned m, unsigned b, unsigned w, unsigned tid, unsigned nt)
{
    const xyzz_mem<FP::N>* src = (b == m ? Wt : A) + (size_t)w * nitems;
    unsigned lgI = 0;
    while ((nt << lgI) < nitems) lgI++;
    const unsigned p = b >= m ? 0 : (b < m - lgI ? b : m - lgI);
    const unsigned lo = tid & ((1u << p) - 1), hi = tid >> p;
    xyzz_dev<FP> acc; acc.set_inf();
    #pragma unroll 1
    for (unsigned i = 0; i < (1u << lgI); i++) {
        const unsigned j = lo | (i << p) | (hi << (p + lgI));
        if (j < nitems && (b == m || ((j >> b) & 1))) bucket_add_fast<FP>(acc, xyzz_dev<FP>::load(&src[j]));
    }
    return acc;
}
// lane 0 after the tree: the weight 2^(b + lgG) of bit b (none for the plain sum of the Wt's, b == m)
template<class FP>
SPPARK_DEVFN void bucket_top_finish(xyzz_dev<FP>& acc, xyzz_mem<FP::N>* parts, unsigned m, unsigned lgG, unsigned b, unsigned w)
{
    if (b < m) {
        #pragma unroll 1
        for (unsigned k = 0; k < b + lgG; k++) bucket_dbl_fast<FP>(acc);
    }
    acc.store(&parts[(size_t)w * (m + 1) + b]);
}
// The sums CUT INTO PIECES (round 6, the cooperative kernels): a work-group per piece instead of per sum.  The work-groups of a
// top are few (13 x 16 at 2^20 points) and each one's time is its lanes' chain of additions -- 8 per lane for a subset sum of
// 4096 items, 16 for the plain sum, which alone sets the kernel's time.  Piece s of |nsub| of a sum is what the lanes
// s nt .. (s + 1) nt - 1 of a work-group of nsub x nt lanes would gather (bucket_top_gather with that lane number: every
// piece owns as many selected items as any other); doubling is linear, so every piece takes its sum's b + lgG doublings
// itself and the pieces of a window are its m sb + sp parts.  q: part number within the window.
struct top_piece { unsigned b, sub, nsub; };
SPPARK_DEVFN top_piece bucket_top_piece(unsigned q, unsigned m, unsigned sb, unsigned sp)
{
    top_piece t;
    if (q < m * sb) { t.b = q / sb; t.sub = q % sb; t.nsub = sb; }
    else            { t.b = m; t.sub = q - m * sb; t.nsub = sp; }
    return t;
}
// pieces per subset sum (sb) and per plain sum (sp) for |nitems| items and work-groups of |nt| lanes.  Measured
// (profiles/r06_msm_top_cut_sweep.log, 2^18 .. 2^24 points): cutting the PLAIN sum of a 4096-item top in two -- the one
// work-group with twice the additions per lane of all the others -- is the whole gain (tail 2^19 0.79 -> 0.72 ms, 2^20
// 0.99 -> 0.94, 2^21 1.24 -> 1.16, 2^22 1.35 -> 1.30); more pieces (2 / 4, 2 / 8: twice the work-groups, two per CU) bring
// nothing further -- what is left is the tree and the doublings -- and at 2048 items (2^18 points) nothing changes.
// At most 32 parts per window (k_bucket_top_sum_coop's image).
static inline void bucket_top_cut(unsigned nitems, unsigned nt, unsigned& sb, unsigned& sp)
{
    sb = 1;
    sp = nitems >= 16 * nt ? 2 : 1;
}

template<class FP>
SPPARK_DEVFN xyzz_dev<FP> bucket_top_sum_gather(const xyzz_mem<FP::N>* parts, unsigned m, unsigned w, unsigned tid)
{
    xyzz_dev<FP> acc; acc.set_inf();
    if (tid <= m) acc = xyzz_dev<FP>::load(&parts[(size_t)w * (m + 1) + tid]);
    return acc;
}

// the small windows' form of the same sum, straight from the buckets (k_bucket_small_bits_coop, msm_coop_kernels.hpp)
static constexpr unsigned SMALL_SUMS_MAX_NB = 256;
// the t-th (0-based) bucket NUMBER in 1 .. NB with bit b set; b == m: NB itself
SPPARK_DEVFN unsigned small_sums_member(unsigned b, unsigned m, unsigned t)
{   return b >= m ? (1u << m) : ((((t >> b) << 1) | 1u) << b) | (t & ((1u << b) - 1u));   }

template<class FP>
SPPARK_DEVFN xyzz_dev<FP> small_sums_gather(const xyzz_mem<FP::N>* buckets, const u32* off, unsigned NB, unsigned m,
                                            unsigned b, unsigned w, unsigned t)
{
    xyzz_dev<FP> r; r.set_inf();
    if (t >= (b >= m ? 1u : NB / 2)) return r;
    const unsigned j = small_sums_member(b, m, t) - 1;              // 0-based bucket
    return bucket_load<FP>(buckets + (size_t)w * NB, off + (size_t)w * (NB + 1), j);
}

template<class FP>
__global__ __launch_bounds__(BUCKET_TOP_NT, 2)
void k_bucket_top_bits(xyzz_mem<FP::N>* __restrict__ parts, const xyzz_mem<FP::N>* __restrict__ A,
                       const xyzz_mem<FP::N>* __restrict__ Wt, unsigned nitems, unsigned m, unsigned lgG)
{
    extern __shared__ unsigned char top_lds[];
    xyzz_mem<FP::N>* img = reinterpret_cast<xyzz_mem<FP::N>*>(top_lds);
    const unsigned b = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
    xyzz_dev<FP> acc = bucket_top_gather<FP>(A, Wt, nitems, m, b, w, tid, BUCKET_TOP_NT);
    lds_tree_sum<FP>(acc, img, tid, BUCKET_TOP_NT);
    if (tid == 0) bucket_top_finish<FP>(acc, parts, m, lgG, b, w);
}

template<class FP>
__global__ __launch_bounds__(32)
void k_bucket_top_sum(xyzz_mem<FP::N>* __restrict__ out, const xyzz_mem<FP::N>* __restrict__ parts, unsigned m)
{
    extern __shared__ unsigned char top_lds[];
    xyzz_mem<FP::N>* img = reinterpret_cast<xyzz_mem<FP::N>*>(top_lds);
    const unsigned w = blockIdx.x, tid = threadIdx.x;
    xyzz_dev<FP> acc = bucket_top_sum_gather<FP>(parts, m, w, tid);
    lds_tree_sum<FP>(acc, img, tid, 32);
    if (tid == 0) acc.store(&out[w]);
}

// ---------------------------------------------------------------------------
// k_convert_points with COALESCED accesses (round 6; plain points of the G1 bucket fields): the wire points of a work-group --
// 256 x 96 contiguous bytes -- are read in 16-byte pieces by consecutive lanes (every 128-byte line once, by one instruction)
// into an LDS image whose point stride is padded to an odd multiple of 16 bytes (112 for 96: the eight lanes of a b128 LDS
// access then hit disjoint banks), each lane converts ITS point from the image, and the 128-byte records go back the same way
// (stride 144 in LDS, whole lines to memory).  The per-lane form (a lane reads 6 x 16 bytes 96 bytes apart from its neighbour)
// asks the L1 for every line six times, with 32 waves' footprints thrashing its 32 KB.
// ---------------------------------------------------------------------------
template<class FP>
SPPARK_DEVFN constexpr unsigned convert_lds_stride(unsigned raw) { return (raw / 4) % 8 == 0 ? raw + 16 : raw; }
// FLAGGED: the Affine_inf_t layout (X | Y | flag byte + padding: 8 bytes more per point, what arkworks / the Rust wrapper pass).
// Its points are 6.5 pieces long, so the LDS image is the memory image itself and a lane reads its point in 8-byte words
// (104 bytes = 26 banks apart: sixteen lanes of a b64 access on disjoint bank pairs).
template<class FP, bool FLAGGED>
__global__ __launch_bounds__(256)
void k_convert_points_staged(unsigned char* __restrict__ dst, const unsigned char* __restrict__ src, unsigned n)
{
    typedef affine_loader<FP> AL;
    constexpr unsigned IN = 2 * FP::NW * 4 + (FLAGGED ? 8 : 0), OUT = AL::STRIDE;     // bytes per wire point / per record
    constexpr unsigned IN_L = FLAGGED ? IN : convert_lds_stride<FP>(IN), OUT_L = convert_lds_stride<FP>(OUT);
    static_assert(IN % 8 == 0 && (256 * IN) % 16 == 0 && OUT % 16 == 0, "8 / 16-byte pieces");
    constexpr unsigned LDS_BYTES = 256 * (IN_L > OUT_L ? IN_L : OUT_L) + 16;
    __shared__ uint4 img[LDS_BYTES / 16];
    unsigned char* lds = reinterpret_cast<unsigned char*>(img);
    const unsigned tid = threadIdx.x;
    const size_t p0 = (size_t)blockIdx.x * 256;
    const unsigned cnt = n - p0 < 256 ? (unsigned)(n - p0) : 256u;              // points of this work-group
    // ---- in: consecutive lanes, consecutive 16-byte pieces
    const unsigned char* gin = src + p0 * IN;
    const unsigned bytes = cnt * IN;
    #pragma unroll
    for (unsigned it = 0; it < (256 * IN / 16 + 255) / 256; it++) {
        const unsigned g = it * 256 + tid, at = g * 16;
        if (at + 16 <= bytes) {
            const uint4 v = *reinterpret_cast<const uint4*>(gin + at);
            if (FLAGGED) *reinterpret_cast<uint4*>(lds + at) = v;
            else         *reinterpret_cast<uint4*>(lds + (g / (IN / 16)) * IN_L + (g % (IN / 16)) * 16) = v;
        } else if (FLAGGED && at + 8 <= bytes) {                                // (an odd number of points: the last 8 bytes)
            *reinterpret_cast<uint2*>(lds + at) = *reinterpret_cast<const uint2*>(gin + at);
        }
    }
    __syncthreads();
    u32 w[OUT / 4] = {};
    if (tid < cnt) {
        u32 in[IN / 4];
        if (FLAGGED) {
            #pragma unroll
            for (unsigned j = 0; j < IN / 8; j++) {
                const uint2 v = *reinterpret_cast<const uint2*>(lds + tid * IN_L + j * 8);
                in[2*j] = v.x; in[2*j+1] = v.y;
            }
        } else {
            #pragma unroll
            for (unsigned j = 0; j < IN / 16; j++) {
                const uint4 v = *reinterpret_cast<const uint4*>(lds + tid * IN_L + j * 16);
                in[4*j] = v.x; in[4*j+1] = v.y; in[4*j+2] = v.z; in[4*j+3] = v.w;
            }
        }
        bool inf;
        if (FLAGGED) inf = (in[2 * FP::NW] & 1u) != 0;                          // the flag byte
        else {
            u32 any = 0;
            #pragma unroll
            for (unsigned i = 0; i < 2 * FP::NW; i++) any |= in[i];
            inf = any == 0;                                                     // Affine_t: infinity = all-zero coordinates
        }
        const FP x = FP::from_std(in), y = FP::from_std(in + FP::NW);
        x.to_wire(w); y.to_wire(w + FP::NL);
        if (inf) w[FP::NL - 1] |= 0x80000000u;
    }
    __syncthreads();                                                            // (every lane has read its point: the image is reused)
    if (tid < cnt) {
        #pragma unroll
        for (unsigned j = 0; j < OUT / 16; j++)
            *reinterpret_cast<uint4*>(lds + tid * OUT_L + j * 16) = make_uint4(w[4*j], w[4*j+1], w[4*j+2], w[4*j+3]);
    }
    __syncthreads();
    // ---- out: whole lines
    uint4* gout = reinterpret_cast<uint4*>(dst + p0 * OUT);
    #pragma unroll
    for (unsigned it = 0; it < OUT / 16; it++) {
        const unsigned g = it * 256 + tid;
        if (g < cnt * (OUT / 16)) gout[g] = *reinterpret_cast<const uint4*>(lds + (g / (OUT / 16)) * OUT_L + (g % (OUT / 16)) * 16);
    }
}

// ---------------------------------------------------------------------------
// Coordinate fields with an internal representation (ff/montx_dev.hpp): the points are
// converted ONCE per MSM (or once per preload) into the field's own records, and the W
// window sums are converted back to the reference's wire image at the end.
// ---------------------------------------------------------------------------

template<class FP, bool FLAGGED>
__global__ __launch_bounds__(256)
void k_convert_points(unsigned char* __restrict__ dst, const unsigned char* __restrict__ src, unsigned n, unsigned stride)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) affine_loader<FP>::template convert<FLAGGED>(dst, src, i, stride);
}

// ---------------------------------------------------------------------------
// Fixed-base tables for preloaded points (msm_t::preload(..., fixed_base)): for every point P_i the affine
// multiples 2^(off_j) * P_i, off_j = the first bit of window j, j < nwins, as records of the field's own format:
// table[j * n + i].  With them all windows of an MSM share ONE bucket set (entry (j, i) of the digit array is the
// point j * n + i of a one-window MSM over nwins * n points): no Horner, the bucket sums once instead of per window,
// and a window as wide as 26 bits (10 instead of 12 additions per point at 2^26 points).  The reference's analogue is
// the msm_t that keeps its bases on the device (msm/pippenger.cuh:351-385); the tables themselves are new.
// One work item per point: off_j doublings in XYZZ, then x = X / ZZ, y = Y / ZZZ with one inversion per eight entries
// (Fermat: ~570 products; a one-time cost of set_points: 2^26 points x 11 windows in ~4 s).  Infinity stays infinity.
// ---------------------------------------------------------------------------
template<class FP>
SPPARK_DEVFN void fixed_base_table_item(unsigned char* table, unsigned n, unsigned nwins, unsigned nbits, size_t i)
{
    typedef affine_loader<FP> AL;
    constexpr unsigned GROUP = 8;                   // entries normalised with ONE inversion (Montgomery's trick; local arrays: scratch)
    if (i >= n) return;
    affine_dev<FP> p = load_affine<FP, false>(table, i, 0);           // level 0 = the converted points themselves
    xyzz_dev<FP> acc; acc.set(p, false);
    #pragma unroll 1
    for (unsigned j0 = 1; j0 < nwins; j0 += GROUP) {
        const unsigned cnt = nwins - j0 < GROUP ? nwins - j0 : GROUP;
        xyzz_dev<FP> sv[GROUP]; FP pre[GROUP]; bool inf[GROUP];
        FP run = FP::one();                         // product of ZZ * ZZZ over the finite entries so far
        #pragma unroll 1
        for (unsigned g = 0; g < cnt; g++) {
            #pragma unroll 1
            for (unsigned k = 0; k < window_len(j0 + g - 1, nwins, nbits); k++) acc.dbl();
            const FP zz = acc.ZZ * acc.ZZZ;         // ZZ, ZZZ normalised
            // (ZZ == 0 mod p without all-zero limbs: the double of a point of order two, which BLS12-377's curve has)
            inf[g] = acc.is_inf() || zz.template is_zero_mod<2>();
            if (inf[g]) acc.set_inf();
            sv[g] = acc; pre[g] = run;
            if (!inf[g]) run = run * zz;
        }
        FP inv = run.inverse();                     // 1 / (all of them); peeled from the back
        #pragma unroll 1
        for (unsigned g = cnt; g--;) {
            u32 w[AL::STRIDE / 4] = {};
            if (inf[g]) {
                w[FP::NL - 1] = 0x80000000u;        // the flag of an infinite record (bit 31 of X's top limb)
            } else {
                const FP zz = sv[g].ZZ * sv[g].ZZZ;
                const FP iz = inv * pre[g];         // 1 / (ZZ * ZZZ) of entry g
                inv = inv * zz;
                const FP x = sv[g].X * (iz * sv[g].ZZZ), y = sv[g].Y * (iz * sv[g].ZZ);    // fat left operands; < 2p, normalised
                x.to_wire(w); y.to_wire(w + FP::NL);
            }
            uint4* q = reinterpret_cast<uint4*>(table + ((size_t)(j0 + g) * n + i) * (size_t)AL::STRIDE);
            #pragma unroll
            for (unsigned k = 0; k < AL::STRIDE / 16; k++) q[k] = make_uint4(w[4*k], w[4*k+1], w[4*k+2], w[4*k+3]);
        }
    }
}
template<class FP>
__global__ __launch_bounds__(256)
void k_fixed_base_table(unsigned char* __restrict__ table, unsigned n, unsigned nwins, unsigned nbits)
{   fixed_base_table_item<FP>(table, n, nwins, nbits, (size_t)blockIdx.x * blockDim.x + threadIdx.x);   }

template<class FP, int STD_WORDS>
__global__ __launch_bounds__(64)
void k_finalize(xyzz_mem<STD_WORDS>* __restrict__ out, const xyzz_mem<FP::N>* __restrict__ in, unsigned count)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) xyzz_dev<FP>::load(&in[i]).store_std(&out[i]);
}

} // namespace sppark_amd
