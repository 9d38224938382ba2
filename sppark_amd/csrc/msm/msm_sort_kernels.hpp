// Counting-sort kernels of the MSM pipeline (non-template; included by exactly
// one translation unit, api/msm_api.hip).  See msm_kernels.hpp for the overview.
//
// Bucket index k = |digit| - 1 in [0, 2^(c-1)) is split k = (k_hi : k_lo), HB + LB
// = c - 1 bits, and grouped in two LDS-resident levels:
//
//   level A  (window, point slab) block: histogram of k_hi in LDS (2^HB counters),
//            slab/bucket exclusive scans, then scatter of 8-byte records
//            (point index | sign<<31, k_lo) into 2^HB partitions per window.  Each
//            block feeds 2^HB output streams (8 KB each at n = 2^26), not 2^(c-1).
//   level B  (window, k_hi) block: histogram of k_lo in LDS (2^LB counters), block
//            scan, bucket offsets out, then scatter of the 4-byte entries inside
//            the partition's own contiguous output range (256 KB at n = 2^26).
//
// All counting and cursor arithmetic happens in LDS (ds_add / ds_add_rtn); no
// global atomics anywhere.  The reference instead sorts with two cooperative
// grid-synchronised passes sized for <= 32 blocks (msm/sort.cuh:120-357).
#pragma once
#include "../ff/mont_dev.hpp"
#include "msm_sort_records.hpp"

namespace sppark_amd {

// Work-group size of the three bulk kernels (k_histA, k_scatterA, k_sortB): 1024 lanes.  (512-lane,
// 16-VGPR work-groups, which fit beside the two 232-VGPR waves per SIMD of k_accumulate, were built
// for the window-group overlap experiment; the overlap gained nothing and the smaller groups cost
// 3.6 ms per 2^26 MSM: k_sortB 6.7 -> 9.3 ms, k_scatterA 8.6 -> 9.6 ms, profiles/r02_msm_groups.log.)
static constexpr unsigned SORT_NT = 1024;
#define SORT_VGPRS 32
static constexpr int SORTB_UNROLL = 4;       // loads in flight per lane in k_sortB (partitions beyond the register path)
// k_sortB keeps a partition of up to SORTB_PER * SORT_NT entries in REGISTERS between its counting
// and its placement step (one read of the partition instead of two) and places the entries in an LDS
// image of the partition's output range, which is then written out in whole lines.  Measured
// before (rocprofv3 WRITE_SIZE, 2^26 points): 14.7 GB written for 3.2 GB of sorted entries -- the
// 4-byte scatter into a 64 KB range reached HBM as partial lines -- and 13 GB read for 6.4 GB.
// 18 x 1024 covers the uniform case at 2^26 (16384 +- 128 entries per partition).
static constexpr int SORTB_PER = 18;
static constexpr unsigned SORTB_STAGE = SORTB_PER * SORT_NT;

// The windows of an MSM are split evenly (msm_kernels.hpp window_len): the first nbits % nwins of them
// are one bit longer than the others.  |LB| is the k_lo width of the long windows; a short window
// (local index >= short_from) keeps all 2^HB partitions busy with a k_lo of LB - 1 bits -- with the
// long windows' split its digits would fill only the lower half of the partitions, at twice the size.
// The bucket-offset rows keep the common stride (NA << LB) + 1.
__host__ __device__ inline unsigned window_lb(unsigned LB, unsigned w, unsigned short_from)
{   return (w >= short_from && LB) ? LB - 1 : LB;   }

// Exclusive prefix of |s| over the SORT_NT lanes of a work-group: wave scan with cross-lane moves,
// 16 wave totals through LDS (two barriers; the ten-step ladder through LDS it replaces cost twenty).
// *total = sum over all lanes.  |wsum|: 16 LDS words.
SPPARK_DEVFN u32 block_scan_excl(u32 s, u32* wsum, u32* total)
{
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 incl = s;
    #pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) { u32 v = __shfl_up(incl, dlt); if (lane >= (unsigned)dlt) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 wv = lane < 16 ? wsum[lane] : 0, before = lane < wave ? wv : 0, tot = wv;
    #pragma unroll
    for (int m = 1; m < 16; m <<= 1) { before += __shfl_xor(before, m); tot += __shfl_xor(tot, m); }
    before = __shfl(before, 0); *total = __shfl(tot, 0);
    __syncthreads();                        // wsum may be rewritten by the next call
    return before + incl - s;
}

// H[(w*nslabs + slab)*NA + k_hi] = count of the slab's window-w digits in partition k_hi
__global__ __launch_bounds__(SORT_NT) __attribute__((amdgpu_num_vgpr(SORT_VGPRS)))
void k_histA(u32* __restrict__ H, const u32* __restrict__ digits, unsigned n,
             unsigned nslabs, unsigned slab_sz, unsigned NA, unsigned LBL, unsigned short_from)
{
    extern __shared__ u32 lds_cnt[];
    const unsigned slab = blockIdx.x, w = blockIdx.y, LB = window_lb(LBL, w, short_from);
    for (unsigned b = threadIdx.x; b < NA; b += blockDim.x) lds_cnt[b] = 0;
    __syncthreads();

    const unsigned lo = slab * slab_sz, hi = min(n, lo + slab_sz);
    const u32* dig = digits + (size_t)w * n;
    // four independent loads in flight per lane before the LDS atomics
    for (unsigned i = lo + threadIdx.x; i < hi; i += 4 * blockDim.x) {
        u32 d[4];
        #pragma unroll
        for (int u = 0; u < 4; u++) { unsigned j = i + u * blockDim.x; d[u] = j < hi ? dig[j] : 0; }
        #pragma unroll
        for (int u = 0; u < 4; u++) if (d[u]) atomicAdd(&lds_cnt[((d[u] & 0x7fffffffu) - 1) >> LB], 1u);
    }
    __syncthreads();

    u32* out = H + ((size_t)w * nslabs + slab) * NA;
    for (unsigned b = threadIdx.x; b < NA; b += blockDim.x) out[b] = lds_cnt[b];
}

// slab-exclusive prefix per (window, partition); tot[w*NA+b] = partition total
__global__ __launch_bounds__(256)
void k_scan_slabs(u32* __restrict__ H, u32* __restrict__ tot, unsigned nslabs, unsigned NA, unsigned nwins)
{
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)nwins * NA) return;
    const unsigned w = id / NA, b = id % NA;
    u32 run = 0;
    for (unsigned s = 0; s < nslabs; s++) {
        u32* p = H + ((size_t)w * nslabs + s) * NA + b;
        u32 t = *p; *p = run; run += t;
    }
    tot[id] = run;
}

// partition-exclusive prefix per window: offA[w*(NA+1) + b], offA[..+NA] = #entries
__global__ __launch_bounds__(1024)
void k_scan_parts(u32* __restrict__ offA, const u32* __restrict__ tot, unsigned NA)
{
    __shared__ u32 part[1024];
    const unsigned w = blockIdx.x, tid = threadIdx.x;
    const unsigned per = (NA + 1023) / 1024;
    const unsigned lo = min(NA, tid * per), hi = min(NA, lo + per);
    const u32* t = tot + (size_t)w * NA;
    u32 sum = 0;
    for (unsigned b = lo; b < hi; b++) sum += t[b];
    part[tid] = sum;
    __syncthreads();
    for (unsigned d = 1; d < 1024; d <<= 1) {
        u32 v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    u32 run = part[tid] - sum;
    u32* o = offA + (size_t)w * (NA + 1);
    for (unsigned b = lo; b < hi; b++) { o[b] = run; run += t[b]; }
    if (tid == 1023) o[NA] = part[1023];
}

// level-A scatter: partA[w*n + pos] = the record of (point index, sign, k_lo)
template<bool PK>
__global__ __launch_bounds__(SORT_NT) __attribute__((amdgpu_num_vgpr(SORT_VGPRS)))
void k_scatterA(typename recA<PK>::type* __restrict__ partA, const u32* __restrict__ digits,
                const u32* __restrict__ H, const u32* __restrict__ offA,
                unsigned n, unsigned nslabs, unsigned slab_sz, unsigned NA, unsigned LBL, unsigned short_from, unsigned IB)
{
    extern __shared__ u32 lds_cur[];
    const unsigned slab = blockIdx.x, w = blockIdx.y, LB = window_lb(LBL, w, short_from);
    const u32* h = H + ((size_t)w * nslabs + slab) * NA;
    const u32* o = offA + (size_t)w * (NA + 1);
    for (unsigned b = threadIdx.x; b < NA; b += blockDim.x) lds_cur[b] = h[b] + o[b];
    __syncthreads();

    const unsigned lo = slab * slab_sz, hi = min(n, lo + slab_sz);
    const u32* dig = digits + (size_t)w * n;
    typename recA<PK>::type* dst = partA + (size_t)w * n;
    const u32 lomask = (1u << LB) - 1;
    for (unsigned i = lo + threadIdx.x; i < hi; i += 4 * blockDim.x) {
        u32 d[4];
        #pragma unroll
        for (int u = 0; u < 4; u++) { unsigned j = i + u * blockDim.x; d[u] = j < hi ? dig[j] : 0; }
        #pragma unroll
        for (int u = 0; u < 4; u++) {
            if (d[u]) {
                u32 k = (d[u] & 0x7fffffffu) - 1;
                u32 pos = atomicAdd(&lds_cur[k >> LB], 1u);
                dst[pos] = recA<PK>::make((i + u * blockDim.x) | (d[u] & 0x80000000u), k & lomask, LBL, IB);
            }
        }
    }
}

// level-A scatter through LDS (NA <= SCATA_MAX_NA partitions).  k_scatterA hands every entry to the
// memory system as an isolated 8-byte write (rocprofv3 WRITE_SIZE at 2^26 points: 23.6 GB for 6.4 GB
// of entries -- one 32-byte sector per entry).  Here a work-group takes its slab in tiles of
// SCATA_PER * SORT_NT entries: tile histogram (LDS atomics), block scan, placement of the entries in
// an LDS image of the tile GROUPED BY PARTITION, then a linear read-out in which adjacent lanes hold
// adjacent entries of the same partition (3-4 per partition and tile at 2^12 partitions) and their
// writes leave the wave as one request.  The running global cursor of a partition lives in a
// register of the lane that owns its counter; per tile it publishes G[p] = cursor - tile offset, so
// the read-out address is G[p] + position in the tile.  The digits of the next tile are loaded
// before the current one is processed.
static constexpr int SCATA_PER = 14;
static constexpr unsigned SCATA_TILE = SCATA_PER * SORT_NT;
static constexpr unsigned SCATA_MAX_NA = 4 * SORT_NT;
static inline size_t scatterA_staged_lds(unsigned NA) { return (size_t)NA * 8 + 64 + (size_t)SCATA_TILE * 8; }

// Work-group barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (its fence covers global memory), which would wait for the digit prefetch of the next tile
// and for the write-out of the previous one at every one of the five barriers of a tile.  The waves of
// this kernel communicate through LDS alone; what they write to global memory is read by later kernels.
// every outstanding vector-memory operation of the wave (s_waitcnt vmcnt(0); through the builtin, so that the compiler's own
// wait insertion knows the counter is zero afterwards)
SPPARK_DEVFN void wait_vmem()
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_waitcnt(0x0f70);         // vmcnt = 0, expcnt and lgkmcnt at their maxima (not waited for)
#endif
}
SPPARK_DEVFN void lds_barrier()
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

template<bool PK>
__global__ __launch_bounds__(SORT_NT)
void k_scatterA_staged(typename recA<PK>::type* __restrict__ partA, const u32* __restrict__ digits,
                       const u32* __restrict__ H, const u32* __restrict__ offA,
                       unsigned n, unsigned nslabs, unsigned slab_sz, unsigned NA, unsigned LBL, unsigned short_from, unsigned IB)
{
    extern __shared__ u32 lds_sa[];
    constexpr unsigned NT = SORT_NT;
    u32* cnt = lds_sa;                      // tile histogram -> placement cursors
    u32* G = cnt + NA;                      // global position of a partition's first tile entry, minus its tile offset
    u32* wsum = G + NA;                     // per-wave totals of the block scan
    uint2* stage = reinterpret_cast<uint2*>(wsum + 16);
    const unsigned slab = blockIdx.x, w = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned LB = window_lb(LBL, w, short_from);
    const u32* h = H + ((size_t)w * nslabs + slab) * NA;
    const u32* o = offA + (size_t)w * (NA + 1);
    u32 cur[4];                             // this lane owns counters 4*tid .. 4*tid+3
    #pragma unroll
    for (int i = 0; i < 4; i++) { unsigned b = 4 * tid + i; cur[i] = b < NA ? h[b] + o[b] : 0; if (b < NA) cnt[b] = 0; }

    const unsigned lo = slab * slab_sz, hi = min(n, lo + slab_sz);
    const u32* dig = digits + (size_t)w * n;
    typename recA<PK>::type* dst = partA + (size_t)w * n;
    const u32 lomask = (1u << LB) - 1;
    u32 d[SCATA_PER], dn[SCATA_PER];
    #pragma unroll
    for (int u = 0; u < SCATA_PER; u++) { unsigned j = lo + u * NT + tid; dn[u] = j < hi ? dig[j] : 0; }
    __syncthreads();

    wait_vmem();                            // (the first tile's digits: the loop below never waits at its top -- see phase D)
    #pragma unroll
    for (int u = 0; u < SCATA_PER; u++) d[u] = dn[u];
    for (unsigned t0 = lo; t0 < hi; t0 += SCATA_TILE) {
        #pragma unroll
        for (int u = 0; u < SCATA_PER; u++) { unsigned j = t0 + SCATA_TILE + u * NT + tid; dn[u] = j < hi ? dig[j] : 0; }
        // A: tile histogram
        #pragma unroll
        for (int u = 0; u < SCATA_PER; u++) if (d[u]) atomicAdd(&cnt[((d[u] & 0x7fffffffu) - 1) >> LB], 1u);
        lds_barrier();
        // B: block scan over the counters, four per lane
        u32 c[4], s = 0;
        #pragma unroll
        for (int i = 0; i < 4; i++) { unsigned b = 4 * tid + i; c[i] = b < NA ? cnt[b] : 0; s += c[i]; }
        u32 incl = s;
        #pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) { u32 v = __shfl_up(incl, dlt); if (lane >= (unsigned)dlt) incl += v; }
        if (lane == 63) wsum[wave] = incl;
        lds_barrier();
        u32 wv = lane < 16 ? wsum[lane] : 0, before = lane < wave ? wv : 0, total = wv;
        #pragma unroll
        for (int m = 1; m < 16; m <<= 1) { before += __shfl_xor(before, m); total += __shfl_xor(total, m); }
        before = __shfl(before, 0); total = __shfl(total, 0);
        u32 run = before + incl - s;
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            unsigned b = 4 * tid + i;
            if (b < NA) { cnt[b] = run; G[b] = cur[i] - run; }
            cur[i] += c[i]; run += c[i];
        }
        lds_barrier();
        // C: placement, grouped by partition
        #pragma unroll
        for (int u = 0; u < SCATA_PER; u++) {
            if (d[u]) {
                u32 k = (d[u] & 0x7fffffffu) - 1, p = k >> LB;
                u32 pos = atomicAdd(&cnt[p], 1u);
                stage[pos] = make_uint2((t0 + u * NT + tid) | (d[u] & 0x80000000u), (p << 16) | (k & lomask));
            }
        }
        lds_barrier();
        // The next tile's digits are taken over HERE, three phases after their loads were issued and before this tile's stores
        // are: gfx9 counts loads and stores in ONE counter (vmcnt) and lets them complete out of order with respect to each
        // other, so a wait for loads while stores are in flight is a wait for EVERYTHING -- at the top of the next iteration
        // (where the compiler puts it) that was this tile's whole write-out, freshly issued.  Here the only stores
        // outstanding are those of the tile before.  (3.79 -> 3.73 ms at 2^26: the kernel is bound by its LDS phases.)
        wait_vmem();
        #pragma unroll
        for (int u = 0; u < SCATA_PER; u++) d[u] = dn[u];
        // D: read-out in tile order; counters cleared for the next tile
        #pragma unroll
        for (int i = 0; i < 4; i++) { unsigned b = 4 * tid + i; if (b < NA) cnt[b] = 0; }
        #pragma unroll
        for (int u = 0; u < SCATA_PER; u++) {
            unsigned e = u * NT + tid;
            if (e < total) { uint2 v = stage[e]; dst[G[v.y >> 16] + e] = recA<PK>::make(v.x, v.y & 0xffffu, LBL, IB); }
        }
        lds_barrier();
    }
}

// level B: block (k_hi, w) groups its partition by k_lo.
//   off[w*(NB+1) + k_hi*2^LB + j] = first position of bucket (k_hi, j) in window w's list
//   sorted[w*n + pos] = point index | sign<<31
template<bool PK>
__global__ __launch_bounds__(SORT_NT) __attribute__((amdgpu_num_vgpr(64)))
void k_sortB(u32* __restrict__ sorted, u32* __restrict__ off, const typename recA<PK>::type* __restrict__ partA,
             const u32* __restrict__ offA, unsigned n, unsigned NA, unsigned LBL, unsigned short_from, unsigned big, partA_fmt fmt)
{
    extern __shared__ u32 lds[];            // 2^LB counters, SORT_NT scan words, SORTB_STAGE staged entries, the group boundaries
    typedef typename recA<PK>::type rec_t;
    constexpr unsigned NT = SORT_NT;
    const unsigned LB = window_lb(LBL, blockIdx.y, short_from);
    const size_t NB = (size_t)NA << LBL;    // buckets per window (stride of off[]); this window uses NA << LB of them
    const unsigned NL = 1u << LB;
    const u32 kmask = (1u << LBL) - 1;
    u32* cnt = lds;
    u32* part = lds + NL;
    u32* stage = part + NT;
    u32* bnd = stage + SORTB_STAGE;
    const unsigned khi = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
    const u32* oA = offA + (size_t)w * (NA + 1);
    const unsigned begin = oA[khi], end = oA[khi + 1];
    const rec_t* src = partA + (size_t)w * n;
    if (end - begin > big) {                // oversized partition: the cooperative kernels below sort it
        if (khi == NA - 1 && tid == 0) off[(size_t)w * (NB + 1) + NB] = end;
        if (LB != LBL) for (unsigned b = tid; b < NL; b += NT) off[(size_t)w * (NB + 1) + ((size_t)(NA + khi) << LB) + b] = oA[NA];
        return;
    }
    // a short window leaves the upper half of its bucket-offset row unused: "empty, at the end of the list"
    if (LB != LBL) for (unsigned b = tid; b < NL; b += NT) off[(size_t)w * (NB + 1) + ((size_t)(NA + khi) << LB) + b] = oA[NA];

    const bool in_regs = end - begin <= SORTB_STAGE;        // uniform over the work-group
    rec_t rr[SORTB_PER];
    for (unsigned b = tid; b < NL; b += NT) cnt[b] = 0;
    if (PK) partA_bounds(bnd, fmt, w, khi, NA, tid);
    __syncthreads();
    if (in_regs) {
        #pragma unroll
        for (int u = 0; u < SORTB_PER; u++) {
            unsigned j = begin + tid + u * NT;
            if (j < end) rr[u] = src[j];
        }
        #pragma unroll
        for (int u = 0; u < SORTB_PER; u++) if (begin + tid + u * NT < end) atomicAdd(&cnt[recA<PK>::key(rr[u], kmask)], 1u);
    } else {
        for (unsigned i = begin + tid; i < end; i += SORTB_UNROLL * NT) {
            rec_t r[SORTB_UNROLL];
            #pragma unroll
            for (int u = 0; u < SORTB_UNROLL; u++) { unsigned j = i + u * NT; if (j < end) r[u] = src[j]; }
            #pragma unroll
            for (int u = 0; u < SORTB_UNROLL; u++) if (i + u * NT < end) atomicAdd(&cnt[recA<PK>::key(r[u], kmask)], 1u);
        }
    }
    __syncthreads();

    // exclusive scan of cnt[0..NL) in place, offsets relative to |begin|
    const unsigned per = (NL + NT - 1) / NT;
    const unsigned lo = min(NL, tid * per), hi = min(NL, lo + per);
    u32 sum = 0;
    #pragma unroll 1                        // (a handful of counters per lane: unrolling only costs registers)
    for (unsigned b = lo; b < hi; b++) sum += cnt[b];
    u32 all;
    u32 run = begin + block_scan_excl(sum, part, &all);
    u32* o = off + (size_t)w * (NB + 1) + ((size_t)khi << LB);
    #pragma unroll 1
    for (unsigned b = lo; b < hi; b++) { u32 c = cnt[b]; cnt[b] = run; o[b] = run; run += c; }
    if (khi == NA - 1 && tid == 0) off[(size_t)w * (NB + 1) + NB] = end;
    __syncthreads();

    u32* dst = sorted + (size_t)w * n;
    if (in_regs) {
        #pragma unroll
        for (int u = 0; u < SORTB_PER; u++)
            if (begin + tid + u * NT < end)
                stage[atomicAdd(&cnt[recA<PK>::key(rr[u], kmask)], 1u) - begin] = recA<PK>::entry(rr[u], tid + u * NT, LBL, fmt, bnd);
        __syncthreads();
        for (unsigned i = tid; i < end - begin; i += NT) dst[begin + i] = stage[i];
        return;
    }
    for (unsigned i = begin + tid; i < end; i += SORTB_UNROLL * NT) {
        rec_t r[SORTB_UNROLL];
        #pragma unroll
        for (int u = 0; u < SORTB_UNROLL; u++) { unsigned j = i + u * NT; if (j < end) r[u] = src[j]; }
        #pragma unroll
        for (int u = 0; u < SORTB_UNROLL; u++) {
            if (i + u * NT < end) {
                u32 pos = atomicAdd(&cnt[recA<PK>::key(r[u], kmask)], 1u);
                dst[pos] = recA<PK>::entry(r[u], i + u * NT - begin, LBL, fmt, bnd);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Oversized level-A partitions (skewed scalars: "all equal", "all ones" put a whole window into
// ONE partition).  k_sortB finishes a partition with one work-group, which is right for the
// ~16 K entries of the uniform case and takes 120 ms for 2^26.  Partitions above |big| entries
// are listed and sorted by SPLIT work-groups each, in three steps that mirror k_sortB:
// slice histograms added into off[] (as counters), one scan per partition (off[] becomes the
// bucket offsets, cur[] the running cursors), slice scatter with one global reservation per
// (slice, bucket) and LDS cursors inside it.  In the uniform case the list is empty and the three
// kernels return at once.
// ---------------------------------------------------------------------------
// |split| slices per listed partition: 64 for the skewed case (a whole window in one partition); the one-window plan
// of the fixed-base mode, whose partitions are ALL a few hundred thousand entries, asks for slices of ~16 K entries
// (one reservation per (slice, bucket) is then one global atomic per ~8 entries, not per entry).
static constexpr unsigned SORTB_SPLIT = 64;
static constexpr int BIG_UNROLL = 4;        // loads in flight per lane in the slice loops

__global__ __launch_bounds__(256)
void k_big_find(u32* __restrict__ nbig, u32* __restrict__ list, u32* __restrict__ off,
                const u32* __restrict__ offA, unsigned NA, unsigned LBL, unsigned short_from, unsigned nwins, unsigned big)
{
    const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
    bool listed = false;
    if (id < NA * nwins) {
        const unsigned w = id / NA, khi = id % NA;
        const u32* oA = offA + (size_t)w * (NA + 1);
        listed = oA[khi + 1] - oA[khi] > big;
        if (listed) list[atomicAdd(nbig, 1u)] = id;
    }
    // counters of the histogram step: the wave clears the rows of its listed partitions together (one lane walking
    // 2^LB words alone was 0.3 ms when every partition is listed -- the one-window plan of the fixed-base mode)
    unsigned long long m = __ballot(listed);
    while (m) {
        const unsigned l = __builtin_ctzll(m); m &= m - 1;
        const unsigned pid = (id & ~63u) + l, w = pid / NA, khi = pid % NA, LB = window_lb(LBL, w, short_from);
        u32* o = off + (size_t)w * (((size_t)NA << LBL) + 1) + ((size_t)khi << LB);
        for (unsigned j = threadIdx.x & 63; j < (1u << LB); j += 64) o[j] = 0;
    }
}

// slice s of listed partition b; returns false when the work item does not exist
__device__ inline bool big_slice(const u32* nbig, const u32* list, const u32* offA, unsigned NA, unsigned item, unsigned split,
                                 unsigned& w, unsigned& khi, unsigned& lo, unsigned& hi, unsigned& begin)
{
    const unsigned b = item / split, s = item % split;
    if (b >= *nbig) return false;
    w = list[b] / NA; khi = list[b] % NA;
    const u32* oA = offA + (size_t)w * (NA + 1);
    begin = oA[khi];
    const unsigned end = oA[khi + 1];
    const unsigned per = (end - begin + split - 1) / split;
    lo = min(end, begin + s * per); hi = min(end, lo + per);
    return true;
}

template<bool PK>
__global__ __launch_bounds__(1024)
void k_big_hist(u32* __restrict__ off, const typename recA<PK>::type* __restrict__ partA, const u32* __restrict__ offA,
                const u32* __restrict__ nbig, const u32* __restrict__ list, unsigned n, unsigned NA, unsigned LBL, unsigned short_from,
                unsigned split)
{
    extern __shared__ u32 lds[];
    const unsigned tid = threadIdx.x;
    const u32 kmask = (1u << LBL) - 1;
    for (unsigned item = blockIdx.x; ; item += gridDim.x) {
        unsigned w, khi, lo, hi, begin;
        if (!big_slice(nbig, list, offA, NA, item, split, w, khi, lo, hi, begin)) return;
        const unsigned LB = window_lb(LBL, w, short_from), NL = 1u << LB;
        for (unsigned j = tid; j < NL; j += 1024) lds[j] = 0;
        __syncthreads();
        const typename recA<PK>::type* src = partA + (size_t)w * n;
        for (unsigned i = lo + tid; i < hi; i += BIG_UNROLL * 1024) {
            u32 k[BIG_UNROLL];
            #pragma unroll
            for (int u = 0; u < BIG_UNROLL; u++) { unsigned j = i + u * 1024; k[u] = j < hi ? recA<PK>::key(src[j], kmask) : 0xffffffffu; }
            #pragma unroll
            for (int u = 0; u < BIG_UNROLL; u++) if (k[u] != 0xffffffffu) atomicAdd(&lds[k[u]], 1u);
        }
        __syncthreads();
        u32* o = off + (size_t)w * (((size_t)NA << LBL) + 1) + ((size_t)khi << LB);
        for (unsigned j = tid; j < NL; j += 1024) if (lds[j]) atomicAdd(&o[j], lds[j]);
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024)
void k_big_scan(u32* __restrict__ off, u32* __restrict__ cur, const u32* __restrict__ offA,
                const u32* __restrict__ nbig, const u32* __restrict__ list, unsigned NA, unsigned LBL, unsigned short_from)
{
    __shared__ u32 part[1024];
    const unsigned tid = threadIdx.x;
    for (unsigned b = blockIdx.x; b < *nbig; b += gridDim.x) {
        const unsigned w = list[b] / NA, khi = list[b] % NA, LB = window_lb(LBL, w, short_from), NL = 1u << LB;
        const u32 begin = offA[(size_t)w * (NA + 1) + khi];
        const size_t base = (size_t)w * (((size_t)NA << LBL) + 1) + ((size_t)khi << LB);
        const unsigned per = (NL + 1023) / 1024, lo = min(NL, tid * per), hi = min(NL, lo + per);
        u32 sum = 0;
        for (unsigned j = lo; j < hi; j++) sum += off[base + j];
        part[tid] = sum;
        __syncthreads();
        for (unsigned d = 1; d < 1024; d <<= 1) {
            u32 v = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        u32 run = begin + part[tid] - sum;
        for (unsigned j = lo; j < hi; j++) { u32 c = off[base + j]; off[base + j] = run; cur[base + j] = run; run += c; }
        __syncthreads();
    }
}

// Slices of at most BIG_STAGE entries (the one-window plan: every partition listed, ~45 slices each) are read ONCE
// into registers, reserved per bucket, put in bucket order in LDS together with their destinations and written out as
// runs -- the direct form below reads the slice twice and stores 4 bytes per lane to 2^LB open rows (7.1 ms for
// 11 x 2^26 entries, staged: see profiles/r03_msm_fixed_base.log).  LDS: 2^LB cursors + 2^LB (global - local) offsets
// + 16 scan words + BIG_STAGE (index, destination) pairs.
static constexpr int BIG_PER = 8;
static constexpr unsigned BIG_STAGE = BIG_PER * 1024;
static inline size_t big_scatter_lds(unsigned LB) { return ((size_t)2 << LB) * 4 + 64 + (size_t)BIG_STAGE * 8 + PARTA_MAX_GROUPS * 4; }

template<bool PK>
__global__ __launch_bounds__(1024)
void k_big_scatter(u32* __restrict__ sorted, u32* __restrict__ cur, const typename recA<PK>::type* __restrict__ partA,
                   const u32* __restrict__ offA, const u32* __restrict__ nbig, const u32* __restrict__ list,
                   unsigned n, unsigned NA, unsigned LBL, unsigned short_from, unsigned split, partA_fmt fmt)
{
    extern __shared__ u32 lds[];
    typedef typename recA<PK>::type rec_t;
    const unsigned tid = threadIdx.x;
    const u32 kmask = (1u << LBL) - 1;
    u32* const gdl = lds + ((size_t)1 << LBL);              // (global - local) start of the slice's run of bucket j
    u32* const wsum = gdl + ((size_t)1 << LBL);
    u32* const st_idx = wsum + 16;
    u32* const st_dst = st_idx + BIG_STAGE;
    u32* const bnd = st_dst + BIG_STAGE;                    // the index-group boundaries of the slice's partition
    for (unsigned item = blockIdx.x; ; item += gridDim.x) {
        unsigned w, khi, lo, hi, begin;
        if (!big_slice(nbig, list, offA, NA, item, split, w, khi, lo, hi, begin)) return;
        const unsigned LB = window_lb(LBL, w, short_from), NL = 1u << LB;
        for (unsigned j = tid; j < NL; j += 1024) lds[j] = 0;
        if (PK) partA_bounds(bnd, fmt, w, khi, NA, tid);
        __syncthreads();
        const rec_t* src = partA + (size_t)w * n;
        u32* c = cur + (size_t)w * (((size_t)NA << LBL) + 1) + ((size_t)khi << LB);
        u32* dst = sorted + (size_t)w * n;
        if (hi - lo <= BIG_STAGE) {                         // uniform over the work-group
            rec_t r[BIG_PER];
            #pragma unroll
            for (int u = 0; u < BIG_PER; u++) { unsigned jx = lo + u * 1024 + tid; if (jx < hi) r[u] = src[jx]; }
            #pragma unroll
            for (int u = 0; u < BIG_PER; u++) if (lo + u * 1024 + tid < hi) atomicAdd(&lds[recA<PK>::key(r[u], kmask)], 1u);
            __syncthreads();
            // local exclusive scan of the bucket counts; one global reservation per occupied bucket
            const unsigned per = (NL + 1023) / 1024, b0 = min(NL, tid * per), b1 = min(NL, b0 + per);
            u32 sum = 0;
            #pragma unroll 1
            for (unsigned b = b0; b < b1; b++) sum += lds[b];
            u32 all;
            u32 run = block_scan_excl(sum, wsum, &all);
            #pragma unroll 1
            for (unsigned b = b0; b < b1; b++) {
                const u32 cb = lds[b];
                if (cb) gdl[b] = atomicAdd(&c[b], cb) - run;
                lds[b] = run; run += cb;
            }
            __syncthreads();
            #pragma unroll
            for (int u = 0; u < BIG_PER; u++)
                if (lo + u * 1024 + tid < hi) {
                    const u32 k = recA<PK>::key(r[u], kmask), pos = atomicAdd(&lds[k], 1u);
                    st_idx[pos] = recA<PK>::entry(r[u], lo + u * 1024 + tid - begin, LBL, fmt, bnd); st_dst[pos] = pos + gdl[k];
                }
            __syncthreads();
            for (unsigned i = tid; i < hi - lo; i += 1024) dst[st_dst[i]] = st_idx[i];
            __syncthreads();
            continue;
        }
        for (unsigned i = lo + tid; i < hi; i += BIG_UNROLL * 1024) {
            u32 k[BIG_UNROLL];
            #pragma unroll
            for (int u = 0; u < BIG_UNROLL; u++) { unsigned jx = i + u * 1024; k[u] = jx < hi ? recA<PK>::key(src[jx], kmask) : 0xffffffffu; }
            #pragma unroll
            for (int u = 0; u < BIG_UNROLL; u++) if (k[u] != 0xffffffffu) atomicAdd(&lds[k[u]], 1u);
        }
        __syncthreads();
        for (unsigned jx = tid; jx < NL; jx += 1024) if (lds[jx]) lds[jx] = atomicAdd(&c[jx], lds[jx]);   // reserve a range
        __syncthreads();
        for (unsigned i = lo + tid; i < hi; i += BIG_UNROLL * 1024) {
            rec_t r[BIG_UNROLL];
            #pragma unroll
            for (int u = 0; u < BIG_UNROLL; u++) { unsigned jx = i + u * 1024; if (jx < hi) r[u] = src[jx]; }
            #pragma unroll
            for (int u = 0; u < BIG_UNROLL; u++)
                if (i + u * 1024 < hi) dst[atomicAdd(&lds[recA<PK>::key(r[u], kmask)], 1u)] = recA<PK>::entry(r[u], i + u * 1024 - begin, LBL, fmt, bnd);
        }
        __syncthreads();
    }
}

} // namespace sppark_amd
