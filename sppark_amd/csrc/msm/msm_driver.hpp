// Host driver of the MI355X MSM pipeline: plays the role of the reference's
// msm_t (msm/pippenger.cuh:325-728) -- window choice, device scratch, kernel
// sequence, error mapping -- re-planned for a 288 GB / 256-CU part:
//
//  * WINDOW GROUPS (a tunable, off by default).  The W windows can be cut into G groups that are
//    sorted and accumulated one after the other, the digits + counting sort of group g+1 on an
//    auxiliary stream (optionally confined to a few CUs, hipExtStreamCreateWithCUMask) beside the
//    bucket accumulation of group g on the main one; the record tree and the bucket sums still run
//    once, for all windows (their ~15 dependent launches are latency-bound).  Scratch for digits /
//    partitions / sorted lists is then two group-sized sets instead of W.  It does NOT make an MSM
//    faster on MI355X: the sort kernels were shrunk to fit beside k_accumulate on a CU (512 lanes,
//    16 VGPRs) and they do run concurrently, but the accumulation slows down by what the sort
//    gains -- measured for 1..12 groups, 4..256 CUs for the sort stream and three stream
//    priorities (profiles/r02_msm_groups.log).  (The reference overlaps its sort with the previous
//    batch's accumulation through a 3-stream flip-flop over POINT batches, pippenger.cuh:494-557.)
//  * CHUNKS.  Host-resident inputs (what mult_pippenger_inf's callers pass) are cut into
//    point chunks: chunk c+1 is copied to the device (scalars first) while chunk c is being
//    computed; every chunk is a complete MSM and the partial results are added on the host.
//    The same loop bounds the scratch memory: when the device cannot hold the scratch of the
//    whole MSM (or tune.max_scratch says so) the chunk is halved until it does, for host AND
//    device-resident inputs.  The reference streams 2^24-point strides for the same two
//    reasons (pippenger.cuh:454-459,494-557).
//  * no cooperative launches or device-global work counters (:157-223): kernel
//    boundaries are the only grid-wide synchronisation;
//  * the host part is O(windows): it receives one XYZZ sum per window and runs
//    the Horner recombination (the reference integrates 256 partial sums per
//    window on a host thread pool, :627-727).
//
// Inputs may be host or device pointers (cf. is_device_ptr, util/gpu_t.cuh:385-395,
// and the preloaded-points constructor :351-388).
#pragma once
#include "msm_kernels.hpp"
#include "msm_coop_kernels.hpp"
#include "msm_plan.hpp"
#include "../ec/jacobian_host.hpp"
#include "../util/runtime.hpp"
#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

namespace sppark_amd {

// FD: device coordinate field (fp_class<P> for G1, fp2_dev<P> for G2), FH: its host
// twin with the same memory image (mont_host<P> / fp2_host<P>), FRp: scalar field.
template<class FD, class FH, class FRp>
class msm_t {
public:
    typedef FD fp_d;
    // the coordinate field is Fp2 over the loosely-reduced field: the cooperative accumulation exists for it
    static constexpr bool G2_COOP_BUILT = field_is_internal<FD>::value && !field_is_montx<FD>::value;
    typedef mont_dev<FRp> fr_d;
    typedef FH fp_h;
    typedef typename xyzz_dev<fp_d>::mem_t bucket_t;       // memory image: wire format
    typedef jacobian_host<fp_h> point_t;
    static constexpr int STD_WORDS = sizeof(FH) / 4;        // 32-bit words of a coordinate in the reference's wire form
    static constexpr size_t FP_BYTES = sizeof(FH);
    static constexpr bool INTERNAL = field_is_internal<FD>::value;      // ff/montx_dev.hpp: own point/bucket records
    static constexpr bool MONTX = field_is_montx<FD>::value;            // ... over the base field (G1): low-latency kernels, fixed-base tables
    typedef xyzz_mem<STD_WORDS> std_bucket_t;
    static_assert(INTERNAL || sizeof(FH) == 4 * FD::N, "host and device coordinate fields must share the wire image");
    static constexpr size_t SCALAR_BYTES = sizeof(fr_d);
    static constexpr unsigned MAX_WINS = 128;
    // one wave per SIMD on 256 CUs: bucket-sum grids up to this size run the paired-product one-wave kernels.  (Larger,
    // work-bound grids are faster with the two-wave kernels: forcing the one-wave ones everywhere costs the tail of a
    // 2^26-point MSM 2.0 ms and 0.5 ms at 2^24, profiles/r04_msm_lat_lanes_negative.log.)
    static constexpr size_t LAT_LANES = 65536;
    static constexpr size_t FIXED_BASE_MIN = (size_t)1 << 23;      // points from which set_points_fixed_base builds tables by itself

private:
    const gpu_info* gpu;
    hipStream_t stream;                 // main stream: the caller's, or a private one
    bool own_stream;
    hipStream_t aux = nullptr;          // sort of the next window group
    unsigned gseq = 0;                  // running window-group counter: parity selects the buffer set
    hipStream_t cpy = nullptr;          // host -> device copies of the next chunk
    hipEvent_t join_ev = nullptr;
    hipEvent_t ev_fork = nullptr, ev_sorted[2] = {nullptr, nullptr}, ev_accdone[2] = {nullptr, nullptr};
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_chunkdone[2] = {nullptr, nullptr};
    unsigned char* blob = nullptr;
    size_t blob_sz = 0;
    unsigned char* stage = nullptr;     // two staging sets for host-resident chunks
    size_t stage_sz = 0;
    std_bucket_t* h_sums = nullptr;     // pinned: window sums of every chunk in flight
    size_t h_sums_cap = 0;
    u32* h_flag = nullptr;              // pinned: "a bucket had more pieces than the piece tree takes" of the last MSM (enqueue)
    const u32* h_flag_cur = nullptr;    // where invoke() reads that flag: |h_flag|, or the word behind the window sums (the small sizes)
    u32* d_pflag = nullptr;             // the small sizes' flag on the device: zero between MSMs (the kernel that hands it over clears it)
    bool piece_pending = false;         // the last enqueue ran the piece tree and left the fan-in tree out
    std::vector<hipEvent_t> tev;        // timing events: [0] start, [1] end, [2+2g], [3+2g] around k_accumulate of group g
    unsigned char* pre_points = nullptr;    // points kept on the device by preload() (msm_t ctor with points, pippenger.cuh:351-385)
    size_t pre_n = 0, pre_stride = 0;
    unsigned pre_fb_wbits = 0, pre_fb_nwins = 0;    // fixed-base tables: window bits / windows they were built for (0: none)
    float last_ms[4] = {0, 0, 0, 0};    // [0] before the first accumulation, [1] accumulation kernels, [2] whole device part, [3] accumulate launches
    unsigned last_chunks = 0;
    unsigned last_redo = 0;             // invocations whose tail ran twice (the piece tree met a bucket beyond its limit)
    bool timing = false;

    struct layout {
        size_t digits[2], sorted[2], partA[2], H[2], tot[2], offA[2], off[2], curB[2], bigl[2];
        size_t buckets, keyA, ptA, keyB, ptB, keyC, flag, A1, W1, A2, W2, conv, sums, total;
    };

    static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
    static constexpr size_t conv_stride()
    {
        if constexpr (INTERNAL) return affine_loader<FD>::STRIDE; else return 0;
    }

    layout make_layout(const msm_plan& p, bool convert) const
    {
        layout l; size_t o = 0;
        auto take = [&](size_t sz) { size_t r = o; o += align_up(sz); return r; };
        const size_t wg = p.wpg;
        for (unsigned b = 0; b < 2; b++) {
            if (b == 1 && p.G == 1) {               // a single group needs one set
                l.digits[1] = l.digits[0]; l.sorted[1] = l.sorted[0]; l.partA[1] = l.partA[0]; l.H[1] = l.H[0]; l.tot[1] = l.tot[0];
                l.offA[1] = l.offA[0]; l.off[1] = l.off[0]; l.curB[1] = l.curB[0]; l.bigl[1] = l.bigl[0];
                break;
            }
            l.digits[b] = take(wg * p.n * 4);
            l.sorted[b] = take(wg * p.n * 4);
            l.partA[b]  = take(wg * p.n * (p.IB ? 4 : 8));                         // level-A records: packed / wide
            l.H[b]      = take(wg * p.nslabs * p.NA * 4);
            l.tot[b]    = take(wg * p.NA * 4);
            l.offA[b]   = take(wg * (p.NA + 1) * 4);
            l.off[b]    = take(wg * ((size_t)p.NB + 1) * 4);
            l.curB[b]   = take(wg * ((size_t)p.NB + 1) * 4);                    // cursors of the cooperative sort
            l.bigl[b]   = take((wg * p.NA + 1) * 4);                            // [count | list of oversized partitions]
        }
        // buckets, records and bucket-sum levels: all windows (one record tree / bucket-sum chain per MSM)
        l.buckets = take((size_t)p.nwins * p.NB * sizeof(bucket_t));
        size_t nrecA = (size_t)2 * p.nwins * p.chunks_per_win;
        size_t nrecB = 2 * ((nrecA + p.F - 1) / p.F);
        l.keyA = take(nrecA * 4); l.ptA = take(nrecA * sizeof(bucket_t));
        l.keyB = take(nrecB * 4); l.ptB = take(nrecB * sizeof(bucket_t));
        l.keyC = take(nrecA * 4); l.flag = take(4);                 // k_join_runs: filtered keys, "a long segment exists"
        size_t n1 = (size_t)p.nwins * std::max<size_t>(p.NB / p.K1, lg2_floor(p.NB) + 1);      // (the small windows' m + 1 parts: k_bucket_small_bits_coop)
        l.A1 = take(n1 * sizeof(bucket_t)); l.W1 = take(n1 * sizeof(bucket_t));
        l.A2 = take(n1 * sizeof(bucket_t)); l.W2 = take(n1 * sizeof(bucket_t));
        l.conv = take(INTERNAL && convert ? (size_t)p.n * conv_stride() : 0);     // points in the field's own records
        l.sums = take(((size_t)p.nwins + 1) * std::max(sizeof(std_bucket_t), sizeof(bucket_t)));     // (+ the small sizes' flag word)
        l.total = o;
        return l;
    }

    void reserve(size_t sz)
    {
        if (sz <= blob_sz) return;
        if (blob) {
            HIP_OK(hipStreamSynchronize(stream));
            if (aux) HIP_OK(hipStreamSynchronize(aux));
            HIP_OK(hipFree(blob)); blob = nullptr; blob_sz = 0;
        }
        HIP_OK(dev_scratch_pool::malloc_or_drain((void**)&blob, sz));      // (idle staging buffers of this library go first)
        blob_sz = sz;
    }
    void reserve_stage(size_t sz)
    {
        if (sz <= stage_sz) return;
        if (stage) { HIP_OK(hipDeviceSynchronize()); HIP_OK(hipFree(stage)); stage = nullptr; stage_sz = 0; }
        HIP_OK(dev_scratch_pool::malloc_or_drain((void**)&stage, sz));
        stage_sz = sz;
    }
    void reserve_sums(size_t count)
    {
        if (count <= h_sums_cap) return;
        if (h_sums) { HIP_OK(hipStreamSynchronize(stream)); HIP_OK(hipHostFree(h_sums)); h_sums = nullptr; h_sums_cap = 0; }
        HIP_OK(hipHostMalloc((void**)&h_sums, count * sizeof(std_bucket_t), hipHostMallocDefault));
        h_sums_cap = count;
    }
    void need_event(hipEvent_t& e) { if (!e) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
    // hipFuncAttributeMaxDynamicSharedMemorySize of a kernel, raised at most once per (context, kernel, size step):
    // the call costs a few microseconds of host time on every launch path otherwise
    // (the record lives in the CONTEXT: a process-wide one would be stale after a hipDeviceReset, and the next launch
    // with more than 64 KB of dynamic LDS would fail; a context does not outlive its device)
    std::map<const void*, size_t> lds_done;
    void lds_attr(const void* fn, size_t bytes)
    {
        size_t& have = lds_done[fn];
        if (bytes <= have) return;
        HIP_OK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        have = bytes;
    }
    void need_aux()
    {
        if (aux) return;
        // The sort of the following window groups: an ordinary stream on all CUs.  (Tuning builds, -DSPPARK_TUNING:
        // SPPARK_MSM_AUX_CUS=N confines it to N compute units.)
        unsigned ncu = 0;
#ifdef SPPARK_TUNING
        if (const char* e = getenv("SPPARK_MSM_AUX_CUS")) ncu = (unsigned)atoi(e);
#endif
        const unsigned total = (unsigned)gpu->prop.multiProcessorCount;
        hipError_t err = hipErrorNotSupported;
        if (ncu && ncu < total) {
            std::vector<uint32_t> mask((total + 31) / 32, 0);
            for (unsigned k = 0; k < ncu; k++) {        // spread over the whole CU index range (all XCDs / shader engines)
                unsigned cu = (unsigned)(((uint64_t)k * total) / ncu);
                mask[cu / 32] |= 1u << (cu % 32);
            }
            err = hipExtStreamCreateWithCUMask(&aux, (uint32_t)mask.size(), mask.data());
            if (err != hipSuccess) { (void)hipGetLastError(); aux = nullptr; }
        }
        if (!aux) HIP_OK(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
        need_event(ev_fork);
        for (int b = 0; b < 2; b++) { need_event(ev_sorted[b]); need_event(ev_accdone[b]); }
    }
    void need_cpy()
    {
        if (cpy) return;
        HIP_OK(hipStreamCreateWithFlags(&cpy, hipStreamNonBlocking));
        for (int b = 0; b < 2; b++) { need_event(ev_copied[b]); need_event(ev_chunkdone[b]); }
    }
    void need_tev(size_t count)
    {
        while (tev.size() < count) { hipEvent_t e; HIP_OK(hipEventCreate(&e)); tev.push_back(e); }
    }

public:
    msm_tunables tune;

    explicit msm_t(int device_id = -1, hipStream_t s = nullptr)
        : gpu(&select_gpu(device_id)), stream(s), own_stream(false)
    {
        {
            // lanes of k_accumulate the device runs at once -- at most two waves per SIMD (two 256-lane work-groups per CU):
            // a third resident wave adds no throughput -- : the plan fits the accumulation's grid to whole rounds of them
            // (asked on the context's OWN device, not whichever one happens to be current: the answer feeds the plan)
            int nb = 0, cur = -1;
            (void)hipGetDevice(&cur);
            if (cur != gpu->hip_id) HIP_OK(hipSetDevice(gpu->hip_id));
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_accumulate<fp_d, false>, 256, 0) == hipSuccess && nb > 0)
                tune.resident_lanes = (size_t)std::min(nb, 2) * 256 * (size_t)gpu->prop.multiProcessorCount;
            else (void)hipGetLastError();
            if (cur >= 0 && cur != gpu->hip_id) (void)hipSetDevice(cur);
        }
        if constexpr (G2_COOP_BUILT) { if (fp_d::NL >= 14) tune.long_runs = 1; }       // (the wave-pair accumulation's run lengths: msm_plan.hpp)
        if (stream == nullptr) {
            // a non-blocking private stream (a blocking one pays an implicit legacy-stream check on
            // every launch: +36 ms on a 2^26 MSM); join_default_stream() orders each call after the
            // work already queued on the legacy default stream instead
            HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            HIP_OK(hipEventCreateWithFlags(&join_ev, hipEventDisableTiming));
            own_stream = true;
        }
    }
    ~msm_t()
    {
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != gpu->hip_id) (void)hipSetDevice(gpu->hip_id);
        (void)hipStreamSynchronize(stream);
        if (aux) (void)hipStreamSynchronize(aux);
        if (cpy) (void)hipStreamSynchronize(cpy);
        if (blob) (void)hipFree(blob);
        if (stage) (void)hipFree(stage);
        if (h_sums) (void)hipHostFree(h_sums);
        if (h_flag) (void)hipHostFree(h_flag);
        if (d_pflag) (void)hipFree(d_pflag);
        if (pre_points) (void)hipFree(pre_points);
        for (auto& e : tev) (void)hipEventDestroy(e);
        for (hipEvent_t e : {ev_fork, ev_sorted[0], ev_sorted[1], ev_accdone[0], ev_accdone[1],
                             ev_copied[0], ev_copied[1], ev_chunkdone[0], ev_chunkdone[1], join_ev})
            if (e) (void)hipEventDestroy(e);
        if (aux) (void)hipStreamDestroy(aux);
        if (cpy) (void)hipStreamDestroy(cpy);
        if (own_stream) (void)hipStreamDestroy(stream);
        if (cur >= 0 && cur != gpu->hip_id) (void)hipSetDevice(cur);
    }
    msm_t(const msm_t&) = delete;
    msm_t& operator=(const msm_t&) = delete;

    int device() const { return gpu->hip_id; }
    void set_stream(hipStream_t s)
    {
        (void)hipStreamSynchronize(stream);
        if (own_stream) { (void)hipStreamDestroy(stream); own_stream = false; }
        stream = s;
        if (stream == nullptr) {
            HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            if (!join_ev) HIP_OK(hipEventCreateWithFlags(&join_ev, hipEventDisableTiming));
            own_stream = true;
        }
    }
    void enable_timing(bool on) { timing = on; }
    float kernel_ms(int which) const { return which >= 0 && which < 4 ? last_ms[which] : -1.f; }
    unsigned chunks_of_last_invoke() const { return last_chunks; }
    unsigned tail_redone() const { return last_redo; }
    size_t scratch_bytes() const { return blob_sz + stage_sz; }
    void release_scratch()
    {
        (void)hipSetDevice(gpu->hip_id);
        (void)hipStreamSynchronize(stream);
        if (aux) (void)hipStreamSynchronize(aux);
        if (cpy) (void)hipStreamSynchronize(cpy);
        if (blob) { (void)hipFree(blob); blob = nullptr; blob_sz = 0; }
        if (stage) { (void)hipFree(stage); stage = nullptr; stage_sz = 0; }
    }
    msm_plan plan_for(size_t npoints) const { return make_plan(npoints, FRp::NBITS, tune); }

    // Size the scratch for |npoints| ahead of time (so a timed invoke does not allocate).
    void reserve_for(size_t npoints, size_t ffi_affine_sz, bool host_points, bool host_scalars)
    {
        HIP_OK(hipSetDevice(gpu->hip_id));
        size_t chunk = choose_chunk(npoints, (host_points ? ffi_affine_sz : 0) + (host_scalars ? SCALAR_BYTES : 0));
        // as invoke(): one plan per chunk length, scratch for the largest layout
        const std::vector<size_t> cb = chunk_bounds(npoints, chunk);
        size_t need = 0;
        for (size_t c = 0; c + 1 < cb.size(); c++) need = std::max(need, make_layout(make_plan(cb[c + 1] - cb[c], FRp::NBITS, tune), true).total);
        reserve(need);
        if (host_points || host_scalars)
            reserve_stage(2 * (align_up(host_points ? chunk * ffi_affine_sz : 0) + align_up(host_scalars ? chunk * SCALAR_BYTES : 0)));
    }

    // Private stream only: wait for everything queued so far on the legacy default stream, where
    // most callers (PyTorch included) produce device-resident inputs.  Callers that work on other
    // streams pass theirs to the constructor / set_stream().
    void join_default_stream()
    {
        if (!own_stream) return;
        HIP_OK(hipEventRecord(join_ev, nullptr));
        HIP_OK(hipStreamWaitEvent(stream, join_ev, 0));
    }

    // Keep a copy of |np| points in HBM for later invoke(out, nullptr, n <= np, scalars, ...)
    // calls: the reference's msm_t(points, np, ffi_affine_sz) + invoke(out, scalars)
    // (pippenger.cuh:351-385,604-605).  |points| may be a host or a device pointer; np == 0 drops the copy.
    // window of the fixed-base tables: the c <= 26 with the least arithmetic -- W(c) * np mixed additions into the buckets
    // + two full additions (~3 mixed ones) for each of the 2^(c-1) buckets of the ONE bucket set (2^26 points: c = 24,
    // W = 11; c = 26, W = 10 costs the same and 4 x the buckets) -- or the forced width
    unsigned fixed_base_width(size_t np) const
    {
        if (tune.wbits) return std::min(26u, std::max(8u, tune.wbits));
        unsigned w = 0; double best = 0;
        for (unsigned c = 8; c <= 26; c++) {
            const double cost = (double)((FRp::NBITS - 1) / c + 1) * (double)np + 3.0 * (double)((size_t)1 << (c - 1));
            if (w == 0 || cost < best) { best = cost; w = c; }
        }
        return w;
    }
    // |fixed_base| (fields with their own records only): also keep the multiples 2^(off_j) * P_i of every point for
    // every window j (k_fixed_base_table), nwins x the memory; invoke(out, nullptr, np, ...) over exactly these np points
    // then runs as ONE window over nwins * np (digit, multiple) pairs -- fixed_plan() / invoke_fixed().
    void preload(const void* points, size_t np, size_t ffi_affine_sz, bool fixed_base = false)
    {
        HIP_OK(hipSetDevice(gpu->hip_id));
        HIP_OK(hipStreamSynchronize(stream));
        join_default_stream();
        // (arguments are checked before the previous set is dropped: a refused call leaves the context as it was)
        if (np != 0 && (points == nullptr || ffi_affine_sz < 2 * FP_BYTES || np > (1u << 31))) HIP_OK(hipErrorInvalidValue);
        if (np != 0 && fixed_base && !MONTX) HIP_OK(hipErrorNotSupported);
        if (np != 0 && fixed_base && (tune.wbits || np >= FIXED_BASE_MIN)
            && (size_t)((FRp::NBITS - 1) / fixed_base_width(np) + 1) * np >= ((size_t)1 << 31)) HIP_OK(hipErrorInvalidValue);
        if (pre_points) { HIP_OK(hipFree(pre_points)); pre_points = nullptr; pre_n = pre_stride = 0; }
        pre_fb_wbits = pre_fb_nwins = 0;
        if (np == 0) return;
        if (fixed_base && !MONTX) HIP_OK(hipErrorNotSupported);
        // measured (profiles/r03_msm_fixed_base.log): the one-window MSM wins from 2^24 points on and loses below 2^23
        // (the tables are gathered from HBM without reuse; the plain path's points are shared by all its windows).  Below the
        // threshold the call is a plain preload unless a window width was forced (tests, measurements).
        if (fixed_base && !tune.wbits && np < FIXED_BASE_MIN) fixed_base = false;
        unsigned fb_w = 0, fb_nw = 1;
        if (fixed_base) {
            fb_w = fixed_base_width(np);
            fb_nw = (FRp::NBITS - 1) / fb_w + 1;
            fb_w = FRp::NBITS / fb_nw + (FRp::NBITS % fb_nw ? 1 : 0);
            if ((size_t)fb_nw * np >= ((size_t)1 << 31)) HIP_OK(hipErrorInvalidValue);
        }
        if (points == nullptr || ffi_affine_sz < 2 * FP_BYTES || np > (1u << 31)) HIP_OK(hipErrorInvalidValue);
        struct dev_buf {                // staging copy of host-resident points: freed on every exit path
            unsigned char* p = nullptr;
            ~dev_buf() { if (p) (void)hipFree(p); }
        } staging;
        try {
            if constexpr (INTERNAL) {
                // bases are constant across invocations: convert them into the field's own records once
                const unsigned char* src = (const unsigned char*)points;
                if (!is_device_pointer(points)) {
                    HIP_OK(hipMalloc((void**)&staging.p, np * ffi_affine_sz));
                    HIP_OK(hipMemcpyAsync(staging.p, points, np * ffi_affine_sz, hipMemcpyHostToDevice, stream));
                    src = staging.p;
                }
                HIP_OK(dev_scratch_pool::malloc_or_drain((void**)&pre_points, (size_t)fb_nw * np * conv_stride()));
                launch_convert(pre_points, src, (unsigned)np, ffi_affine_sz);
                if constexpr (MONTX) if (fixed_base) {
                    hipLaunchKernelGGL(k_fixed_base_table<fp_d>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, stream,
                                       pre_points, (unsigned)np, fb_nw, (unsigned)FRp::NBITS);
                    HIP_OK(hipGetLastError());
                }
            } else {
                HIP_OK(hipMalloc((void**)&pre_points, np * ffi_affine_sz));
                HIP_OK(hipMemcpyAsync(pre_points, points, np * ffi_affine_sz,
                                      is_device_pointer(points) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
            }
            HIP_OK(hipStreamSynchronize(stream));
        } catch (...) {
            if (pre_points) { (void)hipFree(pre_points); pre_points = nullptr; }
            throw;
        }
        pre_n = np; pre_stride = ffi_affine_sz;
        if (fixed_base) { pre_fb_wbits = fb_w; pre_fb_nwins = fb_nw; }
    }
    unsigned fixed_base_windows() const { return pre_fb_nwins; }
    size_t preloaded() const { return pre_n; }

private:
    void launch_convert(unsigned char* dst, const unsigned char* src, unsigned n, size_t stride, hipStream_t on = nullptr)
    {
        if constexpr (INTERNAL) {
            if (on == nullptr) on = stream;
            unsigned grid = (n + 255) / 256;
            // G1 points in either wire layout at a 16-byte-aligned base: the coalesced form (msm_kernels.hpp k_convert_points_staged)
            if constexpr (MONTX) {
                if ((stride == 2 * FP_BYTES || stride == 2 * FP_BYTES + 8) && ((size_t)src & 15) == 0 && tune.join != 6) {
                    if (stride == 2 * FP_BYTES) hipLaunchKernelGGL((k_convert_points_staged<fp_d, false>), dim3(grid), dim3(256), 0, on, dst, src, n);
                    else                        hipLaunchKernelGGL((k_convert_points_staged<fp_d, true>), dim3(grid), dim3(256), 0, on, dst, src, n);
                    HIP_OK(hipGetLastError());
                    return;
                }
            }
            if (stride > 2 * FP_BYTES) hipLaunchKernelGGL((k_convert_points<fp_d, true>), dim3(grid), dim3(256), 0, on, dst, src, n, (unsigned)stride);
            else                       hipLaunchKernelGGL((k_convert_points<fp_d, false>), dim3(grid), dim3(256), 0, on, dst, src, n, (unsigned)stride);
            HIP_OK(hipGetLastError());
        }
    }

    // points per chunk.  |stage_per_point|: bytes of host-resident input per point that have to be
    // copied (0: everything is on the device); chunking then overlaps the copies with the
    // arithmetic.  Device-resident inputs are chunked only when the scratch would not fit.
    size_t choose_chunk(size_t n, size_t stage_per_point) const
    {
        size_t chunk = n;
        if (tune.chunk) chunk = std::min(n, std::max<size_t>(tune.chunk, 1));
        else if (stage_per_point && n > ((size_t)1 << 21)) {
            // 4 chunks of 2^20..2^24 points: the exposed first copy and last computation are 1/4 of
            // the total each, and every chunk is still an efficient MSM (a chunk pays its own bucket
            // sums: 2^21-point chunks cost 2^24 points 87 ms, 2^22-point chunks 75 ms;
            // profiles/r02_msm_host_path.log)
            chunk = std::min<size_t>(std::max<size_t>(n / 4, (size_t)1 << 20), (size_t)1 << 24);
        }
        auto need = [&](size_t c) {
            return make_layout(make_plan(c, FRp::NBITS, tune), true).total + 2 * c * stage_per_point + 1024;
        };
        size_t limit = tune.max_scratch;
        if (!limit) {
            // already reserved (both allocations, each on its own: the scratch blob and the staging sets
            // are separate hipMallocs): no driver query
            const bool blob_ok = make_layout(make_plan(chunk, FRp::NBITS, tune), true).total <= blob_sz;
            const bool stage_ok = stage_per_point == 0 || 2 * (chunk * stage_per_point + 512) <= stage_sz;  // two sets, each of two 256-byte aligned parts
            if (blob_ok && stage_ok) return chunk;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return chunk; }
            limit = free_b - (free_b >> 5) + blob_sz + stage_sz;        // what this context may hold in total
        }
        while (need(chunk) > limit && chunk > 4096) chunk = (chunk + 1) / 2;
        return chunk;
    }

    // Chunk boundaries [b_0 = 0, b_1, ..., b_k = n]: equal chunks, the last one shorter.
    // (Round 4 measured the alternatives for host-resident inputs on a host whose pinned H2D rate is 57 GB/s,
    // profiles/r04_msm_host_path.log: the pipeline behaves exactly like t = sum over chunks of max(copy, compute of the
    // previous chunk) + the last computation, with the copies AT the link rate -- 2^26 points: 163 ms of copies + one
    // exposed 2^24-point MSM = 201 ms.  Half-sized first / last chunks change nothing (the second-to-last computation is
    // exposed instead: 202 ms); the points copy in 2 or 4 slices from as many host threads and streams changes nothing
    // (201 ms: one pageable copy already saturates the link); eight 2^23-point chunks gain 5 % here (190 ms) and lost
    // 8 % on the host of round 3 (266 vs 245 ms, profiles/r03_msm_host_path.log), where small pageable copies are
    // slower per byte.  So: four chunks of at most 2^24 points, as the reference's 2^24-point strides.)
    std::vector<size_t> chunk_bounds(size_t n, size_t chunk) const
    {
        std::vector<size_t> b{0};
        for (size_t lo = chunk; lo < n; lo += chunk) b.push_back(lo);
        b.push_back(n);
        return b;
    }

    // ---- one complete MSM over device-resident data, asynchronously on stream (+ aux) ----------
    // d_points: wire points (stride, flagged) or, when preconverted, the field's own records.
    // h_out: nwins window sums in pinned host memory, valid once |stream| has been synchronised.
    // |fb_n| != 0: fixed-base mode -- |p| describes ONE window over fb_nwins * fb_n entries; the digits come from the
    // fb_n scalars cut into fb_nwins real windows (digit (w, i) is entry w * fb_n + i of the one window)
    void sort_group(hipStream_t ss, const msm_plan& p, const layout& l, unsigned b, unsigned w0, unsigned wn,
                    const u32* d_scalars, bool mont, unsigned fb_n = 0, unsigned fb_nwins = 0)
    {
        u32* digits = (u32*)(blob + l.digits[b]);
        u32* sorted = (u32*)(blob + l.sorted[b]);
        u32* H = (u32*)(blob + l.H[b]);
        u32* tot = (u32*)(blob + l.tot[b]);
        u32* off = (u32*)(blob + l.off[b]);
        void* partA = blob + l.partA[b];
        u32* offA = (u32*)(blob + l.offA[b]);
        // level-A records: 4 bytes where the plan allows it (msm_sort_kernels.hpp), PK = packed
        const bool packed = p.IB != 0;
        unsigned ngp = 1; while (ngp < p.NG) ngp <<= 1;
        const partA_fmt fmt{H, p.IB, p.SH, ngp, p.nslabs};
        // local index of the first short window of this group (window_len: the first nbits % nwins are long)
        const unsigned nlong = p.nbits % p.nwins, sf = nlong == 0 ? wn : (nlong > w0 ? std::min(wn, nlong - w0) : 0u);
        if (fb_n) {
            unsigned grid = std::min<unsigned>((fb_n + 255) / 256, 256 * 16);
            hipLaunchKernelGGL(k_breakdown<fr_d>, dim3(grid), dim3(256), 0, ss,
                               digits, d_scalars, fb_n, fb_nwins, (unsigned)FRp::NBITS, (int)mont, 0u, fb_nwins);
        } else {
            unsigned grid = std::min<unsigned>((p.n + 255) / 256, 256 * 16);
            hipLaunchKernelGGL(k_breakdown<fr_d>, dim3(grid), dim3(256), 0, ss,
                               digits, d_scalars, p.n, p.nwins, p.nbits, (int)mont, w0, wn);
        }
        HIP_OK(hipGetLastError());
        size_t ldsA = (size_t)p.NA * 4, ldsB = ((size_t)1 << p.LB) * 4 + SORT_NT * 4 + (size_t)SORTB_STAGE * 4 + PARTA_MAX_GROUPS * 4;
        // (the attribute is per device and sticky: raised once to the largest size asked for so far,
        // not on every MSM -- lds_attr())
        if (ldsA > 65536) { lds_attr((const void*)k_histA, ldsA); lds_attr(packed ? (const void*)k_scatterA<true> : (const void*)k_scatterA<false>, ldsA); }
        if (ldsB > 65536) lds_attr(packed ? (const void*)k_sortB<true> : (const void*)k_sortB<false>, ldsB);
        hipLaunchKernelGGL(k_histA, dim3(p.nslabs, wn), dim3(SORT_NT), ldsA, ss,
                           H, digits, p.n, p.nslabs, p.slab_sz, p.NA, p.LB, sf);
        HIP_OK(hipGetLastError());
        size_t na_total = (size_t)wn * p.NA;
        hipLaunchKernelGGL(k_scan_slabs, dim3((unsigned)((na_total + 255) / 256)), dim3(256), 0, ss,
                           H, tot, p.nslabs, p.NA, wn);
        HIP_OK(hipGetLastError());
        hipLaunchKernelGGL(k_scan_parts, dim3(wn), dim3(1024), 0, ss, offA, tot, p.NA);
        HIP_OK(hipGetLastError());
        if (p.NA <= SCATA_MAX_NA && p.LB < 16) {      // (always, with the automatic split: HB <= 12)
            const size_t ldsS = scatterA_staged_lds(p.NA);
            lds_attr(packed ? (const void*)k_scatterA_staged<true> : (const void*)k_scatterA_staged<false>, ldsS);
            if (packed) hipLaunchKernelGGL(k_scatterA_staged<true>, dim3(p.nslabs, wn), dim3(SORT_NT), ldsS, ss,
                                           (u32*)partA, digits, H, offA, p.n, p.nslabs, p.slab_sz, p.NA, p.LB, sf, p.IB);
            else        hipLaunchKernelGGL(k_scatterA_staged<false>, dim3(p.nslabs, wn), dim3(SORT_NT), ldsS, ss,
                                           (uint2*)partA, digits, H, offA, p.n, p.nslabs, p.slab_sz, p.NA, p.LB, sf, p.IB);
        } else {
            if (packed) hipLaunchKernelGGL(k_scatterA<true>, dim3(p.nslabs, wn), dim3(SORT_NT), ldsA, ss,
                                           (u32*)partA, digits, H, offA, p.n, p.nslabs, p.slab_sz, p.NA, p.LB, sf, p.IB);
            else        hipLaunchKernelGGL(k_scatterA<false>, dim3(p.nslabs, wn), dim3(SORT_NT), ldsA, ss,
                                           (uint2*)partA, digits, H, offA, p.n, p.nslabs, p.slab_sz, p.NA, p.LB, sf, p.IB);
        }
        HIP_OK(hipGetLastError());
        const unsigned big = tune.big ? tune.big : p.big ? p.big : (1u << 18);
        u32* nbig = (u32*)(blob + l.bigl[b]); u32* blist = nbig + 1; u32* curB = (u32*)(blob + l.curB[b]);
        // (a window of n entries cannot hold a partition above |big| when n <= big: the list of oversized partitions and its
        // three kernels -- which would find nothing and return, ~5 us of launch each -- are not even queued: 25 us of every
        // MSM up to 2^18 points)
        const bool may_be_big = p.n > big;
        if (may_be_big) HIP_OK(hipMemsetAsync(nbig, 0, 4, ss));
        if (packed) hipLaunchKernelGGL(k_sortB<true>, dim3(p.NA, wn), dim3(SORT_NT), ldsB, ss,
                                       sorted, off, (const u32*)partA, offA, p.n, p.NA, p.LB, sf, big, fmt);
        else        hipLaunchKernelGGL(k_sortB<false>, dim3(p.NA, wn), dim3(SORT_NT), ldsB, ss,
                                       sorted, off, (const uint2*)partA, offA, p.n, p.NA, p.LB, sf, big, fmt);
        HIP_OK(hipGetLastError());
        if (!may_be_big) return;
        // oversized partitions (skewed scalars); empty list and immediate return otherwise
        hipLaunchKernelGGL(k_big_find, dim3((p.NA * wn + 255) / 256), dim3(256), 0, ss,
                           nbig, blist, off, offA, p.NA, p.LB, sf, wn, big);
        size_t ldsBig = ((size_t)1 << p.LB) * 4;
        // slices per listed partition: 64 (one whole window in a partition), or about half of what k_big_scatter stages
        // in LDS when the AVERAGE partition is already near the threshold (the one-window plan of the fixed-base mode,
        // whose partitions differ by a factor of two)
        const size_t avg = (size_t)p.n / p.NA;
        const unsigned split = avg > big / 2 ? (unsigned)std::min<size_t>(SORTB_SPLIT, avg / (BIG_STAGE / 2) + 1) : SORTB_SPLIT;
        if (packed) hipLaunchKernelGGL(k_big_hist<true>, dim3(1024), dim3(1024), ldsBig, ss, off, (const u32*)partA, offA, nbig, blist, p.n, p.NA, p.LB, sf, split);
        else        hipLaunchKernelGGL(k_big_hist<false>, dim3(1024), dim3(1024), ldsBig, ss, off, (const uint2*)partA, offA, nbig, blist, p.n, p.NA, p.LB, sf, split);
        hipLaunchKernelGGL(k_big_scan, dim3(p.NA > 64 && avg > big / 2 ? 1024 : 64), dim3(1024), 0, ss, off, curB, offA, nbig, blist, p.NA, p.LB, sf);
        const size_t ldsSc = big_scatter_lds(p.LB);
        if (ldsSc > 65536) lds_attr(packed ? (const void*)k_big_scatter<true> : (const void*)k_big_scatter<false>, ldsSc);
        if (packed) hipLaunchKernelGGL(k_big_scatter<true>, dim3(1024), dim3(1024), ldsSc, ss, sorted, curB, (const u32*)partA, offA, nbig, blist, p.n, p.NA, p.LB, sf, split, fmt);
        else        hipLaunchKernelGGL(k_big_scatter<false>, dim3(1024), dim3(1024), ldsSc, ss, sorted, curB, (const uint2*)partA, offA, nbig, blist, p.n, p.NA, p.LB, sf, split, fmt);
        HIP_OK(hipGetLastError());
    }

    // |redo|: only the fan-in tree over the records the piece tree left, and everything after it (invoke(), when the flag of
    // the first pass says that a bucket had more pieces than the piece tree was sized for)
    void enqueue(const msm_plan& p, const layout& l, const unsigned char* d_points, size_t stride, bool preconverted,
                 const u32* d_scalars, bool mont, std_bucket_t* h_out, bool first_timed, unsigned fb_n = 0, unsigned fb_nwins = 0,
                 bool redo = false, bool may_defer = false)
    {
        const bool flagged = !preconverted && stride > 2 * FP_BYTES;
        const bool multi = p.G > 1;
        if (multi) need_aux();
        if (timing && first_timed && !redo) { need_tev(2 + 2 * p.G); HIP_OK(hipEventRecord(tev[0], stream)); }
        if (multi) {                                // the inputs are ready at this point of the main stream
            HIP_OK(hipEventRecord(ev_fork, stream));
            HIP_OK(hipStreamWaitEvent(aux, ev_fork, 0));
        }
        bucket_t* buckets = (bucket_t*)(blob + l.buckets);
        u32* keyA = (u32*)(blob + l.keyA); bucket_t* ptA = (bucket_t*)(blob + l.ptA);
        u32* keyB = (u32*)(blob + l.keyB); bucket_t* ptB = (bucket_t*)(blob + l.ptB);
        // With ONE window group the bucket offsets of every window are still there when the bucket sums run:
        // empty buckets are recognised from them and never read (k_bucket_level1), so no memset.  With several
        // groups the two offset sets are reused, and the buckets are cleared instead.
        if (multi && !redo) HIP_OK(hipMemsetAsync(buckets, 0, (size_t)p.nwins * p.NB * sizeof(bucket_t), stream));

        for (unsigned g = 0; g < p.G && !redo; g++, gseq++) {
            // sort sets by the parity of a counter that keeps running across MSMs (chunks): the
            // events of set b then always refer to the previous user of set b
            const unsigned b = multi ? (gseq & 1) : 0, w0 = g * p.wpg, wn = std::min(p.wpg, p.nwins - w0);
            if (g == 0) {
                // first group: on the main stream, at full width (stream order protects the set)
                sort_group(stream, p, l, b, w0, wn, d_scalars, mont, fb_n, fb_nwins);
                // (the conversion does not depend on the scalars, but running it on the second stream beside the digit /
                // sort kernels gains nothing: all of them are memory-bound and share HBM -- 13.4 ms before the accumulation
                // either way at 2^26, profiles/r04_msm_convert_beside_sort_negative.log; round 2 measured the same)
                if (INTERNAL && !preconverted) {    // wire points -> the field's own records (2 products per point)
                    launch_convert(blob + l.conv, d_points, p.n, stride);
                    d_points = blob + l.conv;
                }
            } else {
                if (gseq >= 2) HIP_OK(hipStreamWaitEvent(aux, ev_accdone[b], 0));    // the accumulation two groups back has read this set
                sort_group(aux, p, l, b, w0, wn, d_scalars, mont);
                HIP_OK(hipEventRecord(ev_sorted[b], aux));
                HIP_OK(hipStreamWaitEvent(stream, ev_sorted[b], 0));
            }
            if (timing && first_timed) HIP_OK(hipEventRecord(tev[2 + 2 * g], stream));
            {
                const u32* sorted = (const u32*)(blob + l.sorted[b]);
                const u32* off = (const u32*)(blob + l.off[b]);
                dim3 grid((p.chunks_per_win + 255) / 256, wn);
                // G2: one Fp2 component per wave (msm_g2c_kernels.hpp).  The default for the 14-limb base fields: BLS12-381 G2
                // 2^22 47.8 -> 40.2 ms, 2^20 15.3 -> 14.0; NOT for the 10-limb one, whose whole Fp2 bucket fits a lane at two
                // waves per SIMD already (alt_bn128 G2 2^22 21.6 -> 23.6 ms); profiles/r05_g2_coop_ab.log.
                // tune.g2_coop: 0 = that rule, 1 = wave pairs, 2 = one lane per addition (sppark_msm_g2_path).
                bool by_pairs = false;
                if constexpr (G2_COOP_BUILT) {
                    if (tune.g2_coop == 1 || (tune.g2_coop == 0 && fp_d::NL >= 14)) {
                        by_pairs = true;
                        dim3 grid2((p.chunks_per_win + 63) / 64, wn);
                        hipLaunchKernelGGL((k_accumulate_g2c<fp_d>), grid2, dim3(G2C_NT), 0, stream,
                                           buckets, keyA, ptA, d_points, (unsigned)stride, sorted, off,
                                           p.n, p.NB, p.L, p.chunks_per_win, w0);
                    }
                }
                // (fields with their own records read them at their own stride and ignore this one)
                if (by_pairs) {}
                else if (flagged)
                    hipLaunchKernelGGL((k_accumulate<fp_d, true>), grid, dim3(256), 0, stream,
                                       buckets, keyA, ptA, d_points, (unsigned)stride, sorted, off,
                                       p.n, p.NB, p.L, p.chunks_per_win, w0);
                else
                    hipLaunchKernelGGL((k_accumulate<fp_d, false>), grid, dim3(256), 0, stream,
                                       buckets, keyA, ptA, d_points, (unsigned)stride, sorted, off,
                                       p.n, p.NB, p.L, p.chunks_per_win, w0);
                HIP_OK(hipGetLastError());
            }
            if (timing && first_timed) HIP_OK(hipEventRecord(tev[3 + 2 * g], stream));
            if (multi) HIP_OK(hipEventRecord(ev_accdone[b], stream));
        }

        // ---- small MSMs: the pieces of every bucket by a tree over the bucket's own pieces (msm_piece_kernels.hpp) ----
        // Where a bucket is cut into MORE runs than k_join_runs walks (n / NB > 4 L: up to 2^16 points), log2(cmax) launches of
        // one addition each replace the fan-in tree's eleven of up to three (2^16: 0.29 -> 0.09 ms, 2^12: 0.20 -> 0.08).  A
        // bucket with more than cmax pieces (skewed scalars) keeps its records and raises the flag; the fan-in tree is NOT
        // queued behind it -- ten launches that find nothing to do are 50 us -- but run afterwards by invoke() when the flag,
        // which comes back with the window sums, is set.  (One window group, whose offsets are all still there.)
        // (|may_defer|: the caller looks at the flag after this MSM -- invoke() with one chunk)
        const unsigned piece_cm = may_defer ? piece_tree_cmax(p, multi, fb_n) : 0;
        piece_pending = false;
        // windows of up to 256 buckets (MSMs of up to 2^16 points): the subset sums straight from the buckets, then the parts of a
        // window (msm_coop_kernels.hpp k_bucket_small_bits_coop); needs the offsets of every window: one window group
        bool small_sums = false;
        if constexpr (MONTX)
            small_sums = !multi && fb_n == 0 && p.NB <= SMALL_SUMS_MAX_NB && p.NB >= 2 && tune.K1 == 0 && tune.top == 0 && tune.K == 0
                         && tune.join != 3 && tune.join != 4;
        // The flag of the piece tree on that path costs no launch of its own: it lives in a word that is ZERO between MSMs
        // (no memset), and the last kernel of the path -- k_bucket_top_sum_coop, which writes the window sums' wire image --
        // puts it behind the sums (one copy brings both to the host) and clears it.  (2^12: a 5 us fill with a 6 us gap in front
        // of the levels and a 5 us copy behind them, of a 0.39 ms MSM.)  Other paths: memset, levels, a copy of their own.
        bool flag_with_sums = false;
        if constexpr (MONTX) flag_with_sums = small_sums && (piece_cm != 0 || redo);
        if (flag_with_sums && !d_pflag) {
            HIP_OK(hipMalloc((void**)&d_pflag, 64));
            HIP_OK(hipMemsetAsync(d_pflag, 0, 64, stream));
        }
        if (piece_cm && !redo) {
            u32* flag = flag_with_sums ? d_pflag : (u32*)(blob + l.flag);
            if (!flag_with_sums) HIP_OK(hipMemsetAsync(flag, 0, 4, stream));
            const u32* off = (const u32*)(blob + l.off[0]);
            // the levels of few work items in one launch (k_piece_tail_coop; tune.join 8: every level a launch, 16 + x: from 2^x items)
            unsigned t_fused = ~0u;
            if constexpr (MONTX) if (tune.join != 4 && tune.join != 8)
                t_fused = piece_tail_t0((size_t)p.nwins * p.NB, piece_cm, tune.join >= 16 ? (size_t)1 << (tune.join - 16) : PIECE_FUSE_MAX);
            for (unsigned t = 0; (piece_cm >> (t + 1)) >= 1; t++) {
                const unsigned last = (piece_cm >> (t + 2)) == 0;
                const size_t nthr = (size_t)p.nwins * p.NB * (piece_cm >> (t + 1));
                if constexpr (MONTX) if (t == t_fused) {
                    const unsigned lgGB = piece_tail_lgGB(piece_cm, t);
                    const size_t nwg = (((size_t)p.nwins * p.NB) + ((size_t)1 << lgGB) - 1) >> lgGB;
                    hipLaunchKernelGGL(k_piece_tail_coop<fp_d>, dim3((unsigned)nwg), dim3(COOP_NT), 0, stream,
                                       buckets, keyA, ptA, off, p.NB, p.L, p.chunks_per_win, p.nwins, piece_cm, t, lgGB, flag);
                    HIP_OK(hipGetLastError());
                    break;
                }
                bool coop = false;
                if constexpr (MONTX) coop = nthr <= COOP_LEVEL_MAX && tune.join != 4;
                if constexpr (MONTX) {
                    if (coop) hipLaunchKernelGGL(k_piece_level_coop<fp_d>, dim3((unsigned)((nthr + 63) / 64)), dim3(COOP_NT), 0, stream,
                                                 buckets, keyA, ptA, off, p.NB, p.L, p.chunks_per_win, p.nwins, piece_cm, t, last, flag);
                }
                if (!coop) hipLaunchKernelGGL(k_piece_level<fp_d>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream,
                                              buckets, keyA, ptA, off, p.NB, p.L, p.chunks_per_win, p.nwins, piece_cm, t, last, flag);
                HIP_OK(hipGetLastError());
            }
            if (!h_flag) HIP_OK(hipHostMalloc((void**)&h_flag, 64, hipHostMallocDefault));
            if (!flag_with_sums) HIP_OK(hipMemcpyAsync(h_flag, flag, 4, hipMemcpyDeviceToHost, stream));
            h_flag_cur = flag_with_sums ? reinterpret_cast<const u32*>(h_out + p.nwins) : h_flag;
            piece_pending = true;
        }
        // ---- segmented record tree over the records of all windows -----------------------------
        if (!piece_pending) {
            size_t nrec = (size_t)2 * p.nwins * p.chunks_per_win;
            u32* ik = keyA; bucket_t* ip = ptA; u32* ok = keyB; bucket_t* op = ptB;
            // segments of <= JOIN_WALK records (with uniform scalars: all of them) in one launch; the tree
            // below then only sees the records of longer segments and returns at once when there are none
            const u32* skip = nullptr;
            // (not when the average bucket is longer than four runs: every segment is then longer than the join's walk
            // and the launch finds nothing to do -- below ~2^19 points)
            if (tune.join != 1 && !redo && (size_t)p.n / p.NB <= (size_t)4 * p.L) {
                u32* keyC = (u32*)(blob + l.keyC); u32* flag = (u32*)(blob + l.flag);
                HIP_OK(hipMemsetAsync(flag, 0, 4, stream));
                const size_t nthr = nrec / 2 + 1;
                hipLaunchKernelGGL(k_join_runs<fp_d>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream,
                                   buckets, keyC, keyA, ptA, (unsigned)nrec, flag);
                HIP_OK(hipGetLastError());
                ik = keyC; skip = flag;
            }
            bool coop_tree = false;
            if constexpr (MONTX) coop_tree = tune.join != 4 && tune.join != 2;
            for (;;) {
                unsigned nthreads = (unsigned)((nrec + p.F - 1) / p.F);
                if constexpr (MONTX) {
                    // from one work-group per CU on: four waves per addition (msm_coop_kernels.hpp)
                    if (coop_tree && nthreads <= 64) {
                        hipLaunchKernelGGL(k_reduce_tail_coop<fp_d>, dim3(1), dim3(COOP_NT), 0, stream,
                                           buckets, ik, ip, ok, op, (unsigned)nrec, p.F, skip);
                        HIP_OK(hipGetLastError());
                        break;
                    }
                    if (coop_tree && nthreads <= COOP_TREE_MAX) {
                        hipLaunchKernelGGL(k_reduce_runs_coop<fp_d>, dim3((nthreads + 63) / 64), dim3(COOP_NT), 0, stream,
                                           buckets, ok, op, ik, ip, (unsigned)nrec, p.F, nthreads, 0, skip);
                        HIP_OK(hipGetLastError());
                        nrec = (size_t)2 * nthreads;
                        std::swap(ik, ok); std::swap(ip, op);
                        continue;
                    }
                }
                if (nthreads <= REDUCE_TAIL_NT && tune.join != 2) {     // the narrow end: every remaining level in one launch
                    // (the level kernels write into the OTHER buffer pair; here the pairs alternate from |ik| on)
                    hipLaunchKernelGGL(k_reduce_tail<fp_d>, dim3(1), dim3(REDUCE_TAIL_NT), 0, stream,
                                       buckets, ik, ip, ok, op, (unsigned)nrec, p.F, skip);
                    HIP_OK(hipGetLastError());
                    break;
                }
                int last = nthreads == 1;
                hipLaunchKernelGGL(k_reduce_runs<fp_d>, dim3((nthreads + 255) / 256), dim3(256), 0, stream,
                                   buckets, ok, op, ik, ip, (unsigned)nrec, p.F, nthreads, last, skip);
                HIP_OK(hipGetLastError());
                if (last) break;
                nrec = (size_t)2 * nthreads;
                std::swap(ik, ok); std::swap(ip, op);
            }
        }
        // ---- per-window weighted bucket sums ----------------------------------------------------
        bucket_t* A1 = (bucket_t*)(blob + l.A1); bucket_t* W1 = (bucket_t*)(blob + l.W1);
        bucket_t* A2 = (bucket_t*)(blob + l.A2); bucket_t* W2 = (bucket_t*)(blob + l.W2);
        bucket_t* result;
        bool finalized = false;
        if constexpr (MONTX) {
            if (small_sums) {
                const unsigned m = lg2_floor(p.NB);
                hipLaunchKernelGGL(k_bucket_small_bits_coop<fp_d>, dim3(m + 1, p.nwins), dim3(COOP_NT), 0, stream,
                                   A2, buckets, (const u32*)(blob + l.off[0]), p.NB, m);
                HIP_OK(hipGetLastError());
                std_bucket_t* fin = (std_bucket_t*)(blob + l.sums);
                hipLaunchKernelGGL(k_bucket_top_sum_coop<fp_d>, dim3(p.nwins), dim3(COOP_NT), 0, stream, W2, A2, m + 1,
                                   fin,                                         // (the wire image with it: no k_finalize)
                                   flag_with_sums ? d_pflag : (u32*)nullptr, flag_with_sums ? reinterpret_cast<u32*>(fin + p.nwins) : (u32*)nullptr);
                HIP_OK(hipGetLastError());
                result = W2; finalized = true;
            }
        }
        if (!small_sums) {
            unsigned nitems = p.NB / p.K1;
            size_t nthr = (size_t)p.nwins * nitems;
            // grids of at most one resident round (one wave per SIMD: 65 536 lanes) are chains of dependent additions:
            // the _lat kernels (no register cap, products in pairs); larger ones are work: two waves per SIMD
            const u32* offp = multi ? (const u32*)nullptr : (const u32*)(blob + l.off[0]);
            bool lat = false;
            if constexpr (MONTX) lat = nthr <= LAT_LANES && tune.join != 3;
            if constexpr (MONTX) {
                // (at most one work-group of four waves per CU: four waves per operation, msm_coop_kernels.hpp)
                if (lat && nthr <= COOP_LEVEL_MAX && tune.join != 4)
                    hipLaunchKernelGGL(k_bucket_level1_coop<fp_d>, dim3((unsigned)((nthr + 63) / 64)), dim3(COOP_NT), 0, stream,
                                       A1, W1, buckets, p.NB, p.K1, p.nwins, offp);
                // (between that and one resident round of waves: the two chains of a work item on two waves; tune.join 10: on one)
                else if (lat && tune.join != 10 && tune.join != 4)
                    hipLaunchKernelGGL(k_bucket_level1_pipe<fp_d>, dim3((unsigned)((nthr + 63) / 64)), dim3(128), 0, stream,
                                       A1, W1, buckets, p.NB, p.K1, p.nwins, offp);
                else if (lat) hipLaunchKernelGGL(k_bucket_level1_lat<fp_d>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream,
                                                 A1, W1, buckets, p.NB, p.K1, p.nwins, offp);
            }
            if (!lat) hipLaunchKernelGGL(k_bucket_level1<fp_d>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream,
                                         A1, W1, buckets, p.NB, p.K1, p.nwins, offp);
            HIP_OK(hipGetLastError());
            unsigned lgG = lg2_floor(p.K1);
            bucket_t *ia = A1, *iw = W1, *oa = A2, *ow = W2;
            while (nitems > 1) {
                // the top of the sums by bit-weighted subset sums (msm_kernels.hpp k_bucket_top_bits): depth, not work
                if (nitems <= (tune.top ? tune.top : BUCKET_TOP_MAX) && nitems >= 32 && (nitems & (nitems - 1)) == 0
                    && (size_t)p.NB / p.K1 >= 32) {
                    const unsigned m = lg2_floor(nitems);
                    bool coop = false;
                    if constexpr (MONTX) coop = tune.join != 4;     // (4: the one-wave-per-operation kernels, A/B switch)
                    if constexpr (MONTX) {
                        if (coop) {
                            // the tree and the doubling chains by four waves per operation (msm_coop_kernels.hpp)
                            // (a work-group per PIECE of a sum: msm_kernels.hpp bucket_top_piece; join == 7: per sum, the A/B switch)
                            unsigned sb = 1, sp = 1;
                            if (tune.join != 7) bucket_top_cut(nitems, COOP_NT, sb, sp);
#ifdef SPPARK_TUNING
                            if (const char* e = getenv("SPPARK_TOP_CUT")) {         // "sb sp" as two digits, e.g. 24 (sweeps only)
                                const unsigned v = (unsigned)atoi(e), s1 = v / 10, s2 = v % 10;
                                if (s1 >= 1 && s2 >= 1 && m * s1 + s2 <= 32 && nitems >= COOP_NT * s2) { sb = s1; sp = s2; }
                            }
#endif
                            const size_t lds = top_bits_coop_lds(fp_d::N);
                            if (lds > 65536) lds_attr((const void*)k_bucket_top_bits_coop<fp_d>, lds);
                            hipLaunchKernelGGL(k_bucket_top_bits_coop<fp_d>, dim3(m * sb + sp, p.nwins), dim3(COOP_NT), lds, stream,
                                               oa, ia, iw, nitems, m, lgG, sb, sp);
                            HIP_OK(hipGetLastError());
                            hipLaunchKernelGGL(k_bucket_top_sum_coop<fp_d>, dim3(p.nwins), dim3(COOP_NT), 0, stream, ow, oa, m * sb + sp,
                                               (std_bucket_t*)(blob + l.sums),             // (the wire image with it: no k_finalize)
                                               (u32*)nullptr, (u32*)nullptr);
                            HIP_OK(hipGetLastError());
                            finalized = true;
                        }
                    }
                    if (!coop) {
                        const size_t img = (size_t)BUCKET_TOP_NT * sizeof(bucket_t);
                        if (img > 65536)
                            lds_attr((const void*)k_bucket_top_bits<fp_d>, img);
                        hipLaunchKernelGGL(k_bucket_top_bits<fp_d>, dim3(m + 1, p.nwins), dim3(BUCKET_TOP_NT), img, stream,
                                           oa, ia, iw, nitems, m, lgG);
                        HIP_OK(hipGetLastError());
                        hipLaunchKernelGGL(k_bucket_top_sum<fp_d>, dim3(p.nwins), dim3(32), 32 * sizeof(bucket_t), stream, ow, oa, m);
                        HIP_OK(hipGetLastError());
                    }
                    std::swap(iw, ow);
                    break;
                }
                unsigned K = std::min(p.K, nitems);
                nthr = (size_t)p.nwins * (nitems / K);
                lat = false;
                if constexpr (MONTX) lat = nthr <= LAT_LANES && tune.join != 3;
                if constexpr (MONTX) {
                    if (lat && nthr <= COOP_LEVEL_MAX && tune.join != 4)
                        hipLaunchKernelGGL(k_bucket_levelN_coop<fp_d>, dim3((unsigned)((nthr + 63) / 64)), dim3(COOP_NT), 0, stream,
                                           oa, ow, ia, iw, nitems, K, lgG, p.nwins);
                    else if (lat && tune.join != 10 && tune.join != 4)      // (its three sums on three waves; tune.join 10: on one lane)
                        hipLaunchKernelGGL(k_bucket_levelN_pipe<fp_d>, dim3((unsigned)((nthr + 63) / 64)), dim3(192), 0, stream,
                                           oa, ow, ia, iw, nitems, K, lgG, p.nwins);
                    else if (lat) hipLaunchKernelGGL(k_bucket_levelN_lat<fp_d>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream,
                                                     oa, ow, ia, iw, nitems, K, lgG, p.nwins);
                }
                if (!lat) hipLaunchKernelGGL(k_bucket_levelN<fp_d>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream,
                                             oa, ow, ia, iw, nitems, K, lgG, p.nwins);
                HIP_OK(hipGetLastError());
                nitems /= K; lgG += lg2_floor(K);
                std::swap(ia, oa); std::swap(iw, ow);
            }
            result = iw;
        }
        if (timing && first_timed && !redo) HIP_OK(hipEventRecord(tev[1], stream));
        // ---- device -> host: one XYZZ per window (wire image); Horner on the host ------------------
        if constexpr (INTERNAL) {
            std_bucket_t* fin = (std_bucket_t*)(blob + l.sums);
            if (!finalized) {
                hipLaunchKernelGGL((k_finalize<fp_d, STD_WORDS>), dim3((p.nwins + 63) / 64), dim3(64), 0, stream, fin, result, p.nwins);
                HIP_OK(hipGetLastError());
            }
            HIP_OK(hipMemcpyAsync(h_out, fin, (p.nwins + (flag_with_sums ? 1 : 0)) * sizeof(std_bucket_t), hipMemcpyDeviceToHost, stream));
        } else {
            HIP_OK(hipMemcpyAsync(h_out, result, p.nwins * sizeof(std_bucket_t), hipMemcpyDeviceToHost, stream));
        }
    }

public:
    // pieces per bucket the piece tree of a small MSM takes, 0 = the record list goes through k_join_runs / the fan-in tree:
    // one window group, not the fixed-base window, buckets longer than the join's walk, at most 2^10 pieces
    // (tune.join 5: never -- the A/B switch)
    unsigned piece_tree_cmax(const msm_plan& p, bool multi, unsigned fb_n) const
    {
        if (multi || fb_n || tune.join == 5 || tune.join == 1) return 0;
        if ((size_t)p.n / p.NB <= (size_t)4 * p.L) return 0;
        // The TOP window is not uniform even for uniform scalars: the recoding folds s > r/2 to r - s, so its digit is at most
        // (r/2) >> off_top and each of its buckets holds n 2^off_top / (r/2) entries -- BLS12-377's r = 0x12ab... in 4-bit
        // windows: 0.43 n in one bucket against the average n / 8.  The tree is sized for that bucket too (the other short
        // windows at the top are at most twice the average: within piece_cmax's head-room).
        const unsigned off_top = p.nbits - window_len(p.nwins - 1, p.nwins, p.nbits);
        long double r = 0;
        for (int i = FRp::N - 1; i >= 0; i--) r = r * 4294967296.0L + (long double)FRp::MOD[i];
        long double frac = ldexpl(1.0L, (int)off_top) / (r / 2);
        if (frac > 1.0L) frac = 1.0L;
        const size_t top_pieces = (size_t)((long double)p.n * frac / p.L) + 2;
        const unsigned c = std::max(piece_cmax((size_t)p.n / p.NB / p.L + 1), piece_cmax_exact(top_pieces + top_pieces / 4 + 4));
        return c <= 1024 ? c : 0;
    }
private:

    // Horner over the window sums (the reference's host-side collect, pippenger.cuh:627-727, is O(256 * windows))
    static point_t horner(const std_bucket_t* sums, const msm_plan& p)
    {
        point_t out; out.set_inf();
        for (unsigned w = p.nwins; w--;) {
            fp_h c[4];
            memcpy(c, &sums[w], sizeof(c));
            point_t s = point_t::from_xyzz(c[0], c[1], c[2], c[3]);
            out.add(s);
            if (w) for (unsigned k = 0; k < window_len(w - 1, p.nwins, p.nbits); k++) out.dbl();
        }
        return out;
    }

    // The plan of a fixed-base MSM: ONE window of |wbits| bits over fb_nwins * n entries (the multiples are baked into
    // the table, so every window's digits select from one bucket set).
    msm_plan fixed_plan(size_t n) const
    {   return make_fixed_plan(n, pre_fb_wbits, pre_fb_nwins, SORTB_STAGE, tune);   }

    // The one-window layout of the fixed-base mode needs ~16 B x windows x n of sort scratch in ONE piece (it cannot be
    // chunked: the tables cover the whole vector).  When the device cannot hold it next to the tables, invoke() runs the
    // ordinary chunked path on level 0 of the table instead of failing.
    bool fixed_fits(size_t npoints, const void* scalars) const
    {
        const size_t need = make_layout(fixed_plan(npoints), false).total
                          + (is_device_pointer(scalars) ? 0 : 2 * align_up(npoints * SCALAR_BYTES));
        if (need <= blob_sz + stage_sz) return true;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return true; }
        return need <= free_b - (free_b >> 5) + blob_sz + stage_sz;
    }

    // out = sum s_i * P_i over the preloaded points through their fixed-base tables: device-resident scalars or host
    // scalars (copied in one piece), one pass, one window.
    void invoke_fixed(point_t& out, size_t npoints, const void* scalars, bool mont)
    {
        const msm_plan p = fixed_plan(npoints);
        const layout l = make_layout(p, false);
        const bool sc_dev = is_device_pointer(scalars);
        reserve(l.total);
        reserve_sums(MAX_WINS);
        const u32* d_scalars = (const u32*)scalars;
        if (!sc_dev) {
            reserve_stage(2 * align_up(npoints * SCALAR_BYTES));
            HIP_OK(hipMemcpyAsync(stage, scalars, npoints * SCALAR_BYTES, hipMemcpyHostToDevice, stream));
            d_scalars = (const u32*)stage;
        }
        enqueue(p, l, pre_points, conv_stride(), true, d_scalars, mont, h_sums, true, (unsigned)npoints, pre_fb_nwins);
        HIP_OK(hipStreamSynchronize(stream));
        last_chunks = 1;
        if (timing) {
            float acc = 0;
            HIP_OK(hipEventElapsedTime(&acc, tev[2], tev[3]));
            HIP_OK(hipEventElapsedTime(&last_ms[0], tev[0], tev[2]));
            last_ms[1] = acc;
            HIP_OK(hipEventElapsedTime(&last_ms[2], tev[0], tev[1]));
            last_ms[3] = 1.f;
        }
        out = horner(h_sums, p);                            // one window: its sum is the result
    }

public:
    // Bitmap batch addition (msm/batch_addition.cuh:25-132): out = sum of the selected points.
    // points / bitmap / refmap (nullable): host or device pointers; maps are ceil(n/32) words.
    void batch_add(point_t& out, const void* points, size_t npoints, const uint32_t* bitmap, const uint32_t* refmap,
                   size_t ffi_affine_sz)
    {
        out.set_inf();
        if (npoints == 0) return;
        if (npoints > (1u << 31) || !points || !bitmap || ffi_affine_sz < 2 * FP_BYTES) HIP_OK(hipErrorInvalidValue);
        HIP_OK(hipSetDevice(gpu->hip_id));
        join_default_stream();
        const size_t nwords = (npoints + 31) / 32;
        const unsigned span = 4;                                    // 128 points per lane
        const size_t nrec = (nwords + span - 1) / span, F = 8;
        size_t o = 0;
        auto take = [&](size_t sz) { size_t r = o; o += align_up(sz); return r; };
        const bool pts_dev = is_device_pointer(points), bm_dev = is_device_pointer(bitmap), rm_dev = refmap && is_device_pointer(refmap);
        const size_t o_pts = take(pts_dev ? 0 : npoints * ffi_affine_sz), o_bm = take(bm_dev ? 0 : nwords * 4),
                     o_rm = take(refmap && !rm_dev ? nwords * 4 : 0), o_conv = take(INTERNAL ? npoints * conv_stride() : 0),
                     o_keyA = take(nrec * 4), o_ptA = take(nrec * sizeof(bucket_t)),
                     o_keyB = take((nrec / F + 2) * 2 * 4), o_ptB = take((nrec / F + 2) * 2 * sizeof(bucket_t)),
                     o_bucket = take(sizeof(bucket_t)), o_fin = take(sizeof(std_bucket_t));
        reserve(o);
        reserve_sums(1);
        const unsigned char* d_points = (const unsigned char*)points;
        const u32 *d_bm = bitmap, *d_rm = refmap;
        if (!pts_dev) { HIP_OK(hipMemcpyAsync(blob + o_pts, points, npoints * ffi_affine_sz, hipMemcpyHostToDevice, stream)); d_points = blob + o_pts; }
        if (!bm_dev) { HIP_OK(hipMemcpyAsync(blob + o_bm, bitmap, nwords * 4, hipMemcpyHostToDevice, stream)); d_bm = (const u32*)(blob + o_bm); }
        if (refmap && !rm_dev) { HIP_OK(hipMemcpyAsync(blob + o_rm, refmap, nwords * 4, hipMemcpyHostToDevice, stream)); d_rm = (const u32*)(blob + o_rm); }
        const bool flagged = ffi_affine_sz > 2 * FP_BYTES;
        if constexpr (INTERNAL) {
            launch_convert(blob + o_conv, d_points, (unsigned)npoints, ffi_affine_sz);
            d_points = blob + o_conv;
        }
        bucket_t* bucket = (bucket_t*)(blob + o_bucket);
        HIP_OK(hipMemsetAsync(bucket, 0, sizeof(bucket_t), stream));
        u32* ik = (u32*)(blob + o_keyA); bucket_t* ip = (bucket_t*)(blob + o_ptA);
        u32* ok = (u32*)(blob + o_keyB); bucket_t* op = (bucket_t*)(blob + o_ptB);
        {
            dim3 grid((unsigned)((nrec + 255) / 256));
            if (flagged) hipLaunchKernelGGL((k_bitmap_accumulate<fp_d, true>), grid, dim3(256), 0, stream, ik, ip, d_points,
                                            (unsigned)ffi_affine_sz, (unsigned)npoints, d_bm, d_rm, span);
            else         hipLaunchKernelGGL((k_bitmap_accumulate<fp_d, false>), grid, dim3(256), 0, stream, ik, ip, d_points,
                                            (unsigned)ffi_affine_sz, (unsigned)npoints, d_bm, d_rm, span);
            HIP_OK(hipGetLastError());
        }
        for (size_t n = nrec;;) {                                   // the MSM's record tree, every key = 0
            unsigned nthreads = (unsigned)((n + F - 1) / F);
            int last = nthreads == 1;
            hipLaunchKernelGGL(k_reduce_runs<fp_d>, dim3((nthreads + 255) / 256), dim3(256), 0, stream,
                               bucket, ok, op, ik, ip, (unsigned)n, (unsigned)F, nthreads, last, (const u32*)nullptr);
            HIP_OK(hipGetLastError());
            if (last) break;
            n = (size_t)2 * nthreads;
            std::swap(ik, ok); std::swap(ip, op);
        }
        if constexpr (INTERNAL) {
            std_bucket_t* fin = (std_bucket_t*)(blob + o_fin);
            hipLaunchKernelGGL((k_finalize<fp_d, STD_WORDS>), dim3(1), dim3(64), 0, stream, fin, bucket, 1u);
            HIP_OK(hipGetLastError());
            HIP_OK(hipMemcpyAsync(h_sums, fin, sizeof(std_bucket_t), hipMemcpyDeviceToHost, stream));
        } else {
            HIP_OK(hipMemcpyAsync(h_sums, bucket, sizeof(std_bucket_t), hipMemcpyDeviceToHost, stream));
        }
        HIP_OK(hipStreamSynchronize(stream));
        fp_h c[4];
        memcpy(c, h_sums, sizeof(c));
        out = point_t::from_xyzz(c[0], c[1], c[2], c[3]);
    }

    // out: Jacobian X|Y|Z (Montgomery).  points: stride ffi_affine_sz, flagged
    // format iff ffi_affine_sz > 2*FP_BYTES.  scalars: SCALAR_BYTES each.
    void invoke(point_t& out, const void* points, size_t npoints, const void* scalars,
                bool mont, size_t ffi_affine_sz)
    {
        out.set_inf();
        if (npoints == 0) return;
        if (npoints > (1u << 31)) HIP_OK(hipErrorInvalidValue);
        HIP_OK(hipSetDevice(gpu->hip_id));
        const bool preconverted = INTERNAL && points == nullptr;      // preload() already converted them
        if (points == nullptr) {                    // preloaded points, their own stride
            if (npoints > pre_n) HIP_OK(hipErrorInvalidValue);
            points = pre_points; ffi_affine_sz = pre_stride;
        }
        if (scalars == nullptr || ffi_affine_sz < 2 * FP_BYTES) HIP_OK(hipErrorInvalidValue);
        join_default_stream();
        // fixed-base tables cover exactly the preloaded vector: any other length runs the ordinary path on its first level
        if (preconverted && pre_fb_nwins && npoints == pre_n && tune.chunk == 0 && tune.max_scratch == 0 && fixed_fits(npoints, scalars)) {
            invoke_fixed(out, npoints, scalars, mont);
            return;
        }

        const bool pts_dev = is_device_pointer(points), sc_dev = is_device_pointer(scalars);
        const bool host = !pts_dev || !sc_dev;
        const size_t chunk = choose_chunk(npoints, (pts_dev ? 0 : ffi_affine_sz) + (sc_dev ? 0 : SCALAR_BYTES));
        const std::vector<size_t> cb = chunk_bounds(npoints, chunk);
        const size_t nchunks = cb.size() - 1;
        const size_t in_stride = preconverted ? conv_stride() : ffi_affine_sz;     // bytes between input records
        if (nchunks > 4096) HIP_OK(hipErrorOutOfMemory);

        // one plan (and layout) per chunk; scratch for the largest
        std::vector<msm_plan> plans(nchunks);
        std::vector<layout> layouts(nchunks);
        size_t need = 0, longest = 0;
        for (size_t c = 0; c < nchunks; c++) {
            plans[c] = make_plan(cb[c + 1] - cb[c], FRp::NBITS, tune);
            if (plans[c].nwins > MAX_WINS) HIP_OK(hipErrorInvalidValue);
            layouts[c] = make_layout(plans[c], !preconverted);
            need = std::max(need, layouts[c].total); longest = std::max(longest, cb[c + 1] - cb[c]);
        }
        reserve(need);
        reserve_sums(nchunks * MAX_WINS + 1);                    // (+ the small sizes' flag word behind the sums of a one-chunk MSM)
        const size_t st_pts = align_up(pts_dev ? 0 : longest * ffi_affine_sz), st_sc = align_up(sc_dev ? 0 : longest * SCALAR_BYTES);
        if (host) { reserve_stage(2 * (st_pts + st_sc)); if (nchunks > 1) need_cpy(); }

        for (size_t c = 0; c < nchunks; c++) {
            const size_t lo = cb[c], cn = cb[c + 1] - lo;
            const msm_plan& p = plans[c];
            const layout& l = layouts[c];
            const unsigned sb = c & 1;
            const unsigned char* d_points = (const unsigned char*)points + lo * in_stride;
            const u32* d_scalars = (const u32*)((const unsigned char*)scalars + lo * SCALAR_BYTES);
            if (host) {
                // copies of chunk c: scalars first (the sort only needs them).  With more than one chunk they run
                // on the copy stream, under the arithmetic of chunk c-1; pageable source memory makes the call
                // itself block until the data has left the host buffer, which is fine: chunk c-1's kernels are
                // already queued.
                hipStream_t cs = nchunks > 1 ? cpy : stream;
                unsigned char* sp = stage + sb * (st_pts + st_sc);
                if (nchunks > 1 && c >= 2) HIP_OK(hipStreamWaitEvent(cpy, ev_chunkdone[sb], 0));    // chunk c-2 is done with this set
                if (!sc_dev) {
                    HIP_OK(hipMemcpyAsync(sp + st_pts, d_scalars, cn * SCALAR_BYTES, hipMemcpyHostToDevice, cs));
                    d_scalars = (const u32*)(sp + st_pts);
                }
                if (!pts_dev) {
                    HIP_OK(hipMemcpyAsync(sp, d_points, cn * ffi_affine_sz, hipMemcpyHostToDevice, cs));
                    d_points = sp;
                }
                if (nchunks > 1) {
                    HIP_OK(hipEventRecord(ev_copied[sb], cpy));
                    HIP_OK(hipStreamWaitEvent(stream, ev_copied[sb], 0));
                }
            }
            // (chunks of different lengths have layouts of their own: harmless, every sort set of an MSM is
            // either written on the main stream or behind an event recorded on it after the previous MSM)
            enqueue(p, l, d_points, in_stride, preconverted, d_scalars, mont, h_sums + c * MAX_WINS, c == 0, 0, 0, false, nchunks == 1);
            if (host && nchunks > 1) HIP_OK(hipEventRecord(ev_chunkdone[sb], stream));
        }
        HIP_OK(hipStreamSynchronize(stream));
        last_chunks = (unsigned)nchunks;
        // the piece tree left a bucket with more pieces than it takes (skewed scalars): the fan-in tree over the records it
        // left, the bucket sums again (single chunk: the piece tree is for sizes far below a chunk)
        if (piece_pending && *h_flag_cur != 0) {
            enqueue(plans[0], layouts[0], nullptr, in_stride, preconverted, nullptr, mont, h_sums, false, 0, 0, true);
            HIP_OK(hipStreamSynchronize(stream));
            last_redo++;
        }
        piece_pending = false;

        if (timing) {
            const msm_plan& p = plans[0];
            float acc = 0, t;
            for (unsigned g = 0; g < p.G; g++) { HIP_OK(hipEventElapsedTime(&t, tev[2 + 2 * g], tev[3 + 2 * g])); acc += t; }
            HIP_OK(hipEventElapsedTime(&last_ms[0], tev[0], tev[2]));
            last_ms[1] = acc;
            HIP_OK(hipEventElapsedTime(&last_ms[2], tev[0], tev[1]));
            last_ms[3] = (float)p.G;
        }

        for (size_t c = 0; c < nchunks; c++) {
            point_t part = horner(h_sums + c * MAX_WINS, plans[c]);
            out.add(part);
        }
    }
};

} // namespace sppark_amd
