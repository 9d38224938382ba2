// Host driver of the MI355X MSM pipeline: plays the role of the reference's
// msm_t (msm/pippenger.cuh:325-728) -- window choice, one device blob, kernel
// sequence, error mapping -- re-planned for a 288 GB / 256-CU part:
//
//  * no batching: the whole point/scalar vector is resident (the reference
//    streams 2^24-point strides through a 3-stream pipeline, :454-557, to stay
//    inside a few GB);
//  * no cooperative launches or device-global work counters (:157-223): kernel
//    boundaries are the only grid-wide synchronisation;
//  * the host part is O(windows): it receives one XYZZ sum per window and runs
//    the Horner recombination (the reference integrates 256 partial sums per
//    window on a host thread pool, :627-727).
//
// Inputs may be host or device pointers (cf. is_device_ptr, util/gpu_t.cuh:385-395,
// and the preloaded-points constructor :351-388): device pointers are used in
// place, host pointers are staged with one H2D copy each.
#pragma once
#include "msm_kernels.hpp"
#include "../ec/jacobian_host.hpp"
#include "../util/runtime.hpp"
#include <algorithm>
#include <vector>

namespace sppark_amd {

struct msm_plan {
    unsigned n, wbits, nwins, NB;       // wbits = longest window, NB = 2^(wbits-1) buckets per window
    unsigned nbits;                     // scalar bits, split evenly over the windows
    unsigned HB, LB, NA;                // bucket index = (k_hi : k_lo), NA = 2^HB partitions per window
    unsigned L, chunks_per_win;         // accumulate run length
    unsigned nslabs, slab_sz;           // hist/scatter point slabs
    unsigned F;                         // reduce_runs fan-in
    unsigned K;                         // bucket-reduction chunk
};

struct msm_tunables {                   // 0 = automatic
    unsigned wbits = 0, L = 0, F = 0, K = 0, nslabs = 0, LB = 0;
    unsigned big = 0;                   // level-A partitions above this many entries are sorted cooperatively (0 = 2^18)
};

static inline unsigned lg2_floor(size_t x) { unsigned r = 0; while (x >>= 1) r++; return r; }

static inline msm_plan make_plan(size_t npoints, unsigned scalar_bits, const msm_tunables& t)
{
    msm_plan p;
    p.n = (unsigned)npoints;
    unsigned lg = lg2_floor(npoints ? npoints : 1);
    // window: ~2^6 entries per bucket on average; both halves of the bucket index
    // must fit LDS counters (2^15 u32 = 128 KB of the 160 KB) => wbits - 1 <= 30, capped at 24
    // measured optima (profiles/r01_msm_small_sweep.log): ~2^4 entries per bucket at
    // 2^20..2^26; below that the serial depth of the reduction levels dominates and
    // much smaller windows (more, shorter windows in parallel) win
    unsigned autow = lg >= 22 ? std::min(22u, lg - 4) : lg >= 20 ? 16u : lg == 19 ? 11u
                   : lg >= 16 ? 8u : std::max(4u, lg > 10 ? lg - 10 : 0u);
    p.wbits = t.wbits ? t.wbits : autow;
    p.wbits = std::min(24u, std::max(2u, p.wbits));
    p.nwins = (scalar_bits - 1) / p.wbits + 1;      // as pippenger.cuh:365
    p.nbits = scalar_bits;
    p.wbits = scalar_bits / p.nwins + (scalar_bits % p.nwins ? 1 : 0);     // even split (window_len)
    p.NB = 1u << (p.wbits - 1);
    // many small level-A partitions (<= 2^12 per window) keep level B's two passes
    // over a partition inside L2 (measured: profiles/r01_sort_split_sweep.log)
    p.LB = t.LB ? std::min(t.LB, p.wbits - 1) : (p.wbits - 1 > 12 ? p.wbits - 1 - 12 : 0);
    if (p.LB > 13) p.LB = 13;                       // 2^LB LDS counters + scan words
    if (p.wbits - 1 - p.LB > 15) p.LB = p.wbits - 1 - 15;
    p.HB = p.wbits - 1 - p.LB;
    p.NA = 1u << p.HB;
    size_t entries = (size_t)p.n * p.nwins;
    unsigned L = t.L ? t.L : (unsigned)std::min<size_t>(64, std::max<size_t>(4, entries / 262144));
    p.L = L;
    p.chunks_per_win = (p.n + L - 1) / L;
    p.nslabs = t.nslabs ? t.nslabs : (unsigned)std::min<size_t>(64, std::max<size_t>(1, npoints / 262144));
    p.slab_sz = (p.n + p.nslabs - 1) / p.nslabs;
    p.F = std::max(4u, t.F ? t.F : 8u);        // fan-in < 3 would never shrink the record list
    p.K = t.K ? t.K : (lg <= 22 ? 4 : 8);
    p.K = std::min(p.K, p.NB);
    return p;
}

// FD: device coordinate field (fp_class<P> for G1, fp2_dev<P> for G2), FH: its host
// twin with the same memory image (mont_host<P> / fp2_host<P>), FRp: scalar field.
template<class FD, class FH, class FRp>
class msm_t {
public:
    typedef FD fp_d;
    typedef mont_dev<FRp> fr_d;
    typedef FH fp_h;
    typedef typename xyzz_dev<fp_d>::mem_t bucket_t;       // memory image: wire format
    typedef jacobian_host<fp_h> point_t;
    static constexpr int STD_WORDS = sizeof(FH) / 4;        // 32-bit words of a coordinate in the reference's wire form
    static constexpr size_t FP_BYTES = sizeof(FH);
    static constexpr bool INTERNAL = field_is_internal<FD>::value;      // ff/montx_dev.hpp: own point/bucket records
    typedef xyzz_mem<STD_WORDS> std_bucket_t;
    static_assert(INTERNAL || sizeof(FH) == 4 * FD::N, "host and device coordinate fields must share the wire image");
    static constexpr size_t SCALAR_BYTES = sizeof(fr_d);

private:
    const gpu_info* gpu;
    hipStream_t stream;
    bool own_stream;
    hipEvent_t join_ev = nullptr;
    unsigned char* blob = nullptr;
    size_t blob_sz = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned char* pre_points = nullptr;    // points kept on the device by preload() (msm_t ctor with points, pippenger.cuh:351-385)
    size_t pre_n = 0, pre_stride = 0;
    float last_ms[3] = {0, 0, 0};       // [0] digits+sort, [1] accumulate, [2] whole device part
    bool timing = false;

    struct layout {
        size_t points, scalars, digits, sorted, partA, H, tot, offA, off, buckets;
        size_t keyA, ptA, keyB, ptB, A1, W1, A2, W2, conv, fin, curB, bigl, total;
    };

    static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
    static constexpr size_t conv_stride()
    {
        if constexpr (INTERNAL) return affine_loader<FD>::STRIDE; else return 0;
    }

    layout make_layout(const msm_plan& p, size_t pts_bytes, size_t sc_bytes, bool convert = true) const
    {
        layout l; size_t o = 0;
        auto take = [&](size_t sz) { size_t r = o; o += align_up(sz); return r; };
        l.points  = take(pts_bytes);
        l.scalars = take(sc_bytes);
        l.digits  = take((size_t)p.nwins * p.n * 4);
        l.sorted  = take((size_t)p.nwins * p.n * 4);
        l.partA   = take((size_t)p.nwins * p.n * 8);
        l.H       = take((size_t)p.nwins * p.nslabs * p.NA * 4);
        l.tot     = take((size_t)p.nwins * p.NA * 4);
        l.offA    = take((size_t)p.nwins * (p.NA + 1) * 4);
        l.off     = take((size_t)p.nwins * (p.NB + 1) * 4);
        l.buckets = take((size_t)p.nwins * p.NB * sizeof(bucket_t));
        size_t nrecA = (size_t)2 * p.nwins * p.chunks_per_win;
        size_t nrecB = 2 * ((nrecA + p.F - 1) / p.F);
        l.keyA = take(nrecA * 4); l.ptA = take(nrecA * sizeof(bucket_t));
        l.keyB = take(nrecB * 4); l.ptB = take(nrecB * sizeof(bucket_t));
        size_t n1 = (size_t)p.nwins * (p.NB / p.K);
        size_t n2 = n1;
        l.A1 = take(n1 * sizeof(bucket_t)); l.W1 = take(n1 * sizeof(bucket_t));
        l.A2 = take(n2 * sizeof(bucket_t)); l.W2 = take(n2 * sizeof(bucket_t));
        l.conv = take(INTERNAL && convert ? (size_t)p.n * conv_stride() : 0);     // points in the field's own records
        l.fin  = take(INTERNAL ? (size_t)p.nwins * sizeof(std_bucket_t) : 0);
        l.curB = take((size_t)p.nwins * (p.NB + 1) * 4);                    // cursors of the cooperative sort
        l.bigl = take(((size_t)p.nwins * p.NA + 1) * 4);                    // [count | list of oversized partitions]
        l.total = o;
        return l;
    }

    void reserve(size_t sz)
    {
        if (sz <= blob_sz) return;
        if (blob) { HIP_OK(hipFree(blob)); blob = nullptr; blob_sz = 0; }
        HIP_OK(hipMalloc((void**)&blob, sz));
        blob_sz = sz;
    }

public:
    msm_tunables tune;

    explicit msm_t(int device_id = -1, hipStream_t s = nullptr)
        : gpu(&select_gpu(device_id)), stream(s), own_stream(false)
    {
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
        (void)hipStreamSynchronize(stream);
        if (blob) (void)hipFree(blob);
        if (pre_points) (void)hipFree(pre_points);
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        if (own_stream) (void)hipStreamDestroy(stream);
        if (join_ev) (void)hipEventDestroy(join_ev);
    }
    msm_t(const msm_t&) = delete;
    msm_t& operator=(const msm_t&) = delete;

    void set_stream(hipStream_t s)
    {
        if (own_stream) { (void)hipStreamDestroy(stream); own_stream = false; }
        stream = s;
    }
    void enable_timing(bool on)
    {
        timing = on;
        if (on) for (auto& e : ev) if (!e) HIP_OK(hipEventCreate(&e));
    }
    float kernel_ms(int which) const { return which >= 0 && which < 3 ? last_ms[which] : -1.f; }
    size_t scratch_bytes() const { return blob_sz; }
    void release_scratch()
    {
        (void)hipStreamSynchronize(stream);
        if (blob) { (void)hipFree(blob); blob = nullptr; blob_sz = 0; }
    }
    msm_plan plan_for(size_t npoints) const { return make_plan(npoints, FRp::NBITS, tune); }

    // Size the blob for |npoints| ahead of time (so a timed invoke does not allocate).
    void reserve_for(size_t npoints, size_t ffi_affine_sz, bool host_points, bool host_scalars)
    {
        msm_plan p = make_plan(npoints, FRp::NBITS, tune);
        layout l = make_layout(p, host_points ? npoints * ffi_affine_sz : 0,
                                  host_scalars ? npoints * SCALAR_BYTES : 0);
        reserve(l.total);
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
    void preload(const void* points, size_t np, size_t ffi_affine_sz)
    {
        HIP_OK(hipSetDevice(gpu->hip_id));
        HIP_OK(hipStreamSynchronize(stream));
        join_default_stream();
        if (pre_points) { HIP_OK(hipFree(pre_points)); pre_points = nullptr; pre_n = pre_stride = 0; }
        if (np == 0) return;
        if (points == nullptr || ffi_affine_sz < 2 * FP_BYTES || np > (1u << 31)) HIP_OK(hipErrorInvalidValue);
        if constexpr (INTERNAL) {
            // bases are constant across invocations: convert them into the field's own records once
            const unsigned char* src = (const unsigned char*)points;
            unsigned char* staging = nullptr;
            if (!is_device_pointer(points)) {
                HIP_OK(hipMalloc((void**)&staging, np * ffi_affine_sz));
                HIP_OK(hipMemcpyAsync(staging, points, np * ffi_affine_sz, hipMemcpyHostToDevice, stream));
                src = staging;
            }
            hipError_t e = hipMalloc((void**)&pre_points, np * conv_stride());
            if (e == hipSuccess) {
                unsigned grid = (unsigned)((np + 255) / 256);
                if (ffi_affine_sz > 2 * FP_BYTES)
                    hipLaunchKernelGGL((k_convert_points<fp_d, true>), dim3(grid), dim3(256), 0, stream, pre_points, src, (unsigned)np, (unsigned)ffi_affine_sz);
                else
                    hipLaunchKernelGGL((k_convert_points<fp_d, false>), dim3(grid), dim3(256), 0, stream, pre_points, src, (unsigned)np, (unsigned)ffi_affine_sz);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipStreamSynchronize(stream);
            }
            if (staging) (void)hipFree(staging);
            if (e != hipSuccess) { if (pre_points) { (void)hipFree(pre_points); pre_points = nullptr; } HIP_OK(e); }
        } else {
            HIP_OK(hipMalloc((void**)&pre_points, np * ffi_affine_sz));
            HIP_OK(hipMemcpyAsync(pre_points, points, np * ffi_affine_sz,
                                  is_device_pointer(points) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
            HIP_OK(hipStreamSynchronize(stream));
        }
        pre_n = np; pre_stride = ffi_affine_sz;
    }
    size_t preloaded() const { return pre_n; }

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
        if (scalars == nullptr) HIP_OK(hipErrorInvalidValue);
        join_default_stream();

        const bool flagged = ffi_affine_sz > 2 * FP_BYTES;
        const bool pts_dev = is_device_pointer(points), sc_dev = is_device_pointer(scalars);
        const msm_plan p = make_plan(npoints, FRp::NBITS, tune);
        const layout l = make_layout(p, pts_dev ? 0 : npoints * ffi_affine_sz,
                                        sc_dev ? 0 : npoints * SCALAR_BYTES, !preconverted);
        reserve(l.total);

        const unsigned char* d_points = (const unsigned char*)points;
        const u32* d_scalars = (const u32*)scalars;
        if (!pts_dev) {
            HIP_OK(hipMemcpyAsync(blob + l.points, points, npoints * ffi_affine_sz, hipMemcpyHostToDevice, stream));
            d_points = blob + l.points;
        }
        if (!sc_dev) {
            HIP_OK(hipMemcpyAsync(blob + l.scalars, scalars, npoints * SCALAR_BYTES, hipMemcpyHostToDevice, stream));
            d_scalars = (const u32*)(blob + l.scalars);
        }

        u32* digits = (u32*)(blob + l.digits);
        u32* sorted = (u32*)(blob + l.sorted);
        u32* H = (u32*)(blob + l.H);
        u32* tot = (u32*)(blob + l.tot);
        u32* off = (u32*)(blob + l.off);
        bucket_t* buckets = (bucket_t*)(blob + l.buckets);

        if (timing) HIP_OK(hipEventRecord(ev[0], stream));

        // ---- digits + counting sort -------------------------------------
        {
            unsigned grid = std::min<unsigned>((p.n + 255) / 256, 256 * 16);
            hipLaunchKernelGGL(k_breakdown<fr_d>, dim3(grid), dim3(256), 0, stream,
                               digits, d_scalars, p.n, p.nwins, p.nbits, (int)mont);
            HIP_OK(hipGetLastError());
        }
        {
            uint2* partA = (uint2*)(blob + l.partA);
            u32* offA = (u32*)(blob + l.offA);
            size_t ldsA = (size_t)p.NA * 4, ldsB = ((size_t)1 << p.LB) * 4 + 4096;
            if (ldsA > 65536) {
                HIP_OK(hipFuncSetAttribute((const void*)k_histA, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA));
                HIP_OK(hipFuncSetAttribute((const void*)k_scatterA, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA));
            }
            if (ldsB > 65536)
                HIP_OK(hipFuncSetAttribute((const void*)k_sortB, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB));
            hipLaunchKernelGGL(k_histA, dim3(p.nslabs, p.nwins), dim3(1024), ldsA, stream,
                               H, digits, p.n, p.nslabs, p.slab_sz, p.NA, p.LB);
            HIP_OK(hipGetLastError());
            size_t na_total = (size_t)p.nwins * p.NA;
            hipLaunchKernelGGL(k_scan_slabs, dim3((unsigned)((na_total + 255) / 256)), dim3(256), 0, stream,
                               H, tot, p.nslabs, p.NA, p.nwins);
            HIP_OK(hipGetLastError());
            hipLaunchKernelGGL(k_scan_parts, dim3(p.nwins), dim3(1024), 0, stream, offA, tot, p.NA);
            HIP_OK(hipGetLastError());
            hipLaunchKernelGGL(k_scatterA, dim3(p.nslabs, p.nwins), dim3(1024), ldsA, stream,
                               partA, digits, H, offA, p.n, p.nslabs, p.slab_sz, p.NA, p.LB);
            HIP_OK(hipGetLastError());
            const unsigned big = tune.big ? tune.big : (1u << 18);
            u32* nbig = (u32*)(blob + l.bigl); u32* blist = nbig + 1; u32* curB = (u32*)(blob + l.curB);
            HIP_OK(hipMemsetAsync(nbig, 0, 4, stream));
            hipLaunchKernelGGL(k_sortB, dim3(p.NA, p.nwins), dim3(1024), ldsB, stream,
                               sorted, off, partA, offA, p.n, p.NA, p.LB, big);
            HIP_OK(hipGetLastError());
            // oversized partitions (skewed scalars); empty list and immediate return otherwise
            hipLaunchKernelGGL(k_big_find, dim3((p.NA * p.nwins + 255) / 256), dim3(256), 0, stream,
                               nbig, blist, off, offA, p.NA, p.LB, p.nwins, big);
            size_t ldsBig = ((size_t)1 << p.LB) * 4;
            hipLaunchKernelGGL(k_big_hist, dim3(1024), dim3(1024), ldsBig, stream, off, partA, offA, nbig, blist, p.n, p.NA, p.LB);
            hipLaunchKernelGGL(k_big_scan, dim3(64), dim3(1024), 0, stream, off, curB, offA, nbig, blist, p.NA, p.LB);
            hipLaunchKernelGGL(k_big_scatter, dim3(1024), dim3(1024), ldsBig, stream, sorted, curB, partA, offA, nbig, blist, p.n, p.NA, p.LB);
            HIP_OK(hipGetLastError());
        }
        HIP_OK(hipMemsetAsync(buckets, 0, (size_t)p.nwins * p.NB * sizeof(bucket_t), stream));

        if constexpr (INTERNAL) if (!preconverted) {   // wire points -> the field's own records (2 products per point)
            unsigned char* conv = blob + l.conv;
            unsigned grid = (unsigned)((p.n + 255) / 256);
            if (flagged) hipLaunchKernelGGL((k_convert_points<fp_d, true>), dim3(grid), dim3(256), 0, stream,
                                            conv, d_points, p.n, (unsigned)ffi_affine_sz);
            else         hipLaunchKernelGGL((k_convert_points<fp_d, false>), dim3(grid), dim3(256), 0, stream,
                                            conv, d_points, p.n, (unsigned)ffi_affine_sz);
            HIP_OK(hipGetLastError());
            d_points = conv;
        }

        if (timing) HIP_OK(hipEventRecord(ev[1], stream));

        // ---- bucket accumulation: level 0 + segmented tree ------------------
        u32* keyA = (u32*)(blob + l.keyA); bucket_t* ptA = (bucket_t*)(blob + l.ptA);
        u32* keyB = (u32*)(blob + l.keyB); bucket_t* ptB = (bucket_t*)(blob + l.ptB);
        {
            dim3 grid((p.chunks_per_win + 255) / 256, p.nwins);
            if (flagged)
                hipLaunchKernelGGL((k_accumulate<fp_d, true>), grid, dim3(256), 0, stream,
                                   buckets, keyA, ptA, d_points, (unsigned)ffi_affine_sz, sorted, off,
                                   p.n, p.NB, p.L, p.chunks_per_win);
            else
                hipLaunchKernelGGL((k_accumulate<fp_d, false>), grid, dim3(256), 0, stream,
                                   buckets, keyA, ptA, d_points, (unsigned)ffi_affine_sz, sorted, off,
                                   p.n, p.NB, p.L, p.chunks_per_win);
            HIP_OK(hipGetLastError());
        }
        if (timing) HIP_OK(hipEventRecord(ev[2], stream));
        {
            size_t nrec = (size_t)2 * p.nwins * p.chunks_per_win;
            u32* ik = keyA; bucket_t* ip = ptA; u32* ok = keyB; bucket_t* op = ptB;
            for (;;) {
                unsigned nthreads = (unsigned)((nrec + p.F - 1) / p.F);
                int last = nthreads == 1;
                hipLaunchKernelGGL(k_reduce_runs<fp_d>, dim3((nthreads + 255) / 256), dim3(256), 0, stream,
                                   buckets, ok, op, ik, ip, (unsigned)nrec, p.F, nthreads, last);
                HIP_OK(hipGetLastError());
                if (last) break;
                nrec = (size_t)2 * nthreads;
                std::swap(ik, ok); std::swap(ip, op);
            }
        }

        // ---- per-window weighted bucket sums --------------------------------
        bucket_t* A1 = (bucket_t*)(blob + l.A1); bucket_t* W1 = (bucket_t*)(blob + l.W1);
        bucket_t* A2 = (bucket_t*)(blob + l.A2); bucket_t* W2 = (bucket_t*)(blob + l.W2);
        bucket_t* result;
        {
            unsigned nitems = p.NB / p.K;
            size_t nthr = (size_t)p.nwins * nitems;
            hipLaunchKernelGGL(k_bucket_level1<fp_d>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream,
                               A1, W1, buckets, p.NB, p.K, p.nwins);
            HIP_OK(hipGetLastError());
            unsigned lgG = lg2_floor(p.K);
            bucket_t *ia = A1, *iw = W1, *oa = A2, *ow = W2;
            while (nitems > 1) {
                unsigned K = std::min(p.K, nitems);
                nthr = (size_t)p.nwins * (nitems / K);
                hipLaunchKernelGGL(k_bucket_levelN<fp_d>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream,
                                   oa, ow, ia, iw, nitems, K, lgG, p.nwins);
                HIP_OK(hipGetLastError());
                nitems /= K; lgG += lg2_floor(K);
                std::swap(ia, oa); std::swap(iw, ow);
            }
            result = iw;
        }
        if (timing) HIP_OK(hipEventRecord(ev[3], stream));

        // ---- device -> host: one XYZZ per window; Horner on the host --------
        std::vector<std_bucket_t> sums(p.nwins);
        if constexpr (INTERNAL) {           // window sums back to the reference's wire image
            std_bucket_t* fin = (std_bucket_t*)(blob + l.fin);
            hipLaunchKernelGGL((k_finalize<fp_d, STD_WORDS>), dim3((p.nwins + 63) / 64), dim3(64), 0, stream, fin, result, p.nwins);
            HIP_OK(hipGetLastError());
            HIP_OK(hipMemcpyAsync(sums.data(), fin, p.nwins * sizeof(std_bucket_t), hipMemcpyDeviceToHost, stream));
        } else {
            HIP_OK(hipMemcpyAsync(sums.data(), result, p.nwins * sizeof(std_bucket_t), hipMemcpyDeviceToHost, stream));
        }
        HIP_OK(hipStreamSynchronize(stream));

        if (timing) {
            HIP_OK(hipEventElapsedTime(&last_ms[0], ev[0], ev[1]));
            HIP_OK(hipEventElapsedTime(&last_ms[1], ev[1], ev[2]));
            HIP_OK(hipEventElapsedTime(&last_ms[2], ev[0], ev[3]));
        }

        for (unsigned w = p.nwins; w--;) {
            fp_h c[4];
            memcpy(c, &sums[w], sizeof(c));
            point_t s = point_t::from_xyzz(c[0], c[1], c[2], c[3]);
            out.add(s);
            if (w) for (unsigned k = 0; k < window_len(w - 1, p.nwins, p.nbits); k++) out.dbl();
        }
    }
};

} // namespace sppark_amd
