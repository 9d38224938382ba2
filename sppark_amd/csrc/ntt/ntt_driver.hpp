// Host driver of the NTT: the role of the reference's NTT class + NTTParameters
// (ntt/ntt.cuh:31-366, ntt/parameters.cuh:222-337): order/direction/type
// dispatch, per-(device, size, direction) twiddle tables built once on the
// device and kept for the life of the process, kernel sequence on one stream.
#pragma once
#include "ntt_kernels.hpp"
#include "ntt_r64_kernels.hpp"
#include "../ff/fr256_dev.hpp"
#include "../ff/mont_host.hpp"
#include "../util/runtime.hpp"
#include <map>
#include <mutex>
#include <tuple>

namespace sppark_amd {

enum { NTT_NN = 0, NTT_NR = 1, NTT_RN = 2, NTT_RR = 3 };       // ntt/ntt.cuh:33
enum { NTT_FORWARD = 0, NTT_INVERSE = 1 };                     // ntt/ntt.cuh:34
enum { NTT_STANDARD = 0, NTT_COSET = 1 };                      // ntt/ntt.cuh:35

// ---- host-side constants (parameter setup only; no transform code here) ------
template<class F> struct host_field;
template<> struct host_field<gl64_dev> {
    typedef unsigned __int128 u128;
    static u64 mul(u64 a, u64 b) { return (u64)(((u128)a * b) % gl64_dev::MOD); }
    static u64 pow(u64 b, u64 e) { u64 r = 1; while (e) { if (e & 1) r = mul(r, b); b = mul(b, b); e >>= 1; } return r; }
    static u64 inv(u64 a) { return pow(a, gl64_dev::MOD - 2); }
    static u64 top_root() { return gl64_dev::TOP_ROOT; }
    static u64 gen() { return gl64_dev::GROUP_GEN; }
    static u64 two_pow(unsigned lg) { return pow(2, lg); }
    static gl64_dev wire(u64 canonical) { gl64_dev r; r.v = canonical; return r; }
};
template<> struct host_field<bb31_dev> {                        // canonical arithmetic, converted at the end
    static constexpr u64 P = bb31_dev::MOD;
    static u64 mul(u64 a, u64 b) { return a * b % P; }
    static u64 pow(u64 b, u64 e) { u64 r = 1; while (e) { if (e & 1) r = mul(r, b); b = mul(b, b); e >>= 1; } return r; }
    static u64 inv(u64 a) { return pow(a, P - 2); }
    static u64 top_root() { return mul(bb31_dev::TOP_ROOT, inv((1ULL << 32) % P)); }   // out of Montgomery form
    static u64 gen() { return bb31_dev::GROUP_GEN; }
    static u64 two_pow(unsigned lg) { return pow(2, lg); }
    static bb31_dev wire(u64 canonical) { bb31_dev r; r.v = (u32)((canonical << 32) % P); return r; }
};

// 256-bit fields: everything stays in the Montgomery domain, which is the wire format
template<class P> struct host_field<fr256_dev<P>> {
    typedef mont_host<P> H;
    struct elem { H v; };
    static elem mul(const elem& a, const elem& b) { return elem{a.v * b.v}; }
    static elem one() { return elem{H::one()}; }
    static elem small(unsigned k) { elem r{H::zero()}; for (unsigned i = 0; i < k; i++) r.v = r.v + H::one(); return r; }
    static elem inv(const elem& a) { return elem{a.v.inverse()}; }
    static elem gen() { return small(P::GROUP_GEN); }
    static elem top_root()                          // gen^((r-1) >> S)
    {
        uint64_t e[P::N64];
        for (int i = 0; i < P::N64; i++) e[i] = P::MOD64[i];
        e[0] -= 1;                                  // r is odd
        elem r = one(), b = gen();
        for (unsigned bit = P::TWO_ADICITY; bit < 64 * P::N64; bit++) {
            if ((e[bit / 64] >> (bit % 64)) & 1) r = mul(r, b);
            // square AFTER use: b = gen^(2^(bit - S + 1))
            b = mul(b, b);
        }
        return r;
    }
    static elem two_pow(unsigned lg) { elem r = one(), two = small(2); for (unsigned i = 0; i < lg; i++) r = mul(r, two); return r; }
    static fr256_dev<P> wire(const elem& a) { fr256_dev<P> r; memcpy(r.v, a.v.v, sizeof(r.v)); return r; }
};

template<class F>
class ntt_engine {
    struct table_set { F *lo, *hi, *inner, *glo, *ghi; unsigned h; F scale; };
    // Inter-pass twiddle tables.  W = w_n^(n / n_cur) is the primitive n_cur-th root whatever the transform size n, so an
    // unscaled table is shared by every size and keyed (device, direction, family, a, b, 0): family 0 = k_pass_table
    // (a = lg_cur, b = S), family 1 = k_r64_table (a = kind, b = lg_cur).  A table that also carries 1/n belongs to one
    // size: its key ends in lg n.  (Round 3 kept one copy per (size, direction): a forward + inverse 2^24 transform of a
    // 256-bit field pinned > 1.1 GB, and more for every further size.)
    typedef std::tuple<int, int, int, unsigned, unsigned, unsigned> tw_key;
    std::map<tw_key, F*> tw_cache;
    // the radix-64 plan (ntt_r64_kernels.hpp): single-word fields, transforms of >= 2^12 elements
    static constexpr bool R64 = sizeof(F) <= 8;
    static constexpr unsigned R64_DIRECT_MAX_LG = 20;          // one inter-pass table up to 2^20 entries, two small ones above
    static constexpr unsigned PASS_TABLE_MAX_LG = 16;          // tables of <= 2^16 elements (512 KB for Goldilocks): L2-resident
    std::map<std::tuple<int, unsigned, int>, table_set> cache;     // (hip device, lg, inverse)
    std::mutex mtx;

    // columns per tile row (single-word fields: one 128-byte line; 256-bit fields: 16 elements =
    // 512 bytes, so that a 4-stage tile still has 64 register sub-transforms) / elements per LDS tile
    static constexpr unsigned LG_LINE = sizeof(F) == 4 ? 5 : sizeof(F) == 8 ? 4 : 4;
    static constexpr unsigned LG_TILE = sizeof(F) == 4 ? 13 : sizeof(F) == 8 ? 12 : 10;
    // stages per pass.  256-bit elements: radix-4 x radix-4 (S = 4).  Larger register radices keep
    // 8 or 16 eight-word elements per lane (152 VGPRs, scratch) and measured slower in spite of
    // fewer passes: 2^24 in 2.8 ms with S = 4, 3.3 ms with S = 6 (tools/gpu_ntt_wide_knobs.py).
    // Round 3, with the twiddle tables and the <3, 3> kernel freed of its scratch use (its eight-element write-out loop
    // is above clang's #pragma-unroll budget; -mllvm -pragma-unroll-threshold=200000: 193 VGPRs, no scratch): 2.55 ms
    // with S = 6 against 2.54 with S = 4 on the same box, S = 5 2.81 -- a third fewer instructions and two passes less
    // buy nothing at two waves per SIMD (LDS: 256 B per lane).  Not instantiated.  profiles/r03_ntt_wide_stages.log
    static constexpr unsigned S_MAX = sizeof(F) > 8 ? 4 : 8;
    // Round 4: the 256-bit fields run passes of up to 8 stages with ONE butterfly per lane and stage and the tile in LDS
    // throughout (k_ntt_pass_lat, ntt_kernels.hpp): S/2 + 1 products per element for S stages instead of 3.75 for 4, half
    // the passes -- and half the dependent chain on the small, latency-bound sizes.  0 = the register passes above.
    static constexpr unsigned LAT_SMAX = sizeof(F) > 8 ? 8 : 0;

    // 256-bit elements: tables up to 2^24 entries (512 MB; every sub-problem of the pass reads its table once, the small
    // ones from L2 / the Infinity Cache).  The per-element alternative is a lo x hi product per twiddle: one of the ~3.5
    // products per element and pass.  BLS12-381 Fr 2^24 forward (profiles/r03_ntt_wide_tables.log): 2.80 ms without
    // tables, 2.43 with tables up to 2^16, 2.33 up to 2^20, 2.25 up to 2^24 -- even the table as large as the data pays,
    // the passes are bound by their products.  (Tuning builds, -DSPPARK_TUNING: SPPARK_NTT_WIDE_TABLE=<log2> moves the
    // limit, 0 = no tables.)
    static unsigned wide_table_lg()
    {
#ifdef SPPARK_TUNING
        static const unsigned v = [] { const char* e = getenv("SPPARK_NTT_WIDE_TABLE"); return e ? (unsigned)atoi(e) : 24u; }();
        return v;
#else
        return 24u;
#endif
    }
    // transforms up to this size run as one launch of one work-group: 2^11 for the single-word fields, 2^9 for the 256-bit
    // ones (at 2^10 their two one-stage-per-round launches are as fast as eight waves of 256-bit exchanges through
    // LDS: 17.7 against 17.8 us, profiles/r05_ntt_small_sized_ab.log).  Tuning builds: SPPARK_NTT_SMALL_MAX, 0 = never.
    static unsigned small_max_lg()
    {
        static constexpr unsigned cap = ntt_small_cap<F>::value, dflt = sizeof(F) > 8 ? cap - 1 : cap;
#ifdef SPPARK_TUNING
        static const unsigned v = [] { const char* e = getenv("SPPARK_NTT_SMALL_MAX"); return e ? std::min((unsigned)atoi(e), cap) : dflt; }();
        return v;
#else
        return dflt;
#endif
    }
    // tuning builds: SPPARK_NTT_SMALL_SIZED=0 runs every size through the run-time-size instance
    static bool small_sized()
    {
#ifdef SPPARK_TUNING
        static const bool v = [] { const char* e = getenv("SPPARK_NTT_SMALL_SIZED"); return !e || atoi(e) != 0; }();
        return v;
#else
        return true;
#endif
    }
    // the inter-pass twiddle table of a pass on sub-problems of 2^lg_cur elements (built once per
    // (device, size, direction, pass shape); null when the pass generates its twiddles instead)
    static bool has_pass_table(unsigned lg_cur, unsigned S)
    {   return !(lg_cur > (ntt_gen_twiddles<F>::value ? PASS_TABLE_MAX_LG : wide_table_lg()) || lg_cur <= S || S / 2 == 0);   }
    // Find or build one table.  The engine lock is NOT held while the table is allocated, generated and synchronised
    // (another thread's transform of a cached size goes on meanwhile); two threads that miss the same key both build
    // it and the loser frees its copy.  Returns nullptr when the device has no room for it -- the idle scratch buffers
    // are given back first -- and the caller falls back to a plan without that table.
    template<class Launch>
    const F* cached_table(const tw_key& key, size_t count, hipStream_t stream, Launch&& launch)
    {
        {
            std::lock_guard<std::mutex> lk(mtx);
            auto it = tw_cache.find(key);
            if (it != tw_cache.end()) return it->second;
        }
        F* tw = nullptr;
        hipError_t e = dev_scratch_pool::malloc_or_drain((void**)&tw, count * sizeof(F));
        if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return nullptr; }
        HIP_OK(e);
        launch(tw);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(stream);      // visible to every later call on any stream
        if (e != hipSuccess) { (void)hipFree(tw); HIP_OK(e); }
        std::lock_guard<std::mutex> lk(mtx);
        auto ins = tw_cache.emplace(key, tw);
        if (!ins.second) (void)hipFree(tw);                         // built twice: keep the first
        return ins.first->second;
    }
    const F* pass_table(int hip_dev, unsigned lg, int inverse, unsigned lg_cur, unsigned S, int scaled, const ntt_tables<F>& T, hipStream_t stream)
    {
        if (!has_pass_table(lg_cur, S)) return nullptr;
        const size_t n = (size_t)1 << lg_cur;
        return cached_table(tw_key(hip_dev, inverse, 0, lg_cur, S, scaled ? lg : 0u), n, stream, [&](F* tw) {
            hipLaunchKernelGGL(k_pass_table<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tw, T, lg_cur, S, scaled);
        });
    }

    // a table of the radix-64 plan (r64_table_item); |cmode|: with the coset powers folded in (G: the powers of g), which
    // ties it to the transform size
    const F* r64_table(int hip_dev, unsigned lg, int inverse, unsigned kind, unsigned lg_cur, int scaled,
                       const ntt_tables<F>& T, hipStream_t stream, unsigned cmode, const ntt_tables<F>& G)
    {
        if (cmode == 2 && kind == 2) cmode = 0;                     // (no row-only part: the plain table)
        const size_t count = kind == 0 ? (size_t)1 << lg_cur : kind == 1 ? (size_t)1 << (lg_cur - 6) : 4096;
        const tw_key key = cmode ? tw_key(hip_dev, inverse, 1 + 10 * (int)cmode, kind, lg_cur, lg | (scaled ? 256u : 0u))
                                 : tw_key(hip_dev, inverse, 1, kind, lg_cur, scaled ? lg : 0u);
        return cached_table(key, count, stream, [&](F* tw) {
            hipLaunchKernelGGL(k_r64_table<F>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, tw, T, kind, lg_cur, scaled, cmode, G);
        });
    }
    // the 64 constants of a folded coset transform (r64_cz_item)
    const F* r64_cz(int hip_dev, unsigned lg, int inverse, unsigned cmode, const ntt_tables<F>& G, hipStream_t stream)
    {
        return cached_table(tw_key(hip_dev, inverse, 2, cmode, 0u, lg), 64, stream, [&](F* cz) {
            hipLaunchKernelGGL(k_r64_cz<F>, dim3(1), dim3(64), 0, stream, cz, G, cmode);
        });
    }

    table_set tables(int hip_dev, unsigned lg, int inverse, hipStream_t stream)
    {
        std::lock_guard<std::mutex> lk(mtx);
        auto key = std::make_tuple(hip_dev, lg, inverse);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
        typedef host_field<F> H;
        table_set t;
        t.h = lg < 12 ? lg : 12;
        size_t nlo = (size_t)1 << t.h, nhi = (size_t)1 << (lg - t.h);
        HIP_OK(dev_scratch_pool::malloc_or_drain((void**)&t.lo, (2 * (nlo + nhi) + ntt_inner_entries<F>::value) * sizeof(F)));
        t.hi = t.lo + nlo; t.glo = t.hi + nhi; t.ghi = t.glo + nlo; t.inner = t.ghi + nhi;
        auto w = H::top_root();
        for (unsigned k = F::TWO_ADICITY; k > lg; k--) w = H::mul(w, w);
        auto g = H::gen();
        if (inverse) { w = H::inv(w); g = H::inv(g); }
        t.scale = H::wire(H::inv(H::two_pow(lg)));
        unsigned grid = (unsigned)((std::max<size_t>(std::max(nlo, nhi), ntt_inner_entries<F>::value) + 255) / 256);
        hipLaunchKernelGGL(k_tables<F>, dim3(grid), dim3(256), 0, stream, t.lo, t.hi, t.inner, H::wire(w), lg, t.h);
        hipLaunchKernelGGL(k_tables<F>, dim3(grid), dim3(256), 0, stream, t.glo, t.ghi, (F*)nullptr, H::wire(g), lg, t.h);
        HIP_OK(hipGetLastError());
        // one-time: the tables are shared by every later call on ANY stream, so they must be
        // complete before another thread can find them in the cache
        HIP_OK(hipStreamSynchronize(stream));
        return cache.emplace(key, t).first->second;
    }

public:
    static ntt_engine& instance() { static ntt_engine e; return e; }

    // Free every cached table of this library (sppark_ntt_release_cached); they are rebuilt on demand.  The caller
    // guarantees that no transform of this library is in flight.
    void release_tables()
    {
        std::lock_guard<std::mutex> lk(mtx);
        int cur = 0; (void)hipGetDevice(&cur);
        for (auto& kv : tw_cache) { (void)hipSetDevice(std::get<0>(kv.first)); (void)hipFree(kv.second); }
        for (auto& kv : cache) { (void)hipSetDevice(std::get<0>(kv.first)); (void)hipFree(kv.second.lo); }
        tw_cache.clear(); cache.clear();
        (void)hipSetDevice(cur);
    }
    size_t cached_table_count() { std::lock_guard<std::mutex> lk(mtx); return tw_cache.size() + cache.size(); }

    // in-place transform of a DEVICE buffer of 2^lg elements on |stream|
    // |lde| (sppark_lde's forward RN transform only): the input is still the COMPACT coefficients lde->src (2^lg_domain, bit-reversed
    // order, unshifted); the first step of the radix-64 plan reads them as the spread array (k_ntt12<.., LDE>), and where the
    // transform runs another plan the spread is launched here first.
    // |lde->out| instead (sppark_lde's inverse NR transform): the result goes to |out|, |d| is scratch afterwards -- the last step
    // of the radix-64 plan stores there (ntt_r64_args::out), any other plan runs in place and copies.
    struct lde_input { const F* src; unsigned lg_domain, lg_blowup; F* out; };
    void run(const gpu_info& gpu, F* d, unsigned lg, int order, int direction, int type, hipStream_t stream, const lde_input* lde = nullptr)
    {
        if (lg == 0) {                                              // ntt/ntt.cuh:220-221 (one element: the hand-over of sppark_lde still happens)
            if (lde && lde->out) HIP_OK(hipMemcpyAsync(lde->out, d, sizeof(F), hipMemcpyDeviceToDevice, stream));
            else if (lde) lde_spread(gpu, d, lde->src, lde->lg_domain, lde->lg_blowup, true, stream);
            return;
        }
        if (lg > F::TWO_ADICITY || order < 0 || order > 3) HIP_OK(hipErrorInvalidValue);
        const int inverse = direction == NTT_INVERSE;
        const table_set ts = tables(gpu.hip_id, lg, inverse, stream);
        ntt_tables<F> T{ts.lo, ts.hi, ts.inner, lg, ts.h, ts.scale, nullptr}, G{ts.glo, ts.ghi, nullptr, lg, ts.h, ts.scale, nullptr};
        const size_t n = (size_t)1 << lg;
        const unsigned egrid = (unsigned)((n + 255) / 256);

        // up to 2^11 elements (256-bit fields: 2^9): the whole transform -- permutations, coset powers and 1/n included -- by one work-group
        // in one launch (k_ntt_small, ntt_kernels.hpp)
        F* final_out = lde ? lde->out : nullptr;
        if (final_out) lde = nullptr;
        if (lde && lg <= small_max_lg()) { lde_spread(gpu, d, lde->src, lde->lg_domain, lde->lg_blowup, true, stream); lde = nullptr; }
        if (lg <= small_max_lg()) {
            const unsigned flags = ntt_small_flags(order, inverse != 0, type == NTT_COSET);
            const unsigned lanes = (unsigned)std::max<size_t>(64, n / 2);
            const size_t lds = lanes > 64 ? 2 * (size_t)lanes * sizeof(F) : 0;        // the exchanges across waves (ntt_rx_regroup)
            // (single-word fields: the sizes 2^8 ... 2^11 have their own instance, compiled for that size)
#define SPPARK_NTT_SMALL_PICK(LGC) do { \
                if (inverse) hipLaunchKernelGGL((k_ntt_small<F, true, LGC>), dim3(1), dim3(lanes), lds, stream, d, T, G, flags); \
                else         hipLaunchKernelGGL((k_ntt_small<F, false, LGC>), dim3(1), dim3(lanes), lds, stream, d, T, G, flags); } while (0)
            if constexpr (sizeof(F) <= 8) {
                switch (small_sized() ? lg : 0u) {
                    case 8:  SPPARK_NTT_SMALL_PICK(8); break;
                    case 9:  SPPARK_NTT_SMALL_PICK(9); break;
                    case 10: SPPARK_NTT_SMALL_PICK(10); break;
                    case 11: SPPARK_NTT_SMALL_PICK(11); break;
                    default: SPPARK_NTT_SMALL_PICK(0); break;
                }
            } else
                SPPARK_NTT_SMALL_PICK(0);
#undef SPPARK_NTT_SMALL_PICK
            HIP_OK(hipGetLastError());
            if (final_out) HIP_OK(hipMemcpyAsync(final_out, d, n * sizeof(F), hipMemcpyDeviceToDevice, stream));
            return;
        }

        bool bitrev, gs;
        switch (order) {
            case NTT_NN: bit_reverse(d, lg, stream);
                         bitrev = true;  gs = false; break;
            case NTT_NR: bitrev = false; gs = true;  break;
            case NTT_RN: bitrev = true;  gs = false; break;
            default:     bitrev = true;  gs = true;  break;
        }
        // The plan's shape parameters.  A shipped library uses the constants; a tuning build (-DSPPARK_TUNING, e.g.
        // SPPARK_EXTRA_FLAGS=-DSPPARK_TUNING python -m sppark_amd.build; tools/gpu_ntt_sweep.py) reads them from the
        // environment once per process.  The LDS tile is clamped to what the element size allows (160 KB per
        // work-group: 2^14 eight-byte elements, 2^12 32-byte ones; the one-stage-per-round passes stay within the
        // 64 KB a launch gets without raising the kernel's limit).
        struct knobs_t { unsigned smax, lgc, lgt, r64_min, r64_direct, lat_smax; int lat_lgc, lat_lgt; unsigned coset_fold; };
        static const knobs_t knobs = [] {
            knobs_t k{S_MAX, LG_LINE, LG_TILE, 12, R64_DIRECT_MAX_LG, LAT_SMAX, -1, -1, 1};
#ifdef SPPARK_TUNING
            // 256-bit fields: stages per one-stage-per-round pass (0: the register passes), columns per tile row and tile
            // elements (log2; default: by size, lat_shape())
            if (const char* e = getenv("SPPARK_NTT_LAT_SMAX")) { unsigned v = (unsigned)atoi(e); if (v <= 8) k.lat_smax = R64 ? 0 : v; }
            if (const char* e = getenv("SPPARK_NTT_LAT_LGC")) { int v = atoi(e); if (v >= 0 && v <= 3) k.lat_lgc = v; }
            if (const char* e = getenv("SPPARK_NTT_LAT_LGTILE")) { int v = atoi(e); if (v >= 6 && v <= 11) k.lat_lgt = v; }
            if (const char* e = getenv("SPPARK_NTT_COSET_FOLD")) k.coset_fold = (unsigned)atoi(e);    // 0: the separate scaling launch
            if (const char* e = getenv("SPPARK_NTT_R64_MIN")) k.r64_min = (unsigned)atoi(e);          // 99: the 8-stage plan only
            if (const char* e = getenv("SPPARK_NTT_R64_DIRECT")) k.r64_direct = (unsigned)atoi(e);    // largest single inter-pass table (log2)
            if (const char* e = getenv("SPPARK_NTT_SMAX")) { unsigned v = (unsigned)atoi(e); if (v >= 1 && v <= S_MAX) k.smax = v; }
            if (const char* e = getenv("SPPARK_NTT_LGC")) { unsigned v = (unsigned)atoi(e); if (v >= 1 && v <= 8) k.lgc = v; }
            if (const char* e = getenv("SPPARK_NTT_LGTILE")) {
                unsigned v = (unsigned)atoi(e), cap = 17;
                while (((size_t)sizeof(F) << cap) > 160 * 1024) cap--;
                if (v >= 8 && v <= cap) k.lgt = v;
            }
#endif
            if (k.lgt < k.smax + 1) k.lgt = k.smax + 1;
            return k;
        }();
        const unsigned smax = knobs.smax, lgc = knobs.lgc, lgt = knobs.lgt;
        ntt_plan pl;
        r64_plan rp; rp.nsteps = 0;
        // the tables of the radix-64 plan, fetched (first call: built) BEFORE anything is launched: if the device has no
        // room for one of them the transform runs the 8-stage plan, whose passes can generate their twiddles
        const F* r64_tabs[8][2] = {};
        // a coset transform on a plan of k_ntt6 / k_ntt12 steps carries the powers of the coset generator in its tables and 64
        // constants (r64_coset_mode, ntt_r64_kernels.hpp): no separate scaling launch (2^24: 0.217 -> 0.19 ms)
        unsigned cmode = 0;
        const F* r64_czp = nullptr;
        const F* top_crow = nullptr;
        if (R64 && lg >= 12 && lg >= knobs.r64_min) {
            rp = make_r64_plan(lg);
            cmode = knobs.coset_fold ? r64_coset_mode(rp, gs, inverse != 0, type == NTT_COSET && order != NTT_RR) : 0;
            bool ok = true;
            for (unsigned i = 0; i < rp.nsteps && ok; i++) {
                const r64_step st = rp.step[gs ? i : rp.nsteps - 1 - i];
                if (st.kind == 0) continue;
                const int scaled = inverse && i == rp.nsteps - 1;
                const unsigned cm = cmode == 1 && st.lg_cur != lg ? 0 : cmode;     // natural exponents: the step on the whole transform only
                if (st.kind == 2 || st.lg_cur <= knobs.r64_direct)
                    ok = (r64_tabs[i][0] = r64_table(gpu.hip_id, lg, inverse, 0, st.lg_cur, scaled, T, stream, cm, G)) != nullptr;
                else
                    ok = (r64_tabs[i][0] = r64_table(gpu.hip_id, lg, inverse, 1, st.lg_cur, scaled, T, stream, cm, G)) != nullptr
                      && (r64_tabs[i][1] = r64_table(gpu.hip_id, lg, inverse, 2, st.lg_cur, 0, T, stream, cm, G)) != nullptr;
            }
            if (ok && cmode) ok = (r64_czp = r64_cz(gpu.hip_id, lg, inverse, cmode, G, stream)) != nullptr;
            // (a generic pass on top of the plan takes its share as 2^S row constants: g^(row << lgQ) / g^(rev_S(row)))
            if (ok && cmode && rp.step[0].kind == 0) {
                const unsigned S = rp.step[0].S;
                ok = (top_crow = cached_table(tw_key(gpu.hip_id, inverse, 3, cmode, S, lg), (size_t)1 << S, stream, [&](F* t) {
                          hipLaunchKernelGGL(k_pass_crow<F>, dim3(1), dim3(256), 0, stream, t, G, cmode, S);
                      })) != nullptr;
            }
            if (!ok) { rp.nsteps = 0; cmode = 0; top_crow = nullptr; }
        }
        // sppark_lde: the spread + coset shift inside the first step where that is k_ntt12 on a forward RN transform with at
        // most 8 positions per coefficient; otherwise as a launch of its own, now
        bool lde_fused = false;
        if (lde) {
            if constexpr (R64)
                lde_fused = rp.nsteps && !gs && !inverse && order == NTT_RN && type == NTT_STANDARD && rp.step[rp.nsteps - 1].kind == 2
                         && lde->lg_blowup >= 1 && lde->lg_blowup <= 3 && lde->lg_domain + lde->lg_blowup == lg;
            if (!lde_fused) lde_spread(gpu, d, lde->src, lde->lg_domain, lde->lg_blowup, true, stream);
        }
        if (!inverse && type == NTT_COSET && !cmode)
            hipLaunchKernelGGL(k_coset<F>, dim3(egrid), dim3(256), 0, stream, d, G, (int)bitrev);
        const bool lat = !R64 && knobs.lat_smax != 0;
        if (rp.nsteps) pl.npass = rp.nsteps;
        else if (lat)  pl = make_ntt_lat_plan(lg, knobs.lat_smax, knobs.lat_lgc, knobs.lat_lgt);
        else           pl = make_ntt_plan(lg, lgc, lgt, smax);
        int scale_pass = -1;                        // (the radix-64 plan folds the scaling into its own last table)
        if (inverse && !rp.nsteps) {
            unsigned best = ~0u;
            for (unsigned i = 0; i < pl.npass; i++) {
                const ntt_pass& q = pl.pass[gs ? i : pl.npass - 1 - i];
                if (has_pass_table(q.lg_cur, q.S) && q.lg_cur < best) { best = q.lg_cur; scale_pass = (int)i; }
            }
        }
        for (unsigned i = 0; i < pl.npass; i++) {
            const bool last = i == pl.npass - 1;
            ntt_pass P;
            P.cmode = 0; P.crow = P.cg_lo = P.cg_hi = nullptr; P.cg_h = 0;
            if (rp.nsteps) {
                const r64_step st = rp.step[gs ? i : rp.nsteps - 1 - i];
                if constexpr (R64) {
                    if (st.kind != 0) {
                        ntt_r64_args<F> A{T.inner, nullptr, nullptr, nullptr, st.lg_cur, nullptr};
                        if (cmode == 2 ? st.kind == 2 : (cmode == 1 && st.lg_cur == lg)) A.cz = r64_czp;
                        if (st.kind == 2 || st.lg_cur <= knobs.r64_direct) A.tw = r64_tabs[i][0];
                        else { A.t1 = r64_tabs[i][0]; A.t2 = r64_tabs[i][1]; }
                        const unsigned tiles = (unsigned)(n >> 12);
                        const size_t lds = sizeof(F) << 12;
#define SPPARK_R64_LAUNCH(K)                                                                                   \
                        do {                                                                                   \
                            if (gs) { if (inverse) hipLaunchKernelGGL((K<F, true, true>), dim3(tiles), dim3(512), lds, stream, d, A);    \
                                      else         hipLaunchKernelGGL((K<F, true, false>), dim3(tiles), dim3(512), lds, stream, d, A); } \
                            else    { if (inverse) hipLaunchKernelGGL((K<F, false, true>), dim3(tiles), dim3(512), lds, stream, d, A);   \
                                      else         hipLaunchKernelGGL((K<F, false, false>), dim3(tiles), dim3(512), lds, stream, d, A); } \
                        } while (0)
                        if (final_out && last && st.kind == 2 && gs) { A.out = final_out; final_out = nullptr; }
                        if (lde_fused && i == 0) {
                            const table_set tg = tables(gpu.hip_id, lde->lg_domain, 0, stream);
                            A.lde_src = lde->src; A.lde_glo = tg.glo; A.lde_ghi = tg.ghi; A.lde_gh = tg.h;
                            A.lde_lgd = lde->lg_domain; A.lde_lgb = lde->lg_blowup;
                            hipLaunchKernelGGL((k_ntt12<F, false, false, true>), dim3(tiles), dim3(512), lds, stream, d, A);
                        } else if (st.kind == 1) SPPARK_R64_LAUNCH(k_ntt6); else SPPARK_R64_LAUNCH(k_ntt12);
#undef SPPARK_R64_LAUNCH
                        continue;
                    }
                }
                P.lg_cur = st.lg_cur; P.S = st.S; P.lgC = lgc; P.lgG = 0;       // a strided pass above >= 12 further stages
                P.cmode = cmode; P.crow = top_crow; P.cg_lo = G.lo; P.cg_hi = G.hi; P.cg_h = G.h;      // (a folded coset transform)
            } else {
                P = pl.pass[gs ? i : pl.npass - 1 - i];
            }
            // 1/n of an inverse transform: carried by the table of the tabled pass on the smallest sub-problems when the
            // plan has one (a product per element saved), applied by the last pass otherwise
            const bool scale_here = inverse && scale_pass == (int)i;
            T.pass_tw = pass_table(gpu.hip_id, lg, inverse, P.lg_cur, P.S, scale_here ? 1 : 0, T, stream);
            if (scale_here && T.pass_tw == nullptr) {           // no room for the scaled table: the shared unscaled one (or
                scale_pass = -1;                                // none), and the last pass multiplies by 1/n itself
                T.pass_tw = pass_table(gpu.hip_id, lg, inverse, P.lg_cur, P.S, 0, T, stream);
            }
            P.apply_scale = inverse && last && scale_pass < 0;
            size_t tile_elems = (size_t)1 << (P.lgG + P.S + P.lgC);
            unsigned tiles = (unsigned)(n / tile_elems);
            if constexpr (!R64) {
                if (lat) {                                      // one butterfly per lane and stage, the tile in LDS throughout
                    const unsigned lanes = (unsigned)std::min<size_t>(std::max<size_t>(tile_elems / 2, 64), 1024);
                    const size_t lat_lds = tile_elems * sizeof(F);
                    if (lat_lds > 64 * 1024) HIP_OK(hipErrorInvalidValue);      // (only a tuning build can ask for such a tile)
                    if (gs) { if (inverse) hipLaunchKernelGGL((k_ntt_pass_lat<F, true, true>), dim3(tiles), dim3(lanes), lat_lds, stream, d, T, P);
                              else         hipLaunchKernelGGL((k_ntt_pass_lat<F, true, false>), dim3(tiles), dim3(lanes), lat_lds, stream, d, T, P); }
                    else    { if (inverse) hipLaunchKernelGGL((k_ntt_pass_lat<F, false, true>), dim3(tiles), dim3(lanes), lat_lds, stream, d, T, P);
                              else         hipLaunchKernelGGL((k_ntt_pass_lat<F, false, false>), dim3(tiles), dim3(lanes), lat_lds, stream, d, T, P); }
                    continue;
                }
            }
            size_t lds = ntt_lds_elems(P) * sizeof(F);
            // one lane per register sub-transform: a tile has 2^(lgG + R1 + lgC) of them in its
            // larger round (small tiles of wide elements would leave most of 256 lanes idle)
            const unsigned groups = 1u << (P.lgG + (P.S + 1) / 2 + P.lgC);
            const unsigned nthr = groups >= 512 ? 512 : groups <= 64 ? 64 : groups;
#define SPPARK_NTT_LAUNCH(R1, R2)                                                                              \
            do {                                                                                               \
                if (lds > 65536) {                                                                             \
                    const void* fn = gs ? (inverse ? (const void*)k_ntt_pass<F, true, true, R1, R2> : (const void*)k_ntt_pass<F, true, false, R1, R2>)   \
                                        : (inverse ? (const void*)k_ntt_pass<F, false, true, R1, R2> : (const void*)k_ntt_pass<F, false, false, R1, R2>); \
                    HIP_OK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
                }                                                                                              \
                if (gs) { if (inverse) hipLaunchKernelGGL((k_ntt_pass<F, true, true, R1, R2>), dim3(tiles), dim3(nthr), lds, stream, d, T, P);   \
                          else         hipLaunchKernelGGL((k_ntt_pass<F, true, false, R1, R2>), dim3(tiles), dim3(nthr), lds, stream, d, T, P); } \
                else    { if (inverse) hipLaunchKernelGGL((k_ntt_pass<F, false, true, R1, R2>), dim3(tiles), dim3(nthr), lds, stream, d, T, P);  \
                          else         hipLaunchKernelGGL((k_ntt_pass<F, false, false, R1, R2>), dim3(tiles), dim3(nthr), lds, stream, d, T, P); } \
            } while (0)
            if constexpr (S_MAX >= 8) { SPPARK_NTT_DISPATCH_S(P.S, SPPARK_NTT_LAUNCH); }
            else                      { SPPARK_NTT_DISPATCH_S4(P.S, SPPARK_NTT_LAUNCH); }
#undef SPPARK_NTT_LAUNCH
        }
        if (inverse && type == NTT_COSET && !cmode)
            hipLaunchKernelGGL(k_coset<F>, dim3(egrid), dim3(256), 0, stream, d, G, (int)!bitrev);
        if (order == NTT_RR)
            bit_reverse(d, lg, stream);
        HIP_OK(hipGetLastError());
        if (final_out) HIP_OK(hipMemcpyAsync(final_out, d, n * sizeof(F), hipMemcpyDeviceToDevice, stream));      // (no step stored there)
    }

    // in-place bit-reversal permutation (NN and RR orders; ntt/ntt.cuh:44-79)
    static void bit_reverse(F* d, unsigned lg, hipStream_t stream)
    {
        constexpr unsigned TB = bitrev_tile_bits<F>::value;
        const size_t n = (size_t)1 << lg;
        if (lg >= 2 * TB + 1) {
            size_t lds = 2 * (((size_t)(1u << TB) + 1) << TB) * sizeof(F);
            // 16-byte accesses for the single-word fields (any 16-byte aligned buffer; a view at an odd element offset
            // takes the element-wise tiles)
            bool vec = false;
            if constexpr (sizeof(F) <= 8) vec = ((uintptr_t)d & 15) == 0;
            if constexpr (sizeof(F) <= 8) {
                if (vec) hipLaunchKernelGGL((k_bitrev_tiled_vec<F, TB>), dim3((unsigned)(n >> (2 * TB))), dim3(256), lds, stream, d, lg);
            }
            if (!vec) hipLaunchKernelGGL((k_bitrev_tiled<F, TB>), dim3((unsigned)(n >> (2 * TB))), dim3(256), lds, stream, d, lg);
        } else {
            hipLaunchKernelGGL(k_bitrev<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d, lg);
        }
    }

    // d_inout[idx] *= g^(rev(idx))   (NTT::LDE_powers(stream, d_inout, lg), ntt/ntt.cuh:352-356)
    void lde_powers(const gpu_info& gpu, F* d, unsigned lg, hipStream_t stream)
    {
        if (lg > F::TWO_ADICITY) HIP_OK(hipErrorInvalidValue);
        const table_set ts = tables(gpu.hip_id, lg, 0, stream);
        ntt_tables<F> G{ts.glo, ts.ghi, nullptr, lg, ts.h, ts.scale, nullptr};
        const size_t n = (size_t)1 << lg;
        hipLaunchKernelGGL(k_coset<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d, G, 1);
        HIP_OK(hipGetLastError());
    }

    // d_out[idx << lg_blowup] = d_in[idx] (* g^rev(idx) when |shift|), zeros elsewhere
    // (NTT::LDE_expand / LDE_launch, ntt/ntt.cuh:247-281,358-365).  The buffers either do not overlap or
    // d_in is aligned to the END of d_out ("d_out is expected to encompass d_in and d_in is expected to be
    // aligned to the end of d_out", ntt.cuh:358-360; any other overlap is an error, as the reference's
    // assert, ntt/kernels.cu:176).  In place: output o needs input o >> lg_blowup, which sits at position
    // P(o) = ext - dom + (o >> lg_blowup) >= o, so an output range [lo, hi) may be written in one launch
    // as long as P(lo) >= hi: hi = ext - (ext - lo) / blowup.  The ranges shrink geometrically
    // ([0, ext - dom), then dom (1 - 1/blowup) elements, ...): 2 + lg_domain / lg_blowup launches, kernel
    // boundaries instead of the reference's cooperative grid sync (kernels.cu:199-200).
    void lde_spread(const gpu_info& gpu, F* d_out, const F* d_in, unsigned lg_domain, unsigned lg_blowup,
                    bool shift, hipStream_t stream)
    {
        if (lg_domain + lg_blowup > F::TWO_ADICITY) HIP_OK(hipErrorInvalidValue);
        const size_t dom = (size_t)1 << lg_domain, ext = dom << lg_blowup;
        const bool overlap = (d_in < d_out + ext) && (d_out < d_in + dom);
        if (overlap && (lg_blowup == 0 || d_in != d_out + (ext - dom))) HIP_OK(hipErrorInvalidValue);
        const table_set ts = tables(gpu.hip_id, lg_domain, 0, stream);
        ntt_tables<F> G{ts.glo, ts.ghi, nullptr, lg_domain, ts.h, ts.scale, nullptr};
        for (size_t lo = 0; lo < ext;) {
            const size_t hi = overlap ? ext - ((ext - lo) >> lg_blowup) : ext;
            hipLaunchKernelGGL(k_lde_spread<F>, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, stream,
                               d_out, d_in, G, lg_domain, lg_blowup, (int)shift, lo, hi);
            lo = hi;
        }
        HIP_OK(hipGetLastError());
    }

    // Low-degree extension on the coset g*H' (NTT::LDE_aux, ntt/ntt.cuh:283-336):
    // iNTT(NR) of the 2^lg_domain evaluations in d_ext[0 .. 2^lg_domain), coset shift +
    // zero-extension in bit-reversed order, forward NTT(RN) of size 2^(lg_domain+lg_blowup).
    // d_tmp: 2^lg_domain scratch elements; d_aux (nullable): the coefficients, natural order.
    void lde(const gpu_info& gpu, F* d_ext, F* d_tmp, F* d_aux, unsigned lg_domain, unsigned lg_blowup, hipStream_t stream)
    {
        if (lg_domain + lg_blowup > F::TWO_ADICITY) HIP_OK(hipErrorInvalidValue);
        const size_t dom = (size_t)1 << lg_domain;
        // (the inverse transform runs in the first 2^lg_domain elements of d_ext -- overwritten below anyway -- and leaves the
        //  coefficients in d_tmp: its last step stores there, no copy in front of it)
        const lde_input lo{nullptr, 0, 0, d_tmp};
        run(gpu, d_ext, lg_domain, NTT_NR, NTT_INVERSE, NTT_STANDARD, stream, &lo);
        if (d_aux) {
            hipLaunchKernelGGL(k_bitrev_copy<F>, dim3((unsigned)((dom + 255) / 256)), dim3(256), 0, stream, d_aux, d_tmp, lg_domain);
            HIP_OK(hipGetLastError());
        }
        // (the spread + coset shift of LDE_launch ride in the forward transform's first step where the plan allows: 134 MB
        //  never written and a launch less at 2^22 -> 2^24)
        const lde_input li{d_tmp, lg_domain, lg_blowup, nullptr};
        run(gpu, d_ext, lg_domain + lg_blowup, NTT_RN, NTT_FORWARD, NTT_STANDARD, stream, &li);
    }
};

} // namespace sppark_amd
