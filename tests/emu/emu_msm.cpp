// HOST EMULATION TEST HARNESS (tests only; never linked into the product).
//
// Compiles the product's per-work-item kernel bodies (sppark_amd/csrc/msm/
// msm_kernels.hpp, ec/xyzz_dev.hpp, ff/mont_dev.hpp) for the host CPU with
// -DSPPARK_HOST_EMULATION and runs the MSM pipeline by looping over the work
// items the GPU grid would cover, in the order msm_driver.hpp launches them.
// Purpose: exercise digit recoding, chunk walking / flush logic, the short-segment
// join, the record levels and the chunked bucket-sum levels in the GPU-less build
// container before spending GPU minutes.  Results are compared with the oracle by
// tests/test_emulation.py.
//
// WHAT IT COVERS: the product's own bodies of recode_digits / load_scalar_abs, accumulate_chunk,
// join_runs_item, reduce_runs_chunk, bucket_level1_item / bucket_levelN_item, bucket_top_gather /
// bucket_top_finish / bucket_top_sum_gather, the point conversion and finalisation, and the field /
// point classes' C paths.
// WHAT IT DOES NOT: the sort.  The grouped index list is produced below by a PLAIN HOST COUNTING
// SORT with the same output contract (indices grouped by bucket, offsets per bucket) -- it is not a
// translation of msm_sort_kernels.hpp: the LDS-staged scatter (k_scatterA_staged), both k_sortB
// paths, the k_big_* split for oversized partitions, the per-window k_lo width (window_lb), the
// window-group indexing and k_bitmap_accumulate are work-group-level kernels (LDS, barriers,
// cross-lane scans) and are checked by the GPU suite only (tests/test_msm_gpu.py: sort partition
// paths, oversized partitions, window groups, bitmaps).  The subset-sum top of the bucket sums
// (k_bucket_top_bits / _sum) IS covered: its per-lane functions (item -> lane map, bit selection,
// doublings) run here, with the LDS tree between them as a plain pairwise reduction.
#define SPPARK_HOST_EMULATION 1
#include "../../sppark_amd/csrc/msm/curve_select.hpp"
#include "../../sppark_amd/csrc/msm/msm_kernels.hpp"
#include "../../sppark_amd/csrc/msm/msm_sort_records.hpp"
#include "../../sppark_amd/csrc/msm/msm_piece_kernels.hpp"
#include "../../sppark_amd/csrc/ec/jacobian_host.hpp"
#include "../../sppark_amd/csrc/ff/fp2_host.hpp"
#ifdef SPPARK_G2
#include "../../sppark_amd/csrc/msm/msm_g2c_kernels.hpp"
#endif
#include <vector>
#include <algorithm>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <mutex>
#include <thread>

using namespace sppark_amd;

// Work-groups that meet at barriers (the cooperative G2 accumulation, msm/msm_g2c_kernels.hpp: a pair of waves = 128 host
// threads here; as tests/emu/emu_coop.cpp does for the four-wave operations): the two hooks of ec/xyzz_coop.hpp
namespace {
struct wg_barrier {
    std::mutex m; std::condition_variable cv;
    unsigned count = 0, gen = 0, n = 128;
    int vote = 0, result = 0;
    int wait(int v)
    {
        std::unique_lock<std::mutex> lk(m);
        vote |= v;
        const unsigned g = gen;
        if (++count == n) { result = vote; vote = 0; count = 0; gen++; cv.notify_all(); return result; }
        cv.wait(lk, [&] { return gen != g; });
        return result;
    }
} g_bar;
void run_group(unsigned nthreads, const std::function<void(unsigned)>& body)
{
    g_bar.n = nthreads;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; t++) th.emplace_back(body, t);
    for (auto& t : th) t.join();
}
int g_g2c = 0;              // the G2 accumulation by wave pairs (msm_tunables::g2_coop)
} // namespace
extern "C" void sppark_emu_barrier() { (void)g_bar.wait(0); }
extern "C" int sppark_emu_barrier_or(int v) { return g_bar.wait(v); }
extern "C" void emu_g2c(int on) { g_g2c = on; }

// plan/tunables only (no HIP runtime needed)
struct msm_plan { unsigned n, wbits, nwins, nbits, NB, HB, LB, NA, L, chunks_per_win, nslabs, slab_sz, F, K, IB, SH, NG; };

static unsigned lg2_floor(size_t x) { unsigned r = 0; while (x >>= 1) r++; return r; }

extern "C" int emu_field_op(int field, int op, void* out, const void* a, const void* b, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        if (field == 0) {
            fp_d x = fp_d::from_wire((const u32*)a + i * fp_d::N), y = fp_d::from_wire((const u32*)b + i * fp_d::N), r;
            switch (op) { case 0: r = x + y; break; case 1: r = x - y; break; case 2: r = x * y; break; case 3: r = x.sqr(); break;
                          case 4: r = x.neg(); break; case 5: r = x.from(); break; case 6: r = x.to(); break; default: r = x.dbl(); }
            r.to_wire((u32*)out + i * fp_d::N);
        } else {
            fr_d x, y, r; memcpy(&x, (const char*)a + i * sizeof(x), sizeof(x)); memcpy(&y, (const char*)b + i * sizeof(y), sizeof(y));
            switch (op) { case 0: r = x + y; break; case 1: r = x - y; break; case 2: r = x * y; break; case 3: r = x.sqr(); break;
                          case 4: r = x.neg(); break; case 5: r = x.from(); break; case 6: r = x.to(); break; default: r = x.dbl(); }
            memcpy((char*)out + i * sizeof(r), &r, sizeof(r));
        }
    }
    return 0;
}

// montx_dev::from_std (wire words -> the bucket field's own limbs; a shift and the subtraction of q p, ff/montx_dev.hpp): the limbs
// of every result, for a big-integer check on the Python side.  info = { limbs, limb bits, domain shift }; 0 when the library's
// bucket field has no internal form
template<class FX> static int from_std_limbs(unsigned* out, const void* a, size_t n, unsigned info[3])
{
    info[0] = FX::NL; info[1] = FX::RBITS / FX::NL; info[2] = FX::SH;
    for (size_t i = 0; i < n; i++) {
        const FX r = FX::from_std((const u32*)a + i * FX::NW);
        for (int j = 0; j < FX::NL; j++) out[i * FX::NL + j] = r.l[j];
    }
    return 1;
}
// which 0: the G1 bucket field of this library; 1: 28-bit limbs over the same modulus (the base field of the G2 pipeline: for
// alt_bn128 ten limbs, a domain offset of 24 bits)
extern "C" int emu_from_std_limbs(int which, unsigned* out, const void* a, size_t n, unsigned info[3])
{
    if (which == 1) return from_std_limbs<montx_dev<curve_p::fp, 28>>(out, a, n, info);
    if constexpr (field_is_montx<msm_fp_d>::value) return from_std_limbs<msm_fp_d>(out, a, n, info);
    return 0;
}

// op 0: a += b (xyzz)   1: a += affine(b)   2: a -= affine(b)   3: a = 2a
extern "C" int emu_xyzz_op(int op, void* out, const void* a, const void* b, size_t n)
{
    const wire_bucket_m* pa = (const wire_bucket_m*)a; wire_bucket_m* po = (wire_bucket_m*)out;
    for (size_t i = 0; i < n; i++) {
        wire_bucket_d p = wire_bucket_d::load(&pa[i]);
        if (op == 0) p.add(wire_bucket_d::load((const wire_bucket_m*)b + i));
        else if (op == 3) p.dbl();
        else { affine_dev<fp_d> q = load_affine<fp_d, false>((const unsigned char*)b, i, 8 * fp_d::N); p.madd(q, op == 2); }
        p.store(&po[i]);
    }
    return 0;
}

// add_pairs / dbl_pairs (the low-latency forms used at the top of the bucket sums) against add / dbl on the
// same operands, compared in the canonical wire form: |points| = n affine points in the wire format; returns
// the number of mismatches (the fields without their own records have no such forms: 0).
template<class FPX> static int pairs_check(const unsigned char* points, size_t stride, size_t n)
{
    int bad = 0;
    if constexpr (field_is_montx<FPX>::value) {
        typedef FPX inst_fp;                                // (a dependent name: the branch is discarded for the other fields)
        typedef xyzz_dev<inst_fp> B;
        std::vector<uint4> conv((size_t)n * affine_loader<inst_fp>::STRIDE / 16 + 1);
        for (size_t i = 0; i < n; i++) affine_loader<inst_fp>::template convert<false>((unsigned char*)conv.data(), points, i, (unsigned)stride);
        auto pt = [&](size_t i) { return load_affine<inst_fp, false>((const unsigned char*)conv.data(), i, 0); };
        auto same = [&](const B& a, const B& b) {
            xyzz_mem<FPX::NW> sa, sb;
            a.store_std(&sa); b.store_std(&sb);
            // XYZZ representatives differ between formulas only by the common factor; both run the SAME formulas here
            return memcmp(&sa, &sb, sizeof(sa)) == 0;
        };
        B acc; acc.set_inf();
        for (size_t i = 0; i + 2 < n; i++) {
            B q; q.set(pt(i), i & 1); q.madd(pt(i + 1), false); q.madd(pt(i + 2), true);   // a bucket with non-trivial ZZ
            B a = acc, b = acc;
            a.add(q); b.add_pairs(q);
            if (!same(a, b)) bad++;
            B c = q, d = q;
            c.dbl(); d.dbl_pairs();
            if (!same(c, d)) bad++;
            B e = q, f = q; e.add(q); f.add_pairs(q);                                      // equal operands: doubling branch
            if (!same(e, f)) bad++;
            B g = q, h = q; B nq = q; nq.Y = inst_fp::template neg<6, 4>(q.Y).norm();      // q + (-q) = infinity
            g.add(nq); h.add_pairs(nq);
            if (!same(g, h) || !g.is_inf()) bad++;
            acc = a;
        }
    }
    return bad;
}
extern "C" int emu_pairs_check(const unsigned char* points, size_t stride, size_t n)
{   return pairs_check<inst_fp>(points, stride, n);   }

// the two extra steps of fields with their own records (k_convert_points / k_finalize)
template<class F>
static const unsigned char* convert_points(std::vector<uint4>& conv, const unsigned char* pts, size_t npoints, size_t stride, bool flagged)
{
    if constexpr (field_is_internal<F>::value) {
        conv.resize((size_t)npoints * affine_loader<F>::STRIDE / 16 + 1);
        for (size_t i = 0; i < npoints; i++) {
            if (flagged) affine_loader<F>::template convert<true>((unsigned char*)conv.data(), pts, i, (unsigned)stride);
            else         affine_loader<F>::template convert<false>((unsigned char*)conv.data(), pts, i, (unsigned)stride);
        }
        return (const unsigned char*)conv.data();
    } else {
        return pts;
    }
}
template<class F, class M>
static void finalize_sum(M* out, const xyzz_mem<F::N>* in)
{
    if constexpr (field_is_internal<F>::value) xyzz_dev<F>::load(in).store_std(out);
    else memcpy(out, in, sizeof(*out));
}

static unsigned g_pack = 0;
extern "C" void emu_msm_pack(unsigned mode) { g_pack = mode; }       // 0: 8-byte level-A records
static unsigned g_piece_cmax = 0;                                   // join == 2: pieces per bucket the piece tree takes (0: from the average bucket)
extern "C" void emu_msm_piece_cmax(unsigned c) { g_piece_cmax = c; }
static size_t g_piece_fuse = 0;                                     // join == 2: work items from which the rest of the piece tree runs work-group-major (0: never)
extern "C" void emu_msm_piece_fuse(size_t f) { g_piece_fuse = f; }
static unsigned emu_top_pieces = 1;                                 // the subset-sum top by pieces of a sum: 1 = the product's cut, 0 = per sum, "sb sp" forced
extern "C" void emu_msm_top_pieces(unsigned on) { emu_top_pieces = on; }

extern "C" int emu_msm(void* out_jac, const unsigned char* points, size_t stride, size_t npoints,
                       const unsigned char* scalars, int mont,
                       unsigned wbits, unsigned L, unsigned F, unsigned K, unsigned nslabs, int join, unsigned* join_stats,
                       unsigned top /* bucket sums: items per window handed to the subset-sum top; 0 = 4096, 1 = never */)
{
#ifdef SPPARK_G2                                 // the same pipeline over Fp2 (G2)
    typedef fp2_host<curve_p::fp> fp_h;
#else
    typedef mont_host<curve_p::fp> fp_h;
#endif
    typedef jacobian_host<fp_h> point_t;
    point_t out; out.set_inf();
    if (npoints == 0) { memcpy(out_jac, &out, sizeof(out)); return 0; }

    msm_plan p;
    p.n = (unsigned)npoints;
    unsigned lg = lg2_floor(npoints);
    p.wbits = wbits ? wbits : std::min(20u, std::max(6u, lg > 6 ? lg - 6 : 0u));
    p.nwins = (curve_p::fr::NBITS - 1) / p.wbits + 1;
    p.nbits = curve_p::fr::NBITS;
    p.wbits = p.nbits / p.nwins + (p.nbits % p.nwins ? 1 : 0);
    p.NB = 1u << (p.wbits - 1);
    p.LB = (p.wbits - 1) / 2; p.HB = p.wbits - 1 - p.LB; p.NA = 1u << p.HB;
    size_t entries = (size_t)p.n * p.nwins;
    p.L = L ? L : (unsigned)std::min<size_t>(64, std::max<size_t>(4, entries / 262144));
    p.chunks_per_win = (p.n + p.L - 1) / p.L;
    p.nslabs = nslabs ? nslabs : (unsigned)std::min<size_t>(64, std::max<size_t>(1, npoints / 262144));
    p.slab_sz = (p.n + p.nslabs - 1) / p.nslabs;
    // 4-byte level-A records (msm_plan.hpp: IB = 31 - LB, groups of 2^SH slabs).  Here with the SMALLEST index field the slabs
    // allow, so that a few hundred points already spread over several index groups: g_pack = 1: every slab is a group,
    // 2: every pair of slabs
    p.IB = p.SH = 0; p.NG = 1;
    if (g_pack) {
        unsigned lgs = 0; while ((1u << lgs) < p.slab_sz) lgs++;
        p.slab_sz = 1u << lgs; p.nslabs = (p.n + p.slab_sz - 1) / p.slab_sz;
        p.SH = g_pack - 1; p.IB = lgs + p.SH; p.NG = ((p.nslabs - 1) >> p.SH) + 1;
        if (p.IB + p.LB > 31 || p.NG > PARTA_MAX_GROUPS) return -7;
    }
    p.F = std::max(4u, F ? F : 32u);
    p.K = std::min(K ? K : 8u, p.NB);
    const bool flagged = stride > 2 * sizeof(fp_h);

    // ---- breakdown (k_breakdown) ----
    std::vector<u32> digits((size_t)p.nwins * p.n), sorted((size_t)p.nwins * p.n);
    std::vector<u32> sc32((size_t)p.n * fr_d::N + 8);
    memcpy(sc32.data(), scalars, (size_t)p.n * sizeof(fr_d));
    for (unsigned i = 0; i < p.n; i++) {
        bool flip;
        fr_d s = load_scalar_abs<fr_d>(sc32.data(), i, mont, flip);
        u32 limbs[fr_d::N + 2] = {0};
        for (int k = 0; k < fr_d::N; k++) limbs[k] = s.v[k];
        recode_digits(digits.data(), p.n, i, [&](unsigned k) { return limbs[k]; }, flip, p.nwins, p.nbits);
    }

    // ---- two-level counting sort (translation of msm_sort_kernels.hpp) ----
    std::vector<u32> H((size_t)p.nwins * p.nslabs * p.NA, 0), tot((size_t)p.nwins * p.NA), offA((size_t)p.nwins * (p.NA + 1));
    std::vector<u32> off((size_t)p.nwins * (p.NB + 1));
    // level-A records through the product's own pack / unpack (msm_sort_records.hpp): 4 bytes where the plan says so, the
    // index bits above IB recovered from the record's position in its partition
    std::vector<uint2> partA((size_t)p.nwins * p.n);
    std::vector<u32> partP((size_t)p.nwins * p.n);
    const bool packed = p.IB != 0;
    unsigned ngp = 1; while (ngp < p.NG) ngp <<= 1;
    if (ngp > PARTA_MAX_GROUPS) return -7;
    const partA_fmt fmt{H.data(), p.IB, p.SH, ngp, p.nslabs};
    const u32 lomask = (1u << p.LB) - 1;
    for (unsigned w = 0; w < p.nwins; w++)                              // k_histA
        for (unsigned slab = 0; slab < p.nslabs; slab++) {
            unsigned lo = slab * p.slab_sz, hi = std::min(p.n, lo + p.slab_sz);
            u32* cnt = &H[((size_t)w * p.nslabs + slab) * p.NA];
            for (unsigned i = lo; i < hi; i++) { u32 d = digits[(size_t)w * p.n + i]; if (d) cnt[((d & 0x7fffffffu) - 1) >> p.LB]++; }
        }
    for (size_t id = 0; id < (size_t)p.nwins * p.NA; id++) {            // k_scan_slabs
        unsigned w = id / p.NA, b = id % p.NA; u32 run = 0;
        for (unsigned s = 0; s < p.nslabs; s++) { u32* q = &H[((size_t)w * p.nslabs + s) * p.NA + b]; u32 t = *q; *q = run; run += t; }
        tot[id] = run;
    }
    for (unsigned w = 0; w < p.nwins; w++) {                            // k_scan_parts
        u32 run = 0;
        for (unsigned b = 0; b < p.NA; b++) { offA[(size_t)w * (p.NA + 1) + b] = run; run += tot[(size_t)w * p.NA + b]; }
        offA[(size_t)w * (p.NA + 1) + p.NA] = run;
    }
    for (unsigned w = 0; w < p.nwins; w++)                              // k_scatterA
        for (unsigned slab = 0; slab < p.nslabs; slab++) {
            std::vector<u32> cur(p.NA);
            for (unsigned b = 0; b < p.NA; b++) cur[b] = H[((size_t)w * p.nslabs + slab) * p.NA + b] + offA[(size_t)w * (p.NA + 1) + b];
            unsigned lo = slab * p.slab_sz, hi = std::min(p.n, lo + p.slab_sz);
            for (unsigned i = hi; i-- > lo;) {                          // reverse: any order inside a partition must work
                u32 d = digits[(size_t)w * p.n + i];
                if (d) {
                    u32 k = (d & 0x7fffffffu) - 1; const size_t at = (size_t)w * p.n + cur[k >> p.LB]++;
                    if (packed) partP[at] = recA<true>::make(i | (d & 0x80000000u), k & lomask, p.LB, p.IB);
                    else        partA[at] = recA<false>::make(i | (d & 0x80000000u), k & lomask, p.LB, p.IB);
                }
            }
        }
    for (unsigned w = 0; w < p.nwins; w++)                              // k_sortB
        for (unsigned khi = 0; khi < p.NA; khi++) {
            unsigned begin = offA[(size_t)w * (p.NA + 1) + khi], end = offA[(size_t)w * (p.NA + 1) + khi + 1];
            std::vector<u32> cnt(1u << p.LB, 0);
            u32 bnd[PARTA_MAX_GROUPS];
            if (packed) for (unsigned t = 0; t < ngp; t++) partA_bounds(bnd, fmt, w, khi, p.NA, t);
            auto key = [&](unsigned i) { return packed ? recA<true>::key(partP[(size_t)w * p.n + i], lomask) : recA<false>::key(partA[(size_t)w * p.n + i], lomask); };
            auto ent = [&](unsigned i) { return packed ? recA<true>::entry(partP[(size_t)w * p.n + i], i - begin, p.LB, fmt, bnd)
                                                       : recA<false>::entry(partA[(size_t)w * p.n + i], i - begin, p.LB, fmt, bnd); };
            for (unsigned i = begin; i < end; i++) cnt[key(i)]++;
            u32 run = begin;
            for (unsigned b = 0; b < (1u << p.LB); b++) { u32 c = cnt[b]; cnt[b] = run; off[(size_t)w * (p.NB + 1) + ((size_t)khi << p.LB) + b] = run; run += c; }
            if (khi == p.NA - 1) off[(size_t)w * (p.NB + 1) + p.NB] = end;
            for (unsigned i = end; i-- > begin;) sorted[(size_t)w * p.n + cnt[key(i)]++] = ent(i);
        }

    // ---- accumulate + record levels ----
    std::vector<inst_m> buckets((size_t)p.nwins * p.NB);
    // NOT cleared (as msm_driver.hpp with one window group): an empty bucket is recognised from the sort's
    // offsets and never read -- garbage here would show up in the result if it were
    memset(buckets.data(), 0xAB, buckets.size() * sizeof(inst_m));
    size_t nrecA = (size_t)2 * p.nwins * p.chunks_per_win, nrecB = 2 * ((nrecA + p.F - 1) / p.F);
    std::vector<u32> keyA(nrecA), keyB(nrecB);
    std::vector<inst_m> ptA(nrecA), ptB(nrecB);
    // the device gather may read up to the padded stride; copy points into an 8-byte aligned buffer
    std::vector<uint64_t> pts_al((npoints * stride + 15) / 8);
    memcpy(pts_al.data(), points, npoints * stride);
    const unsigned char* pts = (const unsigned char*)pts_al.data();
    // fields with their own records (ff/montx_dev.hpp): k_convert_points first, as msm_driver.hpp does
    std::vector<uint4> conv;
    pts = convert_points<inst_fp>(conv, pts, npoints, stride, flagged);
#ifdef SPPARK_G2
    if constexpr (field_is_internal<inst_fp>::value && !field_is_montx<inst_fp>::value) {
        if (g_g2c) {                                // k_accumulate_g2c: 64 chunks per pair of waves
            static g2c_lds<inst_fp> ex;
            for (unsigned w = 0; w < p.nwins; w++)
                for (unsigned c0 = 0; c0 < ((p.chunks_per_win + 63) / 64) * 64; c0 += 64)
                    run_group(G2C_NT, [&](unsigned tid) {
                        const g2c_ctx<inst_fp> c{&ex, tid >> 6, tid & 63};
                        if (c.role == 0) accumulate_chunk_g2c<inst_fp, 0>(buckets.data(), keyA.data(), ptA.data(), pts, sorted.data(), off.data(),
                                                                                 p.n, p.NB, p.L, p.chunks_per_win, c0 + c.lane, w, 0, c);
                        else             accumulate_chunk_g2c<inst_fp, 1>(buckets.data(), keyA.data(), ptA.data(), pts, sorted.data(), off.data(),
                                                                                 p.n, p.NB, p.L, p.chunks_per_win, c0 + c.lane, w, 0, c);
                    });
        }
    }
    if (!g_g2c)
#endif
    for (unsigned w = 0; w < p.nwins; w++)
        for (unsigned chunk = 0; chunk < ((p.chunks_per_win + 255) / 256) * 256; chunk++) {
            if (flagged) accumulate_chunk<inst_fp, true>(buckets.data(), keyA.data(), ptA.data(), pts, (unsigned)stride,
                                                      sorted.data(), off.data(), p.n, p.NB, p.L, p.chunks_per_win, chunk, w);
            else         accumulate_chunk<inst_fp, false>(buckets.data(), keyA.data(), ptA.data(), pts, (unsigned)stride,
                                                       sorted.data(), off.data(), p.n, p.NB, p.L, p.chunks_per_win, chunk, w);
        }
    {
        size_t nrec = nrecA;
        u32 *ik = keyA.data(), *ok = keyB.data(); inst_m *ip = ptA.data(), *op = ptB.data();
        // k_join_runs: segments of <= JOIN_WALK records are summed here; the tree sees the rest
        std::vector<u32> keyC(nrecA);
        u32 any_long = 0;
        if (join == 1) {
            for (size_t t = 0; t < ((nrec / 2 + 1 + 255) / 256) * 256; t++)
                join_runs_item<inst_fp>(buckets.data(), keyC.data(), keyA.data(), ptA.data(), (unsigned)nrec, &any_long, t);
            ik = keyC.data();
        }
        if (join_stats) { join_stats[0] = any_long; join_stats[1] = 0; for (size_t i = 0; join == 1 && i < nrec; i++) join_stats[1] += keyC[i] != KEY_NONE; }
        if (join == 2) {
            // the piece tree of the small sizes (msm_piece_kernels.hpp), as msm_driver.hpp launches it: log2(cmax) levels over
            // (bucket, pair) work items; the records of buckets with more pieces keep their keys and go through the fan-in tree
            const unsigned cmax = g_piece_cmax ? g_piece_cmax : piece_cmax((size_t)p.n / p.NB / p.L + 1);
            any_long = 0;
            // g_piece_fuse: the levels of at most that many work items in the order of k_piece_tail_coop -- work-group by
            // work-group (2^lgGB buckets with all their pairs), every level of a work-group before the next work-group starts
            const size_t nbk = (size_t)p.nwins * p.NB;
            const unsigned t_fused = g_piece_fuse ? piece_tail_t0(nbk, cmax, g_piece_fuse) : ~0u;
            for (unsigned t = 0; (cmax >> (t + 1)) >= 1; t++) {
                const unsigned last = (cmax >> (t + 2)) == 0;
                const size_t nthr = (size_t)p.nwins * p.NB * (cmax >> (t + 1));
                if (t == t_fused) {
                    const unsigned lgGB = piece_tail_lgGB(cmax, t);
                    for (size_t wg = 0; wg < ((nbk + ((size_t)1 << lgGB) - 1) >> lgGB); wg++)
                        for (unsigned tt = t; (cmax >> (tt + 1)) >= 1; tt++) {
                            const unsigned lst = (cmax >> (tt + 2)) == 0, njobs = (cmax >> (tt + 1)) << lgGB;
                            for (unsigned idx = 0; idx < ((njobs + 63) / 64) * 64; idx++) {
                                const size_t B = (wg << lgGB) + (idx & ((1u << lgGB) - 1));
                                if (idx < njobs && B < nbk)
                                    piece_apply<inst_fp>(buckets.data(), ptA.data(), piece_job_bm(keyA.data(), off.data(), p.NB, p.L, p.chunks_per_win,
                                                                                                cmax, tt, lst, &any_long, B, idx >> lgGB));
                            }
                        }
                    break;
                }
                for (size_t id = 0; id < ((nthr + 255) / 256) * 256; id++)
                    piece_level_item<inst_fp>(buckets.data(), keyA.data(), ptA.data(), off.data(), p.NB, p.L, p.chunks_per_win, p.nwins,
                                              cmax, t, last, &any_long, id);
            }
            ik = keyA.data();
            if (join_stats) { join_stats[0] = any_long; join_stats[1] = 0; for (size_t i = 0; i < nrec; i++) join_stats[1] += keyA[i] != KEY_NONE; }
        }
        if (!join || any_long)
        for (;;) {
            unsigned nthreads = (unsigned)((nrec + p.F - 1) / p.F);
            int last = nthreads == 1;
            for (unsigned t = 0; t < ((nthreads + 255) / 256) * 256; t++)
                reduce_runs_chunk<inst_fp>(buckets.data(), ok, op, ik, ip, (unsigned)nrec, p.F, nthreads, last, t);
            if (last) break;
            nrec = (size_t)2 * nthreads;
            std::swap(ik, ok); std::swap(ip, op);
        }
    }

    // ---- bucket sums ----
    size_t n1 = (size_t)p.nwins * (p.NB / p.K);
    std::vector<inst_m> A1(n1), W1(n1), A2(n1), W2(n1);
    unsigned nitems = p.NB / p.K;
    for (size_t id = 0; id < n1; id++) bucket_level1_item<inst_fp>(A1.data(), W1.data(), buckets.data(), p.NB, p.K, p.nwins, id, off.data());
    unsigned lgG = lg2_floor(p.K);
    inst_m *ia = A1.data(), *iw = W1.data(), *oa = A2.data(), *ow = W2.data();
    // top == 2: the small windows' bucket sums straight from the buckets (k_bucket_small_bits_coop: small_sums_gather, a pairwise
    // tree, b doublings per part; then the parts of a window as in the subset-sum top), as msm_driver.hpp runs them for NB <= 256
    if (top == 2 && p.NB <= SMALL_SUMS_MAX_NB && p.NB >= 2) {
        const unsigned m = lg2_floor(p.NB);
        std::vector<inst_m> parts((size_t)p.nwins * (m + 1));
        std::vector<inst_m> res(p.nwins);
        for (unsigned w = 0; w < p.nwins; w++) {
            for (unsigned b = 0; b <= m; b++) {
                std::vector<xyzz_dev<inst_fp>> acc(SMALL_SUMS_MAX_NB / 2);
                for (unsigned t = 0; t < SMALL_SUMS_MAX_NB / 2; t++) acc[t] = small_sums_gather<inst_fp>(buckets.data(), off.data(), p.NB, m, b, w, t);
                const unsigned cnt = b >= m ? 1u : p.NB / 2;
                for (unsigned s = cnt / 2; s >= 1; s >>= 1)
                    for (unsigned t = 0; t < s; t++) bucket_add_fast<inst_fp>(acc[t], acc[t + s]);
                for (unsigned k = 0; k < b; k++) bucket_dbl_fast<inst_fp>(acc[0]);
                acc[0].store(&parts[(size_t)w * (m + 1) + b]);
            }
            std::vector<xyzz_dev<inst_fp>> acc(32);
            for (unsigned tid = 0; tid < 32; tid++) acc[tid] = bucket_top_sum_gather<inst_fp>(parts.data(), m, w, tid);
            for (unsigned s = 16; s >= 1; s >>= 1)
                for (unsigned tid = 0; tid < s; tid++) bucket_add_fast<inst_fp>(acc[tid], acc[tid + s]);
            acc[0].store(&W2[w]);
        }
        iw = W2.data();
        nitems = 1;
    }
    while (nitems > 1) {
        // the subset-sum top (k_bucket_top_bits / k_bucket_top_sum), as msm_driver.hpp hands over to it: the per-lane parts
        // are the product's own functions, the LDS tree between them a plain pairwise reduction with the same additions
        if (top != 1 && nitems <= (top ? top : BUCKET_TOP_MAX) && nitems >= 32 && (nitems & (nitems - 1)) == 0 && p.NB / p.K >= 32) {
            const unsigned m = lg2_floor(nitems);
            // emu_top_pieces: a work-group per PIECE of a sum (bucket_top_piece / bucket_top_cut, as k_bucket_top_bits_coop
            // runs them: the gather of a piece with its virtual lane numbers, the doublings per piece, the parts of a
            // window = its pieces); off: per sum (k_bucket_top_bits)
            unsigned sb = 1, sp = 1;
            if (emu_top_pieces == 1) bucket_top_cut(nitems, BUCKET_TOP_NT, sb, sp);
            else if (emu_top_pieces >= 10 && nitems >= BUCKET_TOP_NT * (emu_top_pieces % 10) && m * (emu_top_pieces / 10) + emu_top_pieces % 10 <= 32) {
                sb = emu_top_pieces / 10; sp = emu_top_pieces % 10;         // a forced cut "sb sp" (two digits), as SPPARK_TOP_CUT
            }
            const unsigned np = m * sb + sp;
            std::vector<inst_m> parts((size_t)p.nwins * np);
            auto tree = [&](std::vector<xyzz_dev<inst_fp>>& acc, unsigned nt) {
                for (unsigned s = nt >> 1; s >= 1; s >>= 1)
                    for (unsigned tid = 0; tid < s; tid++) bucket_add_fast<inst_fp>(acc[tid], acc[tid + s]);
            };
            for (unsigned w = 0; w < p.nwins; w++) {
                for (unsigned q = 0; q < np; q++) {
                    const top_piece pc = bucket_top_piece(q, m, sb, sp);
                    std::vector<xyzz_dev<inst_fp>> acc(BUCKET_TOP_NT);
                    for (unsigned tid = 0; tid < BUCKET_TOP_NT; tid++)
                        acc[tid] = bucket_top_gather<inst_fp>(ia, iw, nitems, m, pc.b, w, pc.sub * BUCKET_TOP_NT + tid, pc.nsub * BUCKET_TOP_NT);
                    tree(acc, BUCKET_TOP_NT);
                    if (pc.b < m) for (unsigned k = 0; k < pc.b + lgG; k++) bucket_dbl_fast<inst_fp>(acc[0]);
                    acc[0].store(&parts[(size_t)w * np + q]);
                }
                std::vector<xyzz_dev<inst_fp>> acc(32);
                for (unsigned tid = 0; tid < 32; tid++) acc[tid] = bucket_top_sum_gather<inst_fp>(parts.data(), np - 1, w, tid);
                tree(acc, 32);
                acc[0].store(&ow[w]);
            }
            std::swap(iw, ow);
            break;
        }
        unsigned Kc = std::min(p.K, nitems);
        size_t nthr = (size_t)p.nwins * (nitems / Kc);
        for (size_t id = 0; id < nthr; id++) bucket_levelN_item<inst_fp>(oa, ow, ia, iw, nitems, Kc, lgG, p.nwins, id);
        nitems /= Kc; lgG += lg2_floor(Kc);
        std::swap(ia, oa); std::swap(iw, ow);
    }
    // k_finalize: window sums back to the reference's wire image
    std::vector<xyzz_mem<sizeof(fp_h) / 4>> fin(p.nwins);
    for (unsigned w = 0; w < p.nwins; w++) finalize_sum<inst_fp>(&fin[w], &iw[w]);
    for (unsigned w = p.nwins; w--;) {
        fp_h c[4];
        memcpy(c, &fin[w], sizeof(c));
        point_t s = point_t::from_xyzz(c[0], c[1], c[2], c[3]);
        out.add(s);
        if (w) for (unsigned k = 0; k < window_len(w - 1, p.nwins, p.nbits); k++) out.dbl();
    }
    memcpy(out_jac, &out, sizeof(out));
    return 0;
}

// Fixed-base mode (msm_kernels.hpp fixed_base_table_item): build the tables of 2^(off_j) * P_i with the product's
// own work item, recode the scalars with the product's recode_digits for the REAL (n, nwins, nbits), and sum the
// nwins * n (digit, table entry) pairs into ONE bucket set with the entry index w * n + i -- the index contract of
// msm_driver.hpp's fixed_plan() / invoke_fixed().  The grouped walk itself (accumulate_chunk etc.) is emu_msm's
// subject; here the buckets are filled in entry order and summed by the running-sum rule.  G1 fields with their
// own records only (returns 1 otherwise).
template<class FPX> static int fixed_base_emu(void* out_jac, const unsigned char* points, size_t stride, size_t npoints,
                                              const unsigned char* scalars, unsigned wbits)
{
    if constexpr (field_is_montx<FPX>::value) {
        typedef FPX F;
        typedef mont_host<curve_p::fp> fp_h;
        typedef jacobian_host<fp_h> point_t;
        typedef xyzz_dev<F> B;
        const unsigned nbits = curve_p::fr::NBITS, n = (unsigned)npoints;
        unsigned nwins = (nbits - 1) / wbits + 1;
        wbits = nbits / nwins + (nbits % nwins ? 1 : 0);
        const unsigned NB = 1u << (wbits - 1);
        const size_t S = affine_loader<F>::STRIDE;
        std::vector<uint4> table((size_t)nwins * n * S / 16 + 1);
        const bool flagged = stride > 2 * sizeof(fp_h);
        for (size_t i = 0; i < n; i++) {
            if (flagged) affine_loader<F>::template convert<true>((unsigned char*)table.data(), points, i, (unsigned)stride);
            else         affine_loader<F>::template convert<false>((unsigned char*)table.data(), points, i, (unsigned)stride);
        }
        for (size_t i = 0; i < n; i++) fixed_base_table_item<F>((unsigned char*)table.data(), n, nwins, nbits, i);
        std::vector<u32> digits((size_t)nwins * n), sc32((size_t)n * fr_d::N + 8);
        memcpy(sc32.data(), scalars, (size_t)n * sizeof(fr_d));
        for (unsigned i = 0; i < n; i++) {
            bool flip;
            fr_d s = load_scalar_abs<fr_d>(sc32.data(), i, 0, flip);
            u32 limbs[fr_d::N + 2] = {0};
            for (int k = 0; k < fr_d::N; k++) limbs[k] = s.v[k];
            recode_digits(digits.data(), n, i, [&](unsigned k) { return limbs[k]; }, flip, nwins, nbits);
        }
        std::vector<B> bucket(NB);
        for (auto& b : bucket) b.set_inf();
        for (size_t e = 0; e < (size_t)nwins * n; e++) {
            const u32 d = digits[e];
            if (!d) continue;
            bucket[(d & 0x7fffffffu) - 1].madd(load_affine<F, false>((const unsigned char*)table.data(), e, 0), d >> 31);
        }
        B run, sum; run.set_inf(); sum.set_inf();           // sum_b (b + 1) * bucket[b]
        for (unsigned b = NB; b--;) { run.add(bucket[b]); sum.add(run); }
        xyzz_mem<sizeof(fp_h) / 4> fin;
        sum.store_std(&fin);
        fp_h c[4];
        memcpy(c, &fin, sizeof(c));
        point_t out = point_t::from_xyzz(c[0], c[1], c[2], c[3]);
        memcpy(out_jac, &out, sizeof(out));
        return 0;
    } else {
        return 1;
    }
}
extern "C" int emu_fixed_base(void* out_jac, const unsigned char* points, size_t stride, size_t npoints,
                              const unsigned char* scalars, unsigned wbits)
{   return fixed_base_emu<inst_fp>(out_jac, points, stride, npoints, scalars, wbits);   }

// Fp2 over the loosely-reduced field (ff/fp2x_dev.hpp), operation by operation on INTERNAL limbs (2 * NL words per
// element, as the caller built them -- including non-canonical representatives at the edge of the stated bounds):
// op 0: mul<KA>(a, b)   op 1: a.sqr<KA>()   op 2: sub<KA, 1>(a, b).norm()   op 3: neg<KA, 1>(a).norm()
// G2 builds only (returns 1 otherwise); KA in {3, 6, 10, 13}.
template<class FPX, int KA> static void fp2x_ops(int op, u32* out, const u32* a, const u32* b, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        FPX x = FPX::from_wire(a + i * FPX::N), y = FPX::from_wire(b + i * FPX::N), r;
        switch (op) {
            case 0: r = FPX::template mul<KA>(x, y); break;
            case 1: r = x.template sqr<KA>(); break;
            case 2: r = FPX::template sub<KA>(x, y).norm(); break;
            default: r = FPX::template neg<KA>(x).norm(); break;
        }
        r.to_wire(out + i * FPX::N);
    }
}
extern "C" int emu_fp2x_op(int op, int ka, void* out, const void* a, const void* b, size_t n)
{
#ifdef SPPARK_G2
    if constexpr (field_is_internal<inst_fp>::value && !field_is_montx<inst_fp>::value) {
        switch (ka) {
            case 3:  fp2x_ops<inst_fp, 3>(op, (u32*)out, (const u32*)a, (const u32*)b, n); return 0;
            case 6:  fp2x_ops<inst_fp, 6>(op, (u32*)out, (const u32*)a, (const u32*)b, n); return 0;
            case 10: fp2x_ops<inst_fp, 10>(op, (u32*)out, (const u32*)a, (const u32*)b, n); return 0;
            case 13: fp2x_ops<inst_fp, 13>(op, (u32*)out, (const u32*)a, (const u32*)b, n); return 0;
            default: return 2;
        }
    }
#endif
    (void)op; (void)ka; (void)out; (void)a; (void)b; (void)n;
    return 1;
}

// The bucket invariant of ec/xyzzx2_dev.hpp under a chain of operations: after every step the accumulator's internal
// image (X | Y | ZZZ | ZZ, 4 * N words) is appended to |out|.  Steps: set(p0); madd(p_i, i odd) for i < n; then
// add(copy of the state after n/2 steps); dbl(); madd(p_0) again.  G2 builds only (returns the number of images, 0 otherwise).
// The same chain of set / madd steps as emu_g2_chain's first part, run by the COOPERATIVE bucket class for 64 lanes at
// once: lane l starts at point l and adds points l+1, l+2, ... (indices mod n), lane 3 additionally meets its own start
// point again right away (the doubling case), lane 5 the negated start point (infinity), lane 7 starts from a point at
// infinity when one is in the list.  After every step every lane's image is appended to |out| (64 images per step) --
// and to |ref| the image of the serial class run on the same sequence.  Returns the number of steps (0: not a G2 build).
extern "C" int emu_g2c_chain(void* out, void* ref, const unsigned char* points, size_t stride, size_t n, unsigned steps)
{
#ifdef SPPARK_G2
    if constexpr (field_is_internal<inst_fp>::value && !field_is_montx<inst_fp>::value) {
        typedef xyzz_dev<inst_fp> B;
        std::vector<uint4> conv((size_t)n * affine_loader<inst_fp>::STRIDE / 16 + 1);
        const bool flagged = stride > 2 * sizeof(fp2_host<curve_p::fp>);
        for (size_t i = 0; i < n; i++) {
            if (flagged) affine_loader<inst_fp>::template convert<true>((unsigned char*)conv.data(), points, i, (unsigned)stride);
            else         affine_loader<inst_fp>::template convert<false>((unsigned char*)conv.data(), points, i, (unsigned)stride);
        }
        const unsigned char* rec = (const unsigned char*)conv.data();
        // entry s of lane l: (point index, negate)
        auto entry = [&](unsigned l, unsigned s, size_t& idx, bool& neg) {
            idx = (l + s) % n; neg = ((l + s) & 1) != 0;
            if (l == 3 && s == 1) { idx = l % n; neg = (l & 1) != 0; }           // the same point again: doubling
            if (l == 5 && s == 1) { idx = l % n; neg = (l & 1) == 0; }           // its negative: infinity
        };
        B::mem_t* o = (B::mem_t*)out; B::mem_t* rf = (B::mem_t*)ref;
        for (unsigned l = 0; l < 64; l++) {                                      // serial reference
            B acc; acc.set_inf();
            for (unsigned s = 0; s < steps; s++) {
                size_t idx; bool neg; entry(l, s, idx, neg);
                const affine_dev<inst_fp> pt = load_affine<inst_fp, false>(rec, idx, 0);
                if (s == 0) acc.set(pt, neg); else acc.madd(pt, neg);
                acc.store(&rf[(size_t)s * 64 + l]);
            }
        }
        static g2c_lds<inst_fp> ex;
        run_group(G2C_NT, [&](unsigned tid) {
            const g2c_ctx<inst_fp> c{&ex, tid >> 6, tid & 63};
            g2c_bucket<inst_fp> acc; acc.set_inf();
            for (unsigned s = 0; s < steps; s++) {
                size_t idx; bool neg; entry(c.lane, s, idx, neg);
                const g2c_affine<inst_fp> pt = g2c_affine<inst_fp>::load(rec, idx, c.role);
                bool restart = s == 0;
                if (pt.inf && restart) { acc.set_inf(); restart = false; }
                if (c.role == 0) acc.template madd<0>(pt, neg, restart, c); else acc.template madd<1>(pt, neg, restart, c);
                acc.store(&o[(size_t)s * 64 + c.lane], c.role);
            }
        });
        return (int)steps;
    }
#endif
    (void)out; (void)ref; (void)points; (void)stride; (void)n; (void)steps;
    return 0;
}
extern "C" int emu_g2_chain(void* out, const unsigned char* points, size_t stride, size_t n)
{
#ifdef SPPARK_G2
    if constexpr (field_is_internal<inst_fp>::value && !field_is_montx<inst_fp>::value) {
        typedef xyzz_dev<inst_fp> B;
        std::vector<uint4> conv((size_t)n * affine_loader<inst_fp>::STRIDE / 16 + 1);
        const bool flagged = stride > 2 * sizeof(fp2_host<curve_p::fp>);
        for (size_t i = 0; i < n; i++) {
            if (flagged) affine_loader<inst_fp>::template convert<true>((unsigned char*)conv.data(), points, i, (unsigned)stride);
            else         affine_loader<inst_fp>::template convert<false>((unsigned char*)conv.data(), points, i, (unsigned)stride);
        }
        auto pt = [&](size_t i) { return load_affine<inst_fp, false>((const unsigned char*)conv.data(), i, 0); };
        B::mem_t* o = (B::mem_t*)out;
        int k = 0;
        B acc, half; acc.set(pt(0), false); acc.store(&o[k++]);
        half = acc;
        for (size_t i = 1; i < n; i++) { acc.madd(pt(i), i & 1); acc.store(&o[k++]); if (i == n / 2) half = acc; }
        acc.add(half); acc.store(&o[k++]);
        acc.dbl(); acc.store(&o[k++]);
        acc.madd(pt(0), false); acc.store(&o[k++]);
        return k;
    }
#endif
    (void)out; (void)points; (void)stride; (void)n;
    return 0;
}


// ---- the contracts of the G2 wave-pair bucket, machine-checked (-DSPPARK_TRACK_BOUNDS; see tests/emu/emu_bounds.cpp) --------
// g2c_bucket::madd of ec/xyzz2_coop.hpp -- the plain step, the restart, the same point again (the cooperative doubling) --
// from the LOOSEST operands the invariants of ec/xyzzx2_dev.hpp allow, component by component; the claims travel through
// the exchange area with the limbs.  Returns the number of violated contracts (G2 builds; -1 otherwise).
#if defined(SPPARK_TRACK_BOUNDS)
static std::atomic<int> g_violations{0};
static std::mutex g_vm;
static char g_first[512];
extern "C" void sppark_bound_violation(const char* what, double got, double limit)
{
    std::lock_guard<std::mutex> lk(g_vm);
    if (!g_violations.load()) snprintf(g_first, sizeof(g_first), "%s: %.6g against %.6g", what, got, limit);
    g_violations++;
}
extern "C" int emu_g2c_bounds(const unsigned char* points, size_t stride, size_t n, char* msg, size_t msglen)
{
#ifdef SPPARK_G2
    if constexpr (field_is_internal<inst_fp>::value && !field_is_montx<inst_fp>::value) {
        typedef inst_fp F2; typedef F2::fp fp;
        g_violations = 0; g_first[0] = 0;
        if (n < 4) return -1;
        static const double INV[4][2] = {{9, 1}, {5, 1}, {2, 1}, {2, 1}};       // X, Y, ZZZ, ZZ (ec/xyzzx2_dev.hpp)
        std::vector<uint4> conv((size_t)n * affine_loader<F2>::STRIDE / 16 + 1);
        for (size_t i = 0; i < n; i++) affine_loader<F2>::template convert<false>((unsigned char*)conv.data(), points, i, (unsigned)stride);
        const unsigned char* rec = (const unsigned char*)conv.data();
        static g2c_lds<F2> ex;
        run_group(G2C_NT, [&](unsigned tid) {
            const g2c_ctx<F2> c{&ex, tid >> 6, tid & 63};
            auto pt = [&](size_t i) { g2c_affine<F2> p = g2c_affine<F2>::load(rec, i % n, c.role); p.X.bnd_set(2, 1); p.Y.bnd_set(2, 1); return p; };
            auto loosen = [&](g2c_bucket<F2>& b) { b.X.bnd_set(INV[0][0], INV[0][1]); b.Y.bnd_set(INV[1][0], INV[1][1]); b.ZZZ.bnd_set(INV[2][0], INV[2][1]); b.ZZ.bnd_set(INV[3][0], INV[3][1]); };
            auto check = [&](const char* op, const g2c_bucket<F2>& b) {
                const fp* k4[4] = {&b.X, &b.Y, &b.ZZZ, &b.ZZ};
                for (int k = 0; k < 4; k++)
                    if (!(k4[k]->bv >= 0 && k4[k]->bv <= INV[k][0] && k4[k]->bl >= 0 && k4[k]->bl <= INV[k][1])) sppark_bound_violation(op, k4[k]->bv, INV[k][0]);
            };
            auto madd = [&](g2c_bucket<F2>& b, const g2c_affine<F2>& p, bool neg, bool restart) {
                if (c.role == 0) b.template madd<0>(p, neg, restart, c); else b.template madd<1>(p, neg, restart, c);
            };
            for (int neg = 0; neg < 2; neg++) {
                g2c_bucket<F2> a; a.set_inf();
                madd(a, pt(c.lane), false, true); check("g2c madd (restart)", a);
                madd(a, pt(c.lane + 1), neg, false); check("g2c madd after a restart", a);
                loosen(a);
                madd(a, pt(c.lane + 2), neg, false); check("g2c madd from loose operands", a);
                g2c_bucket<F2> e; e.set_inf();
                madd(e, pt(c.lane), neg, true); loosen(e);
                madd(e, pt(c.lane), neg, false); check("g2c madd of the same point (cooperative doubling)", e);
                for (unsigned i = 3; i < 7; i++) { madd(a, pt(c.lane + i), (i + neg) & 1, false); check("g2c madd in a chain", a); }
            }
        });
        if (msg && msglen) { strncpy(msg, g_first, msglen - 1); msg[msglen - 1] = 0; }
        return g_violations.load();
    }
#endif
    (void)points; (void)stride; (void)n; (void)msg; (void)msglen;
    return -1;
}
#endif
