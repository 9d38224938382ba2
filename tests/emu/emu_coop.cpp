// HOST EMULATION TEST HARNESS for the cooperative point operations (tests only; see emu_msm.cpp).
//
// ec/xyzz_coop.hpp and msm/msm_coop_kernels.hpp are written for work-groups of four waves that meet at barriers.  Here a
// work-group is 256 HOST THREADS (role = tid / 64, lane = tid % 64) and the two barrier hooks of xyzz_coop.hpp are a
// counting barrier, so the product's own code -- level schedule, exchange-set parity, exceptional lanes, the tree, the
// record walk with its work-group vote -- runs unchanged in the GPU-less container and is compared with the serial
// formulas by tests/test_emulation.py.
#define SPPARK_HOST_EMULATION 1
#include "../../sppark_amd/csrc/msm/curve_select.hpp"
#include "../../sppark_amd/csrc/msm/msm_kernels.hpp"
#include "../../sppark_amd/csrc/msm/msm_coop_kernels.hpp"
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

using namespace sppark_amd;
typedef msm_fp_d F;
typedef xyzz_mem<F::N> mem_t;

namespace {
struct wg_barrier {
    std::mutex m; std::condition_variable cv;
    unsigned count = 0, gen = 0, n = COOP_NT;
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
// run |body(tid)| as one work-group of COOP_NT threads
void run_group(const std::function<void(unsigned)>& body)
{
    std::vector<std::thread> th;
    for (unsigned t = 0; t < COOP_NT; t++) th.emplace_back(body, t);
    for (auto& t : th) t.join();
}
} // namespace
extern "C" void sppark_emu_barrier() { (void)g_bar.wait(0); }
extern "C" int sppark_emu_barrier_or(int v) { return g_bar.wait(v); }

// op 0: (a + b) + b by two cooperative additions, op 1: 2(2a) by two cooperative doublings, op 2: the same with the
// serial formulas (add / dbl) -- internal XYZZ records in and out
extern "C" int emu_coop_ops(int op, void* out_, const void* a_, const void* b_, size_t n)
{
    mem_t* out = (mem_t*)out_; const mem_t* a = (const mem_t*)a_; const mem_t* b = (const mem_t*)b_;
    if (op >= 2) {
        for (size_t i = 0; i < n; i++) {
            xyzz_dev<F> p = xyzz_dev<F>::load(&a[i]);
            if (op == 2) { xyzz_dev<F> q = xyzz_dev<F>::load(&b[i]); p.add(q); p.add(q); } else { p.dbl(); p.dbl(); }
            p.store(&out[i]);
        }
        return 0;
    }
    static coop_lds<F> ex;
    for (size_t g0 = 0; g0 < n; g0 += 64) {
        run_group([&](unsigned tid) {
            const unsigned lane = tid & 63, role = tid >> 6; const size_t i = g0 + lane;
            xyzz_dev<F> p, q; p.set_inf(); q.set_inf();
            if (i < n) { p = xyzz_dev<F>::load(&a[i]); if (op == 0) q = xyzz_dev<F>::load(&b[i]); }
            coop_ctx<F> c{&ex, role, lane, 0};
            if (op == 0) { coop_add<F>(p, q, c); coop_add<F>(p, q, c); } else { coop_dbl<F>(p, c); coop_dbl<F>(p, c); }
            if (i < n && role == (i & 3)) p.store(&out[i]);
        });
    }
    return 0;
}

// out[0] = sum of pts[0 .. count) by coop_tree_sum (count a power of two <= 256), out[1] = the serial sum
extern "C" int emu_coop_tree(void* out_, const void* pts_, unsigned count)
{
    mem_t* out = (mem_t*)out_; const mem_t* pts = (const mem_t*)pts_;
    static coop_lds<F> ex; static coop_img<F, COOP_NT> img;
    run_group([&](unsigned tid) {
        xyzz_dev<F> v; v.set_inf();
        if (tid < count) v = xyzz_dev<F>::load(&pts[tid]);
        img.store(tid, v);
        coop_barrier();
        coop_ctx<F> c{&ex, tid >> 6, tid & 63, 0};
        coop_tree_sum<F, COOP_NT>(&img, count / 2, c);
        if (tid == 0) img.load(0).store(&out[0]);
    });
    xyzz_dev<F> s; s.set_inf();
    for (unsigned i = 0; i < count; i++) s.add(xyzz_dev<F>::load(&pts[i]));
    s.store(&out[1]);
    return 0;
}

// One level of the record tree over |nrec| (key, point) records with fan-in F: the cooperative walk (64 work items x
// four threads per work-group) and the one-thread-per-work-item walk of msm_kernels.hpp, each into its own outputs.
// buckets_*: nbuckets records, keys < nbuckets; out_key_* / out_pt_*: 2 * nthreads records.
extern "C" int emu_coop_reduce(const u32* in_key, const void* in_pt_, unsigned nrec, unsigned Fan, int last, unsigned nbuckets,
                               void* buckets_coop, u32* out_key_coop, void* out_pt_coop,
                               void* buckets_ref, u32* out_key_ref, void* out_pt_ref)
{
    const mem_t* in_pt = (const mem_t*)in_pt_;
    const unsigned nthreads = (nrec + Fan - 1) / Fan;
    (void)nbuckets;
    for (unsigned t = 0; t < nthreads; t++)
        reduce_runs_chunk<F>((mem_t*)buckets_ref, out_key_ref, (mem_t*)out_pt_ref, in_key, in_pt, nrec, Fan, nthreads, last, t);
    static coop_lds<F> ex;
    for (unsigned g0 = 0; g0 < nthreads; g0 += 64) {
        run_group([&](unsigned tid) {
            coop_ctx<F> c{&ex, tid >> 6, tid & 63, 0};
            reduce_runs_coop_item<F>((mem_t*)buckets_coop, out_key_coop, (mem_t*)out_pt_coop, in_key, in_pt, nrec, Fan, nthreads, last, g0 + c.lane, c);
        });
    }
    return 0;
}

// wire XYZZ (the reference's image, ec/xyzz_t.hpp:17) <-> the internal records these entry points take
extern "C" int emu_coop_from_std(void* out_, const void* in_, size_t n)
{
    constexpr int NW = fp_d::N;
    mem_t* out = (mem_t*)out_; const u32* in = (const u32*)in_;
    for (size_t i = 0; i < n; i++) {
        const u32* w = in + i * 4 * NW;
        xyzz_dev<F> r;
        bool inf = true;
        for (int k = 2 * NW; k < 4 * NW; k++) inf &= w[k] == 0;
        if (inf) r.set_inf();
        else { r.X = F::from_std(w); r.Y = F::from_std(w + NW); r.ZZZ = F::from_std(w + 2 * NW); r.ZZ = F::from_std(w + 3 * NW); }
        r.store(&out[i]);
    }
    return 0;
}
extern "C" int emu_coop_to_std(void* out_, const void* in_, size_t n)
{
    const mem_t* in = (const mem_t*)in_;
    for (size_t i = 0; i < n; i++) xyzz_dev<F>::load(&in[i]).store_std((xyzz_mem<fp_d::N>*)out_ + i);
    return 0;
}
extern "C" unsigned emu_coop_record_words() { return 4 * F::N; }

// number of i with a[i] != b[i] AS POINTS (XYZZ coordinates are projective: sums taken in different orders agree only
// up to scaling): X1 ZZ2 == X2 ZZ1 and Y1 ZZZ2 == Y2 ZZZ1, compared canonically
extern "C" int emu_coop_points_differ(const void* a_, const void* b_, size_t n)
{
    const mem_t* a = (const mem_t*)a_; const mem_t* b = (const mem_t*)b_;
    int bad = 0;
    for (size_t i = 0; i < n; i++) {
        const xyzz_dev<F> p = xyzz_dev<F>::load(&a[i]), q = xyzz_dev<F>::load(&b[i]);
        if (p.is_inf() || q.is_inf()) { bad += p.is_inf() != q.is_inf(); continue; }
        u32 l[4][fp_d::N];
        (p.X.norm() * q.ZZ).to_std(l[0]); (q.X.norm() * p.ZZ).to_std(l[1]);
        (p.Y.norm() * q.ZZZ).to_std(l[2]); (q.Y.norm() * p.ZZZ).to_std(l[3]);
        bad += memcmp(l[0], l[1], sizeof(l[0])) != 0 || memcmp(l[2], l[3], sizeof(l[2])) != 0;
    }
    return bad;
}

// ---- the contracts of the cooperative operations, machine-checked (-DSPPARK_TRACK_BOUNDS; see tests/emu/emu_bounds.cpp) -----
// coop_add / coop_dbl of ec/xyzz_coop.hpp from the LOOSEST operands the bucket invariants of ec/xyzzx_dev.hpp allow; the
// claims travel through the exchange area with the limbs.  Returns the number of violated contracts.
#ifdef SPPARK_TRACK_BOUNDS
static std::atomic<int> g_violations{0};
static std::mutex g_vm;
static char g_first[512];
extern "C" void sppark_bound_violation(const char* what, double got, double limit)
{
    std::lock_guard<std::mutex> lk(g_vm);
    if (!g_violations.load()) snprintf(g_first, sizeof(g_first), "%s: %.6g against %.6g", what, got, limit);
    g_violations++;
}
extern "C" int emu_coop_bounds(const unsigned char* points, size_t stride, size_t n, char* msg, size_t msglen)
{
    g_violations = 0; g_first[0] = 0;
    if (n < 4) return -1;
    static const double INV[4][2] = {{10, F::TIGHT ? 1.0 : 5.0}, {F::TIGHT ? 3.0 : 5.0, F::TIGHT ? 2.0 : 3.0}, {2, 1}, {2, 1}};      // X, Y, ZZZ, ZZ (ec/xyzzx_dev.hpp; TIGHT: 29-bit limbs)
    std::vector<uint4> conv((size_t)n * affine_loader<F>::STRIDE / 16 + 1);
    for (size_t i = 0; i < n; i++) affine_loader<F>::template convert<false>((unsigned char*)conv.data(), points, i, (unsigned)stride);
    auto pt = [&](size_t i) { affine_dev<F> p = load_affine<F, false>((const unsigned char*)conv.data(), i % n, 0); p.X.bnd_set(2, 1); p.Y.bnd_set(2, 1); return p; };
    auto loosen = [&](xyzz_dev<F>& b) { if (b.is_inf()) return; b.X.bnd_set(INV[0][0], INV[0][1]); b.Y.bnd_set(INV[1][0], INV[1][1]); b.ZZZ.bnd_set(INV[2][0], INV[2][1]); b.ZZ.bnd_set(INV[3][0], INV[3][1]); };
    auto check = [&](const char* op, const xyzz_dev<F>& b) {
        if (b.is_inf()) return;
        const F* c[4] = {&b.X, &b.Y, &b.ZZZ, &b.ZZ};
        for (int k = 0; k < 4; k++)
            if (!(c[k]->bv >= 0 && c[k]->bv <= INV[k][0] && c[k]->bl >= 0 && c[k]->bl <= INV[k][1])) sppark_bound_violation(op, c[k]->bv, INV[k][0]);
    };
    static coop_lds<F> ex;
    run_group([&](unsigned tid) {
        const unsigned lane = tid & 63, role = tid >> 6;
        coop_ctx<F> c{&ex, role, lane, 0};
        xyzz_dev<F> a; a.set(pt(lane), false); a.madd(pt(lane + 1), lane & 1); loosen(a);
        xyzz_dev<F> b; b.set(pt(lane + 2), true); b.madd(pt(lane + 3), !(lane & 1)); loosen(b);
        xyzz_dev<F> s = a; coop_add<F>(s, b, c); check("coop_add leaves a coordinate outside its invariant", s);
        xyzz_dev<F> d = a; coop_dbl<F>(d, c); check("coop_dbl leaves a coordinate outside its invariant", d);
        xyzz_dev<F> t = a; coop_add<F>(t, a, c); check("coop_add of equal operands leaves a coordinate outside its invariant", t);
        xyzz_dev<F> h = s; coop_add<F>(h, d, c); check("coop_add in a chain", h); coop_dbl<F>(h, c); check("coop_dbl in a chain", h);
    });
    if (msg && msglen) { strncpy(msg, g_first, msglen - 1); msg[msglen - 1] = 0; }
    return g_violations.load();
}
#endif
