// HOST EMULATION TEST HARNESS for the NTT kernels (tests only; see emu_msm.cpp).
// Runs the phase functions of sppark_amd/csrc/ntt/ntt_kernels.hpp for every
// (tile, thread) the GPU grid would cover, in the driver's launch order.
#define SPPARK_HOST_EMULATION 1
#include "../../sppark_amd/csrc/ff/params.hpp"
#include "../../sppark_amd/csrc/ntt/ntt_kernels.hpp"
#include "../../sppark_amd/csrc/ntt/ntt_r64_kernels.hpp"
#include "../../sppark_amd/csrc/ff/fr256_dev.hpp"
#include <vector>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
using namespace sppark_amd;

#if defined(FEATURE_GOLDILOCKS)
typedef gl64_dev F;
static const unsigned LG_LINE = 4, LG_TILE = 12;
static F finv(F a) { return field_pow(a, (u64)F::MOD - 2); }
static F top_root() { return F::top_root(); }
static F group_gen() { return F::group_gen(); }
#elif defined(FEATURE_BABY_BEAR)
typedef bb31_dev F;
static const unsigned LG_LINE = 5, LG_TILE = 13;
static F finv(F a) { return field_pow(a, (u64)F::MOD - 2); }
static F top_root() { return F::top_root(); }
static F group_gen() { return F::group_gen(); }
#else
# if defined(FEATURE_BLS12_381)
typedef bls12_381_fr_p FRP;
# else
typedef alt_bn128_fr_p FRP;
# endif
typedef fr256_dev<FRP> F;
static const unsigned LG_LINE = 4, LG_TILE = 10;
static F pow_big(F b, const uint64_t* e, unsigned from_bit)
{
    F r = F::one();
    for (unsigned bit = from_bit; bit < 256; bit++) { if ((e[bit / 64] >> (bit % 64)) & 1) r = r * b; b = b * b; }
    return r;
}
static F finv(F a) { uint64_t e[4]; for (int i = 0; i < 4; i++) e[i] = FRP::MOD64[i]; e[0] -= 2; return pow_big(a, e, 0); }
static F group_gen() { F g = F::one(); F one = F::one(); for (unsigned i = 1; i < FRP::GROUP_GEN; i++) g = g + one; return g; }
static F top_root() { uint64_t e[4]; for (int i = 0; i < 4; i++) e[i] = FRP::MOD64[i]; e[0] -= 1; return pow_big(group_gen(), e, FRP::TWO_ADICITY); }
#endif

// ntt_engine::bit_reverse with the tile kernel's two phases run on the host
static void emu_bitrev(F* d, unsigned lg)
{
    constexpr unsigned TB = bitrev_tile_bits<F>::value;
    const size_t n = (size_t)1 << lg;
    if (lg >= 2 * TB + 1) {
        std::vector<F> lds(2 * (((size_t)(1u << TB) + 1) << TB));
        F* A = lds.data(); F* B = A + (((1u << TB) + 1) << TB);
        for (size_t mid = 0; mid < (n >> (2 * TB)); mid++)
            for (int phase = 0; phase < 2; phase++)
                for (unsigned tid = 0; tid < 256; tid++) {
                    // as ntt_engine::bit_reverse: the 16-byte form for single-word fields in 16-byte aligned buffers
                    if constexpr (sizeof(F) <= 8) {
                        if (((uintptr_t)d & 15) == 0) { bitrev_tile_vec_item<F, TB>(d, A, B, lg, mid, tid, phase); continue; }
                    }
                    bitrev_tile_item<F, TB>(d, A, B, lg, mid, tid, 256, phase);
                }
    } else {
        for (size_t i = 0; i < n; i++) bitrev_item(d, lg, i);
    }
}

// ntt_engine::run()'s |lde| argument: the compact coefficients the first k_ntt12 step of sppark_lde's forward transform reads
// (ntt_r64_args::lde_*), set by emu_lde around its emu_ntt call
struct emu_lde_in { const void* src; const void* glo; const void* ghi; unsigned gh, lgd, lgb; };
static const emu_lde_in* g_lde_in = nullptr;
static void* g_final_out = nullptr;     // ntt_r64_args::out of the last k_ntt12 step (DIF), consumed (set to null) by the step that stores there

// the radix-64 plan of ntt_engine::run (single-word fields, lg >= 12): k_ntt6 / k_ntt12 round by round,
// generic passes for the other steps; tables from r64_table_item, as ntt_engine::r64_table builds them
template<class FF> static void emu_r64_passes(FF* d, unsigned lg, bool gs, int inverse, ntt_tables<FF>& T, unsigned nt, unsigned direct_max,
                                              unsigned cmode, const ntt_tables<FF>& G)
{
    if constexpr (sizeof(FF) <= 8) {
        const size_t n = (size_t)1 << lg;
        r64_plan rp = make_r64_plan(lg);
        std::vector<FF> cz(64);
        for (unsigned x = 0; x < 64 && cmode; x++) r64_cz_item(cz.data(), G, cmode, x);
        for (unsigned i = 0; i < rp.nsteps; i++) {
            const r64_step st = rp.step[gs ? i : rp.nsteps - 1 - i];
            const bool last = i == rp.nsteps - 1;
            if (st.kind == 0) {
                ntt_pass P; P.lg_cur = st.lg_cur; P.S = st.S; P.lgC = LG_LINE; P.lgG = 0; P.apply_scale = inverse && last;
                // as ntt_engine::run(): the generic top pass's share of a folded coset transform
                std::vector<FF> crow((size_t)1 << st.S);
                for (unsigned r = 0; r < crow.size() && cmode; r++) pass_crow_item(crow.data(), G, cmode, st.S, r);
                P.cmode = cmode; P.crow = crow.data(); P.cg_lo = G.lo; P.cg_hi = G.hi; P.cg_h = G.h;
                std::vector<FF> tile(ntt_lds_elems(P) + 1);
                T.pass_tw = nullptr;
                const size_t tile_elems = (size_t)1 << (P.S + P.lgC);
                for (size_t tile_id = 0; tile_id < n / tile_elems; tile_id++) {
#define EMU_ROUNDS(R1, R2)                                                                                         \
                    do {                                                                                               \
                        if (gs) {                                                                                      \
                            for (unsigned tid = 0; tid < nt; tid++) { if (inverse) ntt_round_high<FF, true, true, R1, R2>(d, tile.data(), T, P, tile_id, tid, nt); else ntt_round_high<FF, true, false, R1, R2>(d, tile.data(), T, P, tile_id, tid, nt); } \
                            if (R2) for (unsigned tid = 0; tid < nt; tid++) { if (inverse) ntt_round_low<FF, true, true, R1, R2>(d, tile.data(), T, P, tile_id, tid, nt); else ntt_round_low<FF, true, false, R1, R2>(d, tile.data(), T, P, tile_id, tid, nt); } \
                        } else {                                                                                       \
                            if (R2) for (unsigned tid = 0; tid < nt; tid++) { if (inverse) ntt_round_low<FF, false, true, R1, R2>(d, tile.data(), T, P, tile_id, tid, nt); else ntt_round_low<FF, false, false, R1, R2>(d, tile.data(), T, P, tile_id, tid, nt); } \
                            for (unsigned tid = 0; tid < nt; tid++) { if (inverse) ntt_round_high<FF, false, true, R1, R2>(d, tile.data(), T, P, tile_id, tid, nt); else ntt_round_high<FF, false, false, R1, R2>(d, tile.data(), T, P, tile_id, tid, nt); } \
                        }                                                                                              \
                    } while (0)
                    SPPARK_NTT_DISPATCH_S(P.S, EMU_ROUNDS);
#undef EMU_ROUNDS
                }
                continue;
            }
            const int scaled = inverse && last;
            std::vector<FF> tw, t1, t2, tile(4096);
            ntt_r64_args<FF> A{T.inner, nullptr, nullptr, nullptr, st.lg_cur, nullptr};
            // as ntt_engine::run(): the coset powers in the tables of the step on the whole transform (natural exponents) / of
            // every step (bit-reversed exponents), the 64 constants in that step / in k_ntt12
            const unsigned cm = cmode == 1 && st.lg_cur != lg ? 0 : cmode;
            if (cmode == 2 ? st.kind == 2 : (cmode == 1 && st.lg_cur == lg)) A.cz = cz.data();
            auto fill = [&](std::vector<FF>& v, unsigned kind, int sc) {
                v.resize(kind == 0 ? (size_t)1 << st.lg_cur : kind == 1 ? (size_t)1 << (st.lg_cur - 6) : 4096);
                for (size_t k = 0; k < v.size(); k++) r64_table_item(v.data(), T, kind, st.lg_cur, sc, k, cm == 2 && kind == 2 ? 0u : cm, &G);
            };
            if (st.kind == 2 || st.lg_cur <= direct_max) { fill(tw, 0, scaled); A.tw = tw.data(); }
            else { fill(t1, 1, scaled); fill(t2, 2, 0); A.t1 = t1.data(); A.t2 = t2.data(); }
            for (size_t tile_id = 0; tile_id < (n >> 12); tile_id++) {
#define EMU_ALL(CALL) for (unsigned tid = 0; tid < 512; tid++) { CALL; }
                if (st.kind == 1) {
                    if (gs) { if (inverse) { EMU_ALL((ntt6_high<FF, true, true>(d, tile.data(), A, tile_id, tid))) EMU_ALL((ntt6_low<FF, true, true>(d, tile.data(), A, tile_id, tid))) }
                              else         { EMU_ALL((ntt6_high<FF, true, false>(d, tile.data(), A, tile_id, tid))) EMU_ALL((ntt6_low<FF, true, false>(d, tile.data(), A, tile_id, tid))) } }
                    else    { if (inverse) { EMU_ALL((ntt6_low<FF, false, true>(d, tile.data(), A, tile_id, tid))) EMU_ALL((ntt6_high<FF, false, true>(d, tile.data(), A, tile_id, tid))) }
                              else         { EMU_ALL((ntt6_low<FF, false, false>(d, tile.data(), A, tile_id, tid))) EMU_ALL((ntt6_high<FF, false, false>(d, tile.data(), A, tile_id, tid))) } }
                } else {
                    FF* sub = d + (tile_id << 12);
#define EMU_R12(DIF, INV, D) EMU_ALL((ntt12_round<FF, DIF, INV, D>(sub, tile.data(), A, tid)))
                    if (g_final_out && last && gs && inverse) {     // as k_ntt12<F, true, true> with ntt_r64_args::out
                        FF* osub = (FF*)g_final_out + (tile_id << 12);
                        EMU_R12(true, true, R12_A1) EMU_R12(true, true, R12_B1) EMU_R12(true, true, R12_A2)
                        EMU_ALL((ntt12_round<FF, true, true, R12_B2>(sub, tile.data(), A, tid, 0, osub)))
                        if (tile_id + 1 == (n >> 12)) g_final_out = nullptr;
                    } else
                    if (g_lde_in && i == 0 && !gs && !inverse) {    // as k_ntt12<F, false, false, true>
                        A.lde_src = (const FF*)g_lde_in->src; A.lde_glo = (const FF*)g_lde_in->glo; A.lde_ghi = (const FF*)g_lde_in->ghi;
                        A.lde_gh = g_lde_in->gh; A.lde_lgd = g_lde_in->lgd; A.lde_lgb = g_lde_in->lgb;
                        EMU_ALL((ntt12_round<FF, false, false, R12_B2, true>(sub, tile.data(), A, tid, tile_id)))
                        EMU_R12(false, false, R12_A2) EMU_R12(false, false, R12_B1) EMU_R12(false, false, R12_A1)
                    } else
                    if (gs) { if (inverse) { EMU_R12(true, true, R12_A1) EMU_R12(true, true, R12_B1) EMU_R12(true, true, R12_A2) EMU_R12(true, true, R12_B2) }
                              else         { EMU_R12(true, false, R12_A1) EMU_R12(true, false, R12_B1) EMU_R12(true, false, R12_A2) EMU_R12(true, false, R12_B2) } }
                    else    { if (inverse) { EMU_R12(false, true, R12_B2) EMU_R12(false, true, R12_A2) EMU_R12(false, true, R12_B1) EMU_R12(false, true, R12_A1) }
                              else         { EMU_R12(false, false, R12_B2) EMU_R12(false, false, R12_A2) EMU_R12(false, false, R12_B1) EMU_R12(false, false, R12_A1) } }
#undef EMU_R12
                }
#undef EMU_ALL
            }
        }
    }
}

static unsigned g_r64_min = 12, g_r64_direct = 20;          // as ntt_engine's knobs (SPPARK_NTT_R64_MIN / _DIRECT)
extern "C" void emu_ntt_plan(unsigned r64_min, unsigned r64_direct) { g_r64_min = r64_min; g_r64_direct = r64_direct; }
static unsigned g_coset_fold = 1;                           // as ntt_engine's knob (SPPARK_NTT_COSET_FOLD): 0 = the separate coset_item pass
extern "C" void emu_ntt_coset_fold(unsigned on) { g_coset_fold = on; }
// how a coset transform of 2^lg elements runs (r64_coset_mode): 0 separate scaling pass, 1 / 2 folded
extern "C" unsigned emu_ntt_coset_mode(unsigned lg, int order, int direction)
{
    if (sizeof(F) > 8 || lg < 12 || lg < g_r64_min || !g_coset_fold) return 0;
    return r64_coset_mode(make_r64_plan(lg), order == 1 || order == 3, direction == 1, order != 3);
}
// the one-stage-per-round passes of the 256-bit fields (k_ntt_pass_lat): stages per pass (0: the register passes),
// log2 columns per tile row, log2 tile elements -- as ntt_engine's lat_* choices
static unsigned g_lat_smax = sizeof(F) > 8 ? 8 : 0;         // as ntt_engine<F>::LAT_SMAX
static int g_lat_lgc = -1, g_lat_lgtile = -1;               // -1: the shape by size (make_ntt_lat_plan)
// the plan of the one-stage-per-round passes for a size, as the engine makes it: out[4 * i ..] = {lg_cur, S, lgC, lgG} of
// pass i (GS order); returns the number of passes
extern "C" unsigned emu_ntt_lat_plan(unsigned lg, unsigned smax, int lgc, int lgtile, unsigned out[64])
{
    const ntt_plan pl = make_ntt_lat_plan(lg, smax, lgc, lgtile);
    for (unsigned i = 0; i < pl.npass && i < 16; i++) {
        out[4 * i] = pl.pass[i].lg_cur; out[4 * i + 1] = pl.pass[i].S; out[4 * i + 2] = pl.pass[i].lgC; out[4 * i + 3] = pl.pass[i].lgG;
    }
    return pl.npass;
}
extern "C" void emu_ntt_lat(unsigned smax, int lgc, int lgtile) { g_lat_smax = smax; g_lat_lgc = lgc; g_lat_lgtile = lgtile; }

// k_ntt_small keeps a butterfly pair per lane and swaps values between lanes (ntt_rx_xchg): here a work-group is |n|
// HOST THREADS, every exchange goes through the emulated LDS buffers, and the barrier hook of ntt_kernels.hpp is a counting
// barrier -- the product's own index math, twiddle choice, regrouping and permutations run unchanged (only ds_bpermute,
// the within-wave form of the exchange, is hardware-only: the GPU tests cover it).
namespace {
struct wg_barrier {
    std::mutex m; std::condition_variable cv;
    unsigned count = 0, gen = 0, n = 0;
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        const unsigned g = gen;
        if (++count == n) { count = 0; gen++; cv.notify_all(); return; }
        cv.wait(lk, [&] { return gen != g; });
    }
} g_bar;
void run_group(unsigned n, const std::function<void(unsigned)>& body)
{
    g_bar.n = n; g_bar.count = 0;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < n; t++) th.emplace_back(body, t);
    for (auto& t : th) t.join();
}
} // namespace
extern "C" void sppark_emu_barrier() { g_bar.wait(); }

// transforms up to this size as ONE work-group (k_ntt_small); as ntt_engine::small_max_lg(); 0 = the general path
static unsigned g_small_max = sizeof(F) > 8 ? ntt_small_cap<F>::value - 1 : ntt_small_cap<F>::value;
static unsigned g_small_sized = 1;                          // as ntt_engine::small_sized()
extern "C" void emu_ntt_small(unsigned max_lg) { g_small_max = max_lg; }
extern "C" void emu_ntt_small_sized(unsigned on) { g_small_sized = on; }      // 0: the run-time-size instance at every size

extern "C" int emu_ntt(void* inout, unsigned lg, int order, int direction, int type, unsigned nt)
{
    if (lg == 0) return 0;
    F* d = (F*)inout;
    const size_t n = (size_t)1 << lg;
    const int inverse = direction == 1;
    unsigned h = lg < 12 ? lg : 12;
    std::vector<F> lo(1u << h), hi((size_t)1 << (lg - h)), glo(1u << h), ghi((size_t)1 << (lg - h)), inner(ntt_inner_entries<F>::value);
    F w = top_root();
    for (unsigned k = F::TWO_ADICITY; k > lg; k--) w = w * w;
    F g = group_gen();
    if (inverse) { w = finv(w); g = finv(g); }
    for (size_t k = 0; k < std::max<size_t>(std::max(lo.size(), hi.size()), inner.size()); k++) {
        table_item(lo.data(), hi.data(), inner.data(), w, lg, h, k);
        table_item(glo.data(), ghi.data(), (F*)nullptr, g, lg, h, k);
    }
    F two = F::one() + F::one();
    ntt_tables<F> T{lo.data(), hi.data(), inner.data(), lg, h, finv(field_pow(two, lg))}, G{glo.data(), ghi.data(), nullptr, lg, h, F::one()};

    if (lg <= g_small_max) {                                    // k_ntt_small: its lanes as host threads (barrier hook below)
        const unsigned flags = ntt_small_flags(order, inverse != 0, type == 1);
        const unsigned lanes = (unsigned)std::max<size_t>(64, n / 2);
        std::vector<F> lds(2 * (size_t)lanes + 1);
        // as ntt_engine::run(): the single-word fields have an instance compiled for each of the sizes 2^8 ... 2^11
        auto lane = [&](unsigned tid, auto LGC) {
            constexpr unsigned C = sizeof(F) <= 8 ? decltype(LGC)::value : 0;      // (256-bit fields: the run-time form only)
            if (flags & NTT_SMALL_GS) { if (inverse) ntt_rx_run<F, true, true, C>(d, lds.data(), T, G, flags, tid, lanes); else ntt_rx_run<F, false, true, C>(d, lds.data(), T, G, flags, tid, lanes); }
            else                      { if (inverse) ntt_rx_run<F, true, false, C>(d, lds.data(), T, G, flags, tid, lanes); else ntt_rx_run<F, false, false, C>(d, lds.data(), T, G, flags, tid, lanes); }
        };
        const unsigned sized = (sizeof(F) <= 8 && g_small_sized) ? lg : 0;
        run_group(lanes, [&](unsigned tid) {
            switch (sized) {
                case 8:  lane(tid, std::integral_constant<unsigned, 8>()); break;
                case 9:  lane(tid, std::integral_constant<unsigned, 9>()); break;
                case 10: lane(tid, std::integral_constant<unsigned, 10>()); break;
                case 11: lane(tid, std::integral_constant<unsigned, 11>()); break;
                default: lane(tid, std::integral_constant<unsigned, 0>()); break;
            }
        });
        return 0;
    }

    bool bitrev, gs;
    switch (order) {
        case 0: emu_bitrev(d, lg); bitrev = true; gs = false; break;
        case 1: bitrev = false; gs = true; break;
        case 2: bitrev = true; gs = false; break;
        default: bitrev = true; gs = true; break;
    }
    // as ntt_engine::run(): a coset transform on a plan of k_ntt6 / k_ntt12 steps is folded into the passes
    unsigned cmode = 0;
    if (sizeof(F) <= 8 && lg >= 12 && lg >= g_r64_min && g_coset_fold)
        cmode = r64_coset_mode(make_r64_plan(lg), gs, inverse != 0, type == 1 && order != 3);
    if (!inverse && type == 1 && !cmode) for (size_t i = 0; i < n; i++) coset_item(d, G, (int)bitrev, i);

#ifndef EMU_SMAX
#define EMU_SMAX (sizeof(F) > 8 ? 4 : 8)        // as ntt_engine<F>::S_MAX
#endif
    const bool lat = g_lat_smax != 0;
    ntt_plan pl = lat ? make_ntt_lat_plan(lg, g_lat_smax, g_lat_lgc, g_lat_lgtile) : make_ntt_plan(lg, LG_LINE, LG_TILE, EMU_SMAX);
    if (sizeof(F) <= 8 && lg >= 12 && lg >= g_r64_min) {
        emu_r64_passes<F>(d, lg, gs, inverse, T, nt, g_r64_direct, cmode, G);
        pl.npass = 0;
    }
    // as ntt_engine::has_pass_table(): passes on sub-problems of <= 2^16 (256-bit fields: 2^24) elements read their
    // inter-pass twiddles from a table; as ntt_engine::run(): the table of the smallest tabled pass carries 1/n
    auto has_table = [](const ntt_pass& q) { return !(q.lg_cur > (ntt_gen_twiddles<F>::value ? 16u : 24u) || q.lg_cur <= q.S || q.S / 2 == 0); };
    int scale_pass = -1;
    if (inverse) {
        unsigned best = ~0u;
        for (unsigned i = 0; i < pl.npass; i++) {
            const ntt_pass& q = pl.pass[gs ? i : pl.npass - 1 - i];
            if (has_table(q) && q.lg_cur < best) { best = q.lg_cur; scale_pass = (int)i; }
        }
    }
    for (unsigned i = 0; i < pl.npass; i++) {
        ntt_pass P = pl.pass[gs ? i : pl.npass - 1 - i];
        P.apply_scale = inverse && i == pl.npass - 1 && scale_pass < 0;
        size_t tile_elems = (size_t)1 << (P.lgG + P.S + P.lgC);
        std::vector<F> tile(ntt_lds_elems(P) + 1);
        std::vector<F> pass_tw;
        T.pass_tw = nullptr;
        if (has_table(P)) {
            pass_tw.resize((size_t)1 << P.lg_cur);
            for (size_t k = 0; k < pass_tw.size(); k++) pass_table_item(pass_tw.data(), T, P.lg_cur, P.S, k, inverse && scale_pass == (int)i);
            T.pass_tw = pass_tw.data();
        }
        if (lat) {
            tile.resize(tile_elems);
#define EMU_LAT(DIF, INV)                                                                                          \
            for (size_t tile_id = 0; tile_id < n / tile_elems; tile_id++) {                                        \
                for (unsigned tid = 0; tid < nt; tid++) ntt_lat_load<F, DIF>(d, tile.data(), T, P, tile_id, tid, nt); \
                for (unsigned t = 0; t < P.S; t++)                                                                 \
                    for (unsigned tid = 0; tid < nt; tid++) ntt_lat_stage<F, DIF, INV>(tile.data(), T, P, t, tid, nt); \
                for (unsigned tid = 0; tid < nt; tid++) ntt_lat_store<F, DIF>(d, tile.data(), T, P, tile_id, tid, nt); \
            }
            if (gs) { if (inverse) { EMU_LAT(true, true) } else { EMU_LAT(true, false) } }
            else    { if (inverse) { EMU_LAT(false, true) } else { EMU_LAT(false, false) } }
#undef EMU_LAT
            continue;
        }
        for (size_t tile_id = 0; tile_id < n / tile_elems; tile_id++) {
#define EMU_ROUNDS(R1, R2)                                                                                         \
            do {                                                                                                   \
                if (gs) {                                                                                          \
                    for (unsigned tid = 0; tid < nt; tid++) { if (inverse) ntt_round_high<F, true, true, R1, R2>(d, tile.data(), T, P, tile_id, tid, nt); else ntt_round_high<F, true, false, R1, R2>(d, tile.data(), T, P, tile_id, tid, nt); } \
                    if (R2) for (unsigned tid = 0; tid < nt; tid++) { if (inverse) ntt_round_low<F, true, true, R1, R2>(d, tile.data(), T, P, tile_id, tid, nt); else ntt_round_low<F, true, false, R1, R2>(d, tile.data(), T, P, tile_id, tid, nt); } \
                } else {                                                                                           \
                    if (R2) for (unsigned tid = 0; tid < nt; tid++) { if (inverse) ntt_round_low<F, false, true, R1, R2>(d, tile.data(), T, P, tile_id, tid, nt); else ntt_round_low<F, false, false, R1, R2>(d, tile.data(), T, P, tile_id, tid, nt); } \
                    for (unsigned tid = 0; tid < nt; tid++) { if (inverse) ntt_round_high<F, false, true, R1, R2>(d, tile.data(), T, P, tile_id, tid, nt); else ntt_round_high<F, false, false, R1, R2>(d, tile.data(), T, P, tile_id, tid, nt); } \
                }                                                                                                  \
            } while (0)
            SPPARK_NTT_DISPATCH_S(P.S, EMU_ROUNDS);
#undef EMU_ROUNDS
        }
    }
    if (inverse && type == 1 && !cmode) for (size_t i = 0; i < n; i++) coset_item(d, G, (int)!bitrev, i);
    if (order == 3) emu_bitrev(d, lg);
    return 0;
}

// ntt_engine::lde() with the kernel bodies run on the host: iNTT(NR) of a copy,
// lde_spread_item over every output element, NTT(RN) of the extended buffer.
extern "C" int emu_lde(void* inout, unsigned lg_domain, unsigned lg_blowup, void* aux)
{
    F* ext = (F*)inout;
    const size_t dom = (size_t)1 << lg_domain, n_ext = dom << lg_blowup;
    // as ntt_engine::lde(): the inverse transform in place in ext[0 .. dom), its last step (where that is k_ntt12) storing to tmp
    std::vector<F> tmp(dom);
    g_final_out = tmp.data();
    emu_ntt(ext, lg_domain, 1, 1, 0, 64);
    if (g_final_out) { memcpy((void*)tmp.data(), (const void*)ext, dom * sizeof(F)); g_final_out = nullptr; }
    if (aux) {
        for (size_t i = 0; i < dom; i++) {
            size_t r = 0;
            for (unsigned k = 0; k < lg_domain; k++) r |= ((i >> k) & 1) << (lg_domain - 1 - k);
            ((F*)aux)[r] = tmp[i];
        }
    }
    unsigned h = lg_domain < 12 ? lg_domain : 12;
    std::vector<F> glo(1u << h), ghi((size_t)1 << (lg_domain - h));
    for (size_t k = 0; k < std::max(glo.size(), ghi.size()); k++)
        table_item(glo.data(), ghi.data(), (F*)nullptr, group_gen(), lg_domain, h, k);
    ntt_tables<F> G{glo.data(), ghi.data(), nullptr, lg_domain, h, F::one()};
    // as ntt_engine::run(.., lde): the spread + coset shift inside the first step of the radix-64 plan (the extended buffer is
    // then never read: filled with a pattern here), as a pass of its own otherwise
    const unsigned lg_ext = lg_domain + lg_blowup;
    if (sizeof(F) <= 8 && lg_ext >= 12 && lg_ext >= g_r64_min && lg_blowup >= 1 && lg_blowup <= 3
        && make_r64_plan(lg_ext).step[make_r64_plan(lg_ext).nsteps - 1].kind == 2) {
        memset((void*)ext, 0xA5, n_ext * sizeof(F));
        const emu_lde_in li{tmp.data(), glo.data(), ghi.data(), h, lg_domain, lg_blowup};
        g_lde_in = &li;
        emu_ntt(ext, lg_ext, 2, 0, 0, 64);
        g_lde_in = nullptr;
        return 1;
    }
    for (size_t o = 0; o < n_ext; o++) lde_spread_item(ext, tmp.data(), G, lg_domain, lg_blowup, 1, o);
    emu_ntt(ext, lg_ext, 2, 0, 0, 64);
    return 0;
}
