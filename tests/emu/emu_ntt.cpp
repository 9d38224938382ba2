// HOST EMULATION TEST HARNESS for the NTT kernels (tests only; see emu_msm.cpp).
// Runs the phase functions of sppark_amd/csrc/ntt/ntt_kernels.hpp for every
// (tile, thread) the GPU grid would cover, in the driver's launch order.
#define SPPARK_HOST_EMULATION 1
#include "../../sppark_amd/csrc/ff/params.hpp"
#include "../../sppark_amd/csrc/ntt/ntt_kernels.hpp"
#include "../../sppark_amd/csrc/ff/fr256_dev.hpp"
#include <vector>
#include <cstring>
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
                for (unsigned tid = 0; tid < 256; tid++) bitrev_tile_item<F, TB>(d, A, B, lg, mid, tid, 256, phase);
    } else {
        for (size_t i = 0; i < n; i++) bitrev_item(d, lg, i);
    }
}

extern "C" int emu_ntt(void* inout, unsigned lg, int order, int direction, int type, unsigned nt)
{
    if (lg == 0) return 0;
    F* d = (F*)inout;
    const size_t n = (size_t)1 << lg;
    const int inverse = direction == 1;
    unsigned h = lg < 12 ? lg : 12;
    std::vector<F> lo(1u << h), hi((size_t)1 << (lg - h)), glo(1u << h), ghi((size_t)1 << (lg - h)), inner(512);
    F w = top_root();
    for (unsigned k = F::TWO_ADICITY; k > lg; k--) w = w * w;
    F g = group_gen();
    if (inverse) { w = finv(w); g = finv(g); }
    for (size_t k = 0; k < std::max<size_t>(std::max(lo.size(), hi.size()), 512); k++) {
        table_item(lo.data(), hi.data(), inner.data(), w, lg, h, k);
        table_item(glo.data(), ghi.data(), (F*)nullptr, g, lg, h, k);
    }
    F two = F::one() + F::one();
    ntt_tables<F> T{lo.data(), hi.data(), inner.data(), lg, h, finv(field_pow(two, lg))}, G{glo.data(), ghi.data(), nullptr, lg, h, F::one()};

    bool bitrev, gs;
    switch (order) {
        case 0: emu_bitrev(d, lg); bitrev = true; gs = false; break;
        case 1: bitrev = false; gs = true; break;
        case 2: bitrev = true; gs = false; break;
        default: bitrev = true; gs = true; break;
    }
    if (!inverse && type == 1) for (size_t i = 0; i < n; i++) coset_item(d, G, (int)bitrev, i);

#ifndef EMU_SMAX
#define EMU_SMAX (sizeof(F) > 8 ? 4 : 8)        // as ntt_engine<F>::S_MAX
#endif
    ntt_plan pl = make_ntt_plan(lg, LG_LINE, LG_TILE, EMU_SMAX);
    for (unsigned i = 0; i < pl.npass; i++) {
        ntt_pass P = pl.pass[gs ? i : pl.npass - 1 - i];
        P.apply_scale = inverse && i == pl.npass - 1;
        size_t tile_elems = (size_t)1 << (P.lgG + P.S + P.lgC);
        std::vector<F> tile(ntt_lds_elems(P) + 1);
        // as ntt_engine::pass_table(): passes on sub-problems of <= 2^16 elements read their inter-pass
        // twiddles from a table instead of generating them
        std::vector<F> pass_tw;
        T.pass_tw = nullptr;
        if (ntt_gen_twiddles<F>::value && P.lg_cur <= 16 && P.lg_cur > P.S && P.S / 2 != 0) {
            pass_tw.resize((size_t)1 << P.lg_cur);
            for (size_t k = 0; k < pass_tw.size(); k++) pass_table_item(pass_tw.data(), T, P.lg_cur, P.S, k);
            T.pass_tw = pass_tw.data();
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
    if (inverse && type == 1) for (size_t i = 0; i < n; i++) coset_item(d, G, (int)!bitrev, i);
    if (order == 3) emu_bitrev(d, lg);
    return 0;
}

// ntt_engine::lde() with the kernel bodies run on the host: iNTT(NR) of a copy,
// lde_spread_item over every output element, NTT(RN) of the extended buffer.
extern "C" int emu_lde(void* inout, unsigned lg_domain, unsigned lg_blowup, void* aux)
{
    F* ext = (F*)inout;
    const size_t dom = (size_t)1 << lg_domain, n_ext = dom << lg_blowup;
    std::vector<F> tmp(ext, ext + dom);
    emu_ntt(tmp.data(), lg_domain, 1, 1, 0, 64);
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
    for (size_t o = 0; o < n_ext; o++) lde_spread_item(ext, tmp.data(), G, lg_domain, lg_blowup, 1, o);
    emu_ntt(ext, lg_domain + lg_blowup, 2, 0, 0, 64);
    return 0;
}
