// MI355X NTT kernels for single-word fields (Goldilocks, BabyBear).
//
// Same observable semantics as the reference's driver (ntt/ntt.cuh:161-213:
// NN = bit_rev + CT, NR = GS, RN = CT, RR = GS + bit_rev; coset scaling of
// ntt/kernels.cu:131-153) but a different decomposition.  The transform is cut
// into PASSES of S <= 8 stages; a pass is a batch of independent 2^S-point
// transforms on a tile of [2^S rows] x [C adjacent columns] (C*sizeof(F) = one
// 128-byte line, so strided passes stay coalesced) followed (GS) or preceded
// (CT) by ONE diagonal twiddle per element:
//
//   GS/DIF pass on a sub-problem of size n_cur = 2^S * Q, element (mid, lo):
//        y[mid][lo] = DIF_{2^S}(x[.][lo])[mid] * w_{n_cur}^(lo * rev_S(mid))
//   the next pass works on the Q-sized rows independently.  CT/DIT is the
//   transposed network: passes in reverse order, twiddle first.
//
// Inside a pass the 2^S-point transform is itself split 2^R1 x 2^R2 and done
// in REGISTERS: each lane loads 2^R1 (<= 16) elements straight from HBM, runs
// the R1 butterfly stages on them, applies the w_{2^S} twiddle, exchanges once
// through LDS, runs the R2 stages, applies the inter-pass twiddle and stores
// straight back to HBM -- one LDS write + one LDS read per element per pass
// (the reference's narrow kernels, ntt/kernels/ct_mixed_radix_narrow.cu:98-154,
// do 5 shuffle stages + shared-memory stages on 32-lane warps).
// For Goldilocks the in-register butterflies need no multiplier at all: every
// root of order <= 64 is a power of two (gl64_dev::mul_pow2).
//
// Twiddles w_n^e come from two tables of <= 2^12 entries (w^e_lo, w^(e_hi<<h))
// instead of the reference's four 7-bit windows (ntt/parameters.cuh:72-145).
//
// Each round is a SPPARK_DEVFN function of (tid, nthreads) so that the host
// emulation harness (tests/emu) runs the same index math on the CPU.
#pragma once
#include "../ff/small_fields_dev.hpp"
#include <type_traits>

namespace sppark_amd {

// what k_ntt_small (whole transforms by one work-group, below) is compiled for: 2^11 elements of a single-word field (1024
// lanes, eleven twiddles in registers), 2^10 of a 256-bit one (512 lanes).  What the driver USES by default is
// ntt_engine::small_max_lg().
template<class F> struct ntt_small_cap { static constexpr unsigned value = sizeof(F) > 8 ? 10 : 11; };
// entries of ntt_tables::inner: the levels R <= 8 of the register radices and, for k_ntt_small, R <= its cap
template<class F> struct ntt_inner_entries { static constexpr unsigned value = 2u << (ntt_small_cap<F>::value > 8 ? ntt_small_cap<F>::value : 8); };

template<class F> struct ntt_tables {
    const F* lo;        // w^k,            k < 2^h
    const F* hi;        // w^(k << h),     k < 2^(lg_n - h)
    const F* inner;     // inner[(1 << R) + k] = w_{2^R}^k, R <= min(lg_n, cap), k < 2^R: every level in consecutive entries
    unsigned lg_n, h;
    F scale;            // 1/n for the inverse transform (Montgomery form where applicable)
    // Inter-pass twiddles of THIS pass as a table, pass_tw[(mid << lgQ) + col] = w_{n_cur}^(col * rev_S(mid)),
    // or null.  A pass on sub-problems of n_cur <= 2^16 elements has at most 2^16 distinct twiddles,
    // shared by all its 2^(lg_n - lg_cur) sub-problems: one L2-resident table read in whole rows
    // (16 coalesced loads per work item) replaces their generation (2 look-ups + 18 products).
    const F* pass_tw;
};

struct ntt_pass {
    unsigned lg_cur;    // log2 of the sub-problem size this pass splits
    unsigned S;         // stages in this pass (<= 8)
    unsigned lgC;       // log2 columns (lo) per tile
    unsigned lgG;       // log2 sub-problems per tile (only when the tile spans all Q columns)
    int apply_scale;    // multiply by tables.scale when storing (inverse, last executed pass)
    // A coset transform folded into the pass on the WHOLE transform when that is a generic pass above the radix-64 steps
    // (ntt_r64_kernels.hpp r64_coset_mode; 0 = none).  cmode 1, natural exponents: crow[row] = g^(row << lgQ) on the data
    // side (DIF load / DIT store) and g^col, from the powers of g (cg_lo / cg_hi / cg_h as ntt_tables::lo / hi / h), on the
    // twiddle side (DIF store / DIT load); cmode 2, bit-reversed exponents: crow[row] = g^(rev_S(row)) on the twiddle side.
    // One or two products per element of this pass instead of a scaling pass over the array.
    unsigned cmode;
    const void *crow, *cg_lo, *cg_hi;
    unsigned cg_h;
};
// the coset factors of one element of such a pass (typed views of the fields above)
template<class F> SPPARK_DEVFN F ntt_pass_crow(const ntt_pass& P, unsigned row) { return reinterpret_cast<const F*>(P.crow)[row]; }
template<class F> SPPARK_DEVFN F ntt_pass_gcol(const ntt_pass& P, size_t col)
{   return reinterpret_cast<const F*>(P.cg_lo)[col & (((size_t)1 << P.cg_h) - 1)] * reinterpret_cast<const F*>(P.cg_hi)[col >> P.cg_h];   }

SPPARK_DEVFN unsigned bit_rev32(unsigned x, unsigned bits)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? __brev(x) >> (32 - bits) : 0;
#else
    unsigned r = 0;
    for (unsigned k = 0; k < bits; k++) r |= ((x >> k) & 1u) << (bits - 1 - k);
    return r;
#endif
}

template<class F> SPPARK_DEVFN F ntt_twiddle(const ntt_tables<F>& T, size_t e)
{   return T.lo[e & (((size_t)1 << T.h) - 1)] * T.hi[e >> T.h];   }


// Inter-pass twiddles of one work item.  Its 2^R elements need w^(col * rev_S(mid))
// where rev_S(mid) = rev_R(j) * 2^lo_bits + fixed, j = the item's own index: that is
// t0 * step^(rev_R(j)) with t0 = w^(col*fixed), step = w^(col << lo_bits).  Two table
// look-ups (each a 64-lane gather of lo[] and hi[]) and 2^R + R - 2 products
// generate all of them, instead of 2^R look-ups + 2^R products: the gathers, not
// the arithmetic, were what bounded the passes.  pw[k] = t0 * step^k, natural k.
template<class F, unsigned R, bool HAS_T0>
SPPARK_DEVFN void ntt_twiddle_powers(F* pw, const ntt_tables<F>& T, size_t e_t0, size_t e_step)
{
    pw[0] = HAS_T0 ? ntt_twiddle(T, e_t0) : F::one();
    if (R == 0) return;
    F sj = ntt_twiddle(T, e_step);
    #pragma unroll
    for (unsigned j = 0; j < R; j++) {
        #pragma unroll
        for (unsigned k = 1u << j; k < (2u << j); k++)
            pw[k] = (!HAS_T0 && k == (1u << j)) ? sj : pw[k - (1u << j)] * sj;
        if (j + 1 < R) sj = sj * sj;
    }
}
// wide elements keep the per-element look-up: 2^R extra 256-bit values do not fit the registers
template<class F> struct ntt_gen_twiddles { static constexpr bool value = sizeof(F) <= 8; };

// in-register 2^R-point transforms on x[0 .. 2^R)
template<class F, bool INV, unsigned R>
SPPARK_DEVFN void radix_dif(F* x, const F* inner)              // natural in -> bit-reversed out
{
    if (R == 0) return;
    #pragma unroll
    for (unsigned t = 0; t < R; t++) {
        const unsigned lgh = R - 1 - t, half = 1u << lgh;
        #pragma unroll
        for (unsigned j = 0; j < ((1u << R) >> 1); j++) {
            const unsigned off = j & (half - 1), m0 = ((j >> lgh) << (lgh + 1)) + off, m1 = m0 + half;
            // (a - b) * w with w = +-(root): a minus sign turns a - b into b - a
            F sum, dif;
            if (F::template root_neg<INV>(R, off << t)) F::bfly(x[m1], x[m0], sum, dif);
            else                                        F::bfly(x[m0], x[m1], sum, dif);
            x[m0] = sum;
            x[m1] = F::template mul_root<INV>(dif, R, off << t, inner);
        }
    }
}
template<class F, bool INV, unsigned R>
SPPARK_DEVFN void radix_dit(F* x, const F* inner)              // bit-reversed in -> natural out
{
    if (R == 0) return;
    #pragma unroll
    for (unsigned t = 0; t < R; t++) {
        const unsigned lgh = t, half = 1u << lgh;
        #pragma unroll
        for (unsigned j = 0; j < ((1u << R) >> 1); j++) {
            const unsigned off = j & (half - 1), m0 = ((j >> lgh) << (lgh + 1)) + off, m1 = m0 + half;
            // a +- b*w with w = +-(root): a minus sign swaps the two outputs
            F bw = F::template mul_root<INV>(x[m1], R, off << (R - 1 - t), inner), sum, dif;
            F::bfly(x[m0], bw, sum, dif);
            const bool neg = F::template root_neg<INV>(R, off << (R - 1 - t));
            x[m0] = neg ? dif : sum;
            x[m1] = neg ? sum : dif;
        }
    }
}

// tile geometry helpers ------------------------------------------------------
struct ntt_tile_geom { size_t row0; unsigned c0, lgQ; };
SPPARK_DEVFN ntt_tile_geom ntt_geom(const ntt_pass& P, size_t tile_id)
{
    ntt_tile_geom g; g.lgQ = P.lg_cur - P.S;
    if (P.lgG) { g.row0 = (tile_id << P.lgG) << P.S; g.c0 = 0; }
    else {
        size_t tiles_per_sub = (size_t)1 << (g.lgQ - P.lgC);
        g.row0 = (tile_id / tiles_per_sub) << P.S;
        g.c0 = (unsigned)(tile_id % tiles_per_sub) << P.lgC;
    }
    return g;
}
// LDS index of tile row gm (= g*2^S + mid), column c.  Adjacent rows are swapped when bit R2 of
// the row number is set: in the strided round two consecutive values of `a` then land on
// opposite halves of the 64 banks instead of on the same ones, without the pad rows that an
// offset-based scheme needs -- a 2^12-element Goldilocks tile is exactly 32 KB, so five
// work-groups (instead of four) share a CU's 160 KB.
template<unsigned R2>
SPPARK_DEVFN unsigned ntt_lds_index(unsigned gm, unsigned c, unsigned lgC)
{   return ((gm ^ ((gm >> R2) & 1u)) << lgC) + c;   }

// A work group of the HIGH-bits round: fixed (g, b, c), the 2^R1 values of a.
// A work group of the LOW-bits round:  fixed (g, a, c), the 2^R2 values of b.
// Row (mid) = a * 2^R2 + b.

// GS/DIF pass, round 1 (high bits): HBM -> registers -> DIF_{2^R1} -> * w_{2^S}^(b*rev(a)) -> LDS
// CT/DIT pass, round 2 (high bits): LDS -> * w_{2^S}^(b*rev(a)) -> DIT_{2^R1} -> scale -> HBM
template<class F, bool DIF, bool INV, unsigned R1, unsigned R2>
SPPARK_DEVFN void ntt_round_high(F* data, F* tile, const ntt_tables<F>& T, const ntt_pass& P,
                                 size_t tile_id, unsigned tid, unsigned nt)
{
    constexpr unsigned S = R1 + R2;
    const ntt_tile_geom geo = ntt_geom(P, tile_id);
    const unsigned ngroups = 1u << (P.lgG + R2 + P.lgC), C = 1u << P.lgC;
    for (unsigned gi = tid; gi < ngroups; gi += nt) {
        const unsigned c = gi & (C - 1), rest = gi >> P.lgC;
        const unsigned b = rest & ((1u << R2) - 1), g = rest >> R2;
        F x[1u << R1];
        #pragma unroll
        for (unsigned a = 0; a < (1u << R1); a++) {
            const unsigned gm = (g << S) + (a << R2) + b;
            if (DIF || R2 == 0) x[a] = data[((geo.row0 + gm) << geo.lgQ) + geo.c0 + c];
            else                x[a] = tile[ntt_lds_index<R2>(gm, c, P.lgC)];
            if (DIF && P.cmode == 1) x[a] = x[a] * ntt_pass_crow<F>(P, (a << R2) + b);     // coset, data side (uniform branch)
        }
        // single-round pass: the twiddle side is here too
        F cgc = F();
        if (R2 == 0 && P.cmode == 1) cgc = ntt_pass_gcol<F>(P, geo.c0 + c);
        // single-round pass (R2 == 0): the inter-pass twiddles w^(col * rev(a)) are applied here
        constexpr bool GEN = ntt_gen_twiddles<F>::value && R2 == 0;
        F pw[GEN ? (1u << R1) : 1];
        if (GEN && geo.lgQ)
            ntt_twiddle_powers<F, R1, false>(pw, T, 0, (size_t)(geo.c0 + c) << (T.lg_n - P.lg_cur));
        if (DIF) {
            radix_dif<F, INV, R1>(x, T.inner);
            #pragma unroll
            for (unsigned a = 0; a < (1u << R1); a++) {
                if (R2) { unsigned e = b * bit_rev32(a, R1); if (e) x[a] = x[a] * T.inner[(1u << S) + e]; }
                if (R2 == 0) {                          // single-round pass: finish here
                    if (GEN) { if (geo.lgQ && a) x[a] = x[a] * pw[GEN ? bit_rev32(a, R1) : 0]; }
                    else if (geo.lgQ) { unsigned ex = (geo.c0 + c) * bit_rev32(a, R1); x[a] = x[a] * ntt_twiddle(T, (size_t)ex << (T.lg_n - P.lg_cur)); }
                    if (P.cmode == 1) x[a] = x[a] * cgc;
                    if (P.cmode == 2) x[a] = x[a] * ntt_pass_crow<F>(P, a);
                    if (P.apply_scale) x[a] = x[a] * T.scale;
                    data[((geo.row0 + (g << S) + a) << geo.lgQ) + geo.c0 + c] = x[a];
                } else {
                    tile[ntt_lds_index<R2>((g << S) + (a << R2) + b, c, P.lgC)] = x[a];
                }
            }
        } else {
            #pragma unroll
            for (unsigned a = 0; a < (1u << R1); a++) {
                if (R2) { unsigned e = b * bit_rev32(a, R1); if (e) x[a] = x[a] * T.inner[(1u << S) + e]; }
                if (R2 == 0 && geo.lgQ) {
                    if (GEN) { if (a) x[a] = x[a] * pw[GEN ? bit_rev32(a, R1) : 0]; }
                    else { unsigned ex = (geo.c0 + c) * bit_rev32(a, R1); x[a] = x[a] * ntt_twiddle(T, (size_t)ex << (T.lg_n - P.lg_cur)); }
                }
                if (R2 == 0 && P.cmode == 1) x[a] = x[a] * cgc;
                if (R2 == 0 && P.cmode == 2) x[a] = x[a] * ntt_pass_crow<F>(P, a);
            }
            radix_dit<F, INV, R1>(x, T.inner);
            #pragma unroll
            for (unsigned a = 0; a < (1u << R1); a++) {
                if (P.cmode == 1) x[a] = x[a] * ntt_pass_crow<F>(P, (a << R2) + b);        // coset, data side
                if (P.apply_scale) x[a] = x[a] * T.scale;
                data[((geo.row0 + (g << S) + (a << R2) + b) << geo.lgQ) + geo.c0 + c] = x[a];
            }
        }
    }
}

// GS/DIF pass, round 2 (low bits): LDS -> DIF_{2^R2} -> * w_{n_cur}^(lo*rev_S(mid)) -> scale -> HBM
// CT/DIT pass, round 1 (low bits): HBM -> * w_{n_cur}^(lo*rev_S(mid)) -> DIT_{2^R2} -> LDS
template<class F, bool DIF, bool INV, unsigned R1, unsigned R2>
SPPARK_DEVFN void ntt_round_low(F* data, F* tile, const ntt_tables<F>& T, const ntt_pass& P,
                                size_t tile_id, unsigned tid, unsigned nt)
{
    constexpr unsigned S = R1 + R2;
    const ntt_tile_geom geo = ntt_geom(P, tile_id);
    const unsigned ngroups = 1u << (P.lgG + R1 + P.lgC), C = 1u << P.lgC;
    for (unsigned gi = tid; gi < ngroups; gi += nt) {
        const unsigned c = gi & (C - 1), rest = gi >> P.lgC;
        const unsigned a = rest & ((1u << R1) - 1), g = rest >> R1;
        F x[1u << R2];
        // w^(col * rev_S(a*2^R2 + b)) = w^(col*rev(a)) * (w^(col << R1))^rev(b)
        constexpr bool GEN = ntt_gen_twiddles<F>::value;
        F pw[GEN ? (1u << R2) : 1];
        const bool tabled = T.pass_tw != nullptr;               // uniform over the launch
        // (wide elements: the table entry is loaded where it is used -- one product per element instead of the
        // lo x hi product that makes the twiddle plus the one that applies it)
        if (GEN && geo.lgQ) {
            if (GEN && tabled) {
                #pragma unroll
                for (unsigned b = 0; b < (1u << R2); b++)        // natural b here; used as pw[b] below
                    pw[b] = T.pass_tw[((size_t)((a << R2) + b) << geo.lgQ) + geo.c0 + c];
            } else {
                const unsigned sh = T.lg_n - P.lg_cur;
                const size_t col = geo.c0 + c;
                ntt_twiddle_powers<F, R2, true>(pw, T, (col * bit_rev32(a, R1)) << sh, (col << R1) << sh);
            }
        }
        F cgc = F();                                            // coset, twiddle side (uniform branches)
        if (P.cmode == 1) cgc = ntt_pass_gcol<F>(P, geo.c0 + c);
        #pragma unroll
        for (unsigned b = 0; b < (1u << R2); b++) {
            const unsigned gm = (g << S) + (a << R2) + b;
            if (DIF) x[b] = tile[ntt_lds_index<R2>(gm, c, P.lgC)];
            else {
                x[b] = data[((geo.row0 + gm) << geo.lgQ) + geo.c0 + c];
                if (geo.lgQ) {
                    if (GEN) x[b] = x[b] * pw[GEN ? (tabled ? b : bit_rev32(b, R2)) : 0];
                    else if (tabled) x[b] = x[b] * T.pass_tw[((size_t)((a << R2) + b) << geo.lgQ) + geo.c0 + c];
                    else { unsigned ex = (geo.c0 + c) * bit_rev32((a << R2) + b, S); x[b] = x[b] * ntt_twiddle(T, (size_t)ex << (T.lg_n - P.lg_cur)); }
                }
                if (P.cmode == 1) x[b] = x[b] * cgc;
                if (P.cmode == 2) x[b] = x[b] * ntt_pass_crow<F>(P, (a << R2) + b);
            }
        }
        if (DIF) {
            radix_dif<F, INV, R2>(x, T.inner);
            #pragma unroll
            for (unsigned b = 0; b < (1u << R2); b++) {
                if (geo.lgQ) {
                    if (GEN) x[b] = x[b] * pw[GEN ? (tabled ? b : bit_rev32(b, R2)) : 0];
                    else if (tabled) x[b] = x[b] * T.pass_tw[((size_t)((a << R2) + b) << geo.lgQ) + geo.c0 + c];
                    else { unsigned ex = (geo.c0 + c) * bit_rev32((a << R2) + b, S); x[b] = x[b] * ntt_twiddle(T, (size_t)ex << (T.lg_n - P.lg_cur)); }
                }
                if (P.cmode == 1) x[b] = x[b] * cgc;
                if (P.cmode == 2) x[b] = x[b] * ntt_pass_crow<F>(P, (a << R2) + b);
                if (P.apply_scale) x[b] = x[b] * T.scale;
                data[((geo.row0 + (g << S) + (a << R2) + b) << geo.lgQ) + geo.c0 + c] = x[b];
            }
        } else {
            radix_dit<F, INV, R2>(x, T.inner);
            #pragma unroll
            for (unsigned b = 0; b < (1u << R2); b++)
                tile[ntt_lds_index<R2>((g << S) + (a << R2) + b, c, P.lgC)] = x[b];
        }
    }
}

template<class F, bool DIF, bool INV, unsigned R1, unsigned R2>
// (99 VGPRs for Goldilocks <4,4>: four waves per SIMD.  Forcing five -- 94 VGPRs, no spills, five
// 32 KB tiles per CU -- changes nothing: 0.264 vs 0.263 ms at 2^24, tools/gpu_r2_job14.sh; the pass is
// bound by its instruction count.)
__global__ __launch_bounds__(512)
void k_ntt_pass(F* data, ntt_tables<F> T, ntt_pass P)
{
    extern __shared__ unsigned char ntt_lds[];
    F* tile = reinterpret_cast<F*>(ntt_lds);
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    if constexpr (R2 == 0) {
        ntt_round_high<F, DIF, INV, R1, R2>(data, tile, T, P, blockIdx.x, tid, nt);
    } else if (DIF) {
        ntt_round_high<F, DIF, INV, R1, R2>(data, tile, T, P, blockIdx.x, tid, nt);
        __syncthreads();
        ntt_round_low<F, DIF, INV, R1, R2>(data, tile, T, P, blockIdx.x, tid, nt);
    } else {
        ntt_round_low<F, DIF, INV, R1, R2>(data, tile, T, P, blockIdx.x, tid, nt);
        __syncthreads();
        ntt_round_high<F, DIF, INV, R1, R2>(data, tile, T, P, blockIdx.x, tid, nt);
    }
}

// ---- the same pass, ONE STAGE PER ROUND: 256-bit fields (round 4) ------------------------------------------------------
// A pass of the 256-bit fields in registers is radix-4 x radix-4 (S = 4: sixteen 8-word elements per lane is where the
// registers end), i.e. six passes at 2^24 and FOUR at 2^16, and every pass costs each element one inter-pass and most
// elements one w_{2^S} twiddle product on top of its S/2 butterfly products: 3.75 products per element for 4 stages.
// Here the tile stays in LDS for all S <= 8 stages of a pass, a lane does ONE butterfly per stage (two elements: 16
// data registers, any occupancy), and a barrier separates the stages: S/2 + 1 = 5 products per element for 8 stages,
// half the passes, half the HBM traffic -- and on small transforms, which are latency-bound (a 2^16 transform is 256
// waves on 1024 SIMDs), half the dependent chain per element and twice the lanes.  LDS moves 128 bytes per lane and
// stage against >= 300 instructions of one 256-bit product: not a limit.  Same function of (data, T, P) as
// k_ntt_pass<R1, R2> with R1 + R2 = P.S (the reference splits its steps the same way, 8 stages per launch with
// 1 butterfly per thread and stage: ntt/kernels/gs_mixed_radix_wide.cu:9-119; measured beside this one in
// profiles/r04_ntt_vs_reference_timing.log).
// LDS image of a tile of 32-byte elements: 16-byte CHUNK k of element e lives in plane k, at chunk index
// k * elems + e.  A ds_read_b128 serves 16 lanes per LDS cycle and wants their sixteen 16-byte slots distinct modulo
// 256 bytes; with whole elements side by side the two halves of an element are two separate reads that each touch only
// every other slot -- at least 2-way conflicts in every stage, 4- to 16-way in the last ones (counters of the first
// version: SQ_LDS_BANK_CONFLICT = 80 % of SQ_LDS_IDX_ACTIVE, the pass LDS-bound at 1.0 ms; profiles/r04_ntt_wide_pmc.txt).
// In planes, consecutive elements are consecutive slots: the strided stages are conflict-free down to half = 4 rows.
typedef uint4 __attribute__((may_alias)) ntt_lat_chunk;
template<class F> SPPARK_DEVFN F ntt_lat_get(const F* tile, unsigned e, unsigned elems)
{
    if constexpr (sizeof(F) % 16 == 0 && sizeof(F) > 16) {
        F r;
        #pragma unroll
        for (unsigned k = 0; k < sizeof(F) / 16; k++)
            reinterpret_cast<ntt_lat_chunk*>(&r)[k] = reinterpret_cast<const ntt_lat_chunk*>(tile)[k * elems + e];
        return r;
    } else return tile[e];
}
template<class F> SPPARK_DEVFN void ntt_lat_put(F* tile, unsigned e, unsigned elems, const F& x)
{
    if constexpr (sizeof(F) % 16 == 0 && sizeof(F) > 16) {
        #pragma unroll
        for (unsigned k = 0; k < sizeof(F) / 16; k++)
            reinterpret_cast<ntt_lat_chunk*>(tile)[k * elems + e] = reinterpret_cast<const ntt_lat_chunk*>(&x)[k];
    } else tile[e] = x;
}
template<class F>
SPPARK_DEVFN F ntt_lat_twiddle(const ntt_tables<F>& T, const ntt_pass& P, const ntt_tile_geom& geo, unsigned mid, unsigned c)
{
    if (T.pass_tw != nullptr) return T.pass_tw[((size_t)mid << geo.lgQ) + geo.c0 + c];
    const size_t ex = (size_t)(geo.c0 + c) * bit_rev32(mid, P.S);
    return ntt_twiddle(T, ex << (T.lg_n - P.lg_cur));
}
// HBM -> (CT/DIT: * w_{n_cur}^(col * rev_S(mid))) -> LDS; tile element e = row gm * C + column c
template<class F, bool DIF>
SPPARK_DEVFN void ntt_lat_load(const F* data, F* tile, const ntt_tables<F>& T, const ntt_pass& P, size_t tile_id, unsigned tid, unsigned nt)
{
    const ntt_tile_geom geo = ntt_geom(P, tile_id);
    const unsigned elems = 1u << (P.lgG + P.S + P.lgC), C = 1u << P.lgC;
    for (unsigned e = tid; e < elems; e += nt) {
        const unsigned c = e & (C - 1), gm = e >> P.lgC;
        F x = data[((geo.row0 + gm) << geo.lgQ) + geo.c0 + c];
        if (!DIF && geo.lgQ) x = x * ntt_lat_twiddle(T, P, geo, gm & ((1u << P.S) - 1), c);
        ntt_lat_put(tile, e, elems, x);
    }
}
// stage t of the 2^S-point transforms of the tile's columns: lane = (sub-problem g, butterfly j, column c)
template<class F, bool DIF, bool INV>
SPPARK_DEVFN void ntt_lat_stage(F* tile, const ntt_tables<F>& T, const ntt_pass& P, unsigned t, unsigned tid, unsigned nt)
{
    const unsigned S = P.S, nb = 1u << (P.lgG + S - 1 + P.lgC), C = 1u << P.lgC, elems = 2 * nb;
    const unsigned lgh = DIF ? S - 1 - t : t, half = 1u << lgh;
    for (unsigned bf = tid; bf < nb; bf += nt) {
        const unsigned c = bf & (C - 1), r = bf >> P.lgC, j = r & ((1u << (S - 1)) - 1), g = r >> (S - 1);
        const unsigned off = j & (half - 1), m0 = ((j >> lgh) << (lgh + 1)) + off, m1 = m0 + half;
        const unsigned i0 = (((g << S) + m0) << P.lgC) + c, i1 = (((g << S) + m1) << P.lgC) + c;
        F sum, dif;
        const F x0 = ntt_lat_get(tile, i0, elems), x1 = ntt_lat_get(tile, i1, elems);
        if (DIF) {                                              // natural in -> bit-reversed out (as radix_dif)
            const unsigned k = off << t;
            if (F::template root_neg<INV>(S, k)) F::bfly(x1, x0, sum, dif);
            else                                 F::bfly(x0, x1, sum, dif);
            ntt_lat_put(tile, i0, elems, sum);
            ntt_lat_put(tile, i1, elems, F::template mul_root<INV>(dif, S, k, T.inner));
        } else {                                                // bit-reversed in -> natural out (as radix_dit)
            const unsigned k = off << (S - 1 - t);
            const F bw = F::template mul_root<INV>(x1, S, k, T.inner);
            F::bfly(x0, bw, sum, dif);
            const bool neg = F::template root_neg<INV>(S, k);
            ntt_lat_put(tile, i0, elems, neg ? dif : sum);
            ntt_lat_put(tile, i1, elems, neg ? sum : dif);
        }
    }
}
// LDS -> (GS/DIF: * w_{n_cur}^(col * rev_S(mid))) -> (* 1/n) -> HBM
template<class F, bool DIF>
SPPARK_DEVFN void ntt_lat_store(F* data, const F* tile, const ntt_tables<F>& T, const ntt_pass& P, size_t tile_id, unsigned tid, unsigned nt)
{
    const ntt_tile_geom geo = ntt_geom(P, tile_id);
    const unsigned elems = 1u << (P.lgG + P.S + P.lgC), C = 1u << P.lgC;
    for (unsigned e = tid; e < elems; e += nt) {
        const unsigned c = e & (C - 1), gm = e >> P.lgC;
        F x = ntt_lat_get(tile, e, elems);
        if (DIF && geo.lgQ) x = x * ntt_lat_twiddle(T, P, geo, gm & ((1u << P.S) - 1), c);
        if (P.apply_scale) x = x * T.scale;
        data[((geo.row0 + gm) << geo.lgQ) + geo.c0 + c] = x;
    }
}
// Fusing the small-half stages with the store / load in registers (radix_dif<R> / radix_dit<R> of 2^R consecutive rows,
// R = 2 | 3) was written in round 4 and measured in round 5: slower at every size (BLS12-381 Fr 2^24 forward 2.12 ->
// 2.28 ms with R = 2, 3.01 with R = 3; profiles/r05_ntt_lat_tail_ab.log) -- a quarter of the lanes busy in that phase
// costs more than the 0.375 products per element it saves.  Removed.
template<class F, bool DIF, bool INV>
__global__ __launch_bounds__(1024)
void k_ntt_pass_lat(F* data, ntt_tables<F> T, ntt_pass P)
{
    extern __shared__ unsigned char ntt_lds[];
    F* tile = reinterpret_cast<F*>(ntt_lds);
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    ntt_lat_load<F, DIF>(data, tile, T, P, blockIdx.x, tid, nt);
    __syncthreads();
    for (unsigned t = 0; t < P.S; t++) {
        ntt_lat_stage<F, DIF, INV>(tile, T, P, t, tid, nt);
        __syncthreads();
    }
    ntt_lat_store<F, DIF>(data, tile, T, P, blockIdx.x, tid, nt);
}

// ---- whole transforms of <= 2^11 elements: ONE work-group, ONE launch (round 5) ----------------------------------------
// A small transform is pure latency: one launch costs ~3 us whatever it does, and the passes above gave a 2^8 transform
// to SIXTEEN lanes (radix-16 in registers: 8 us of dependent arithmetic in one wave) and a 2^9 / 2^10 one two launches,
// with a third for the bit reversal of the NN / RR orders and a fourth for a coset -- the only sizes at which the
// reference's own build was ahead (3.0 / 4.6 / 5.1 us against 8.0 / 6.4 / 8.2, profiles/r04_ntt_vs_reference_timing.log;
// it runs <= 2^10 in one launch too, ntt/ntt.cuh:106-107).
// Here n/2 lanes each keep ONE butterfly pair in registers for the whole transform (the shortest dependent chain there
// is).  After a stage a lane swaps one of its two values with the lane at distance 2^d -- inside a wave by lane-permute
// instructions (d < 6: no barrier, no LDS round trip; ntt_rx_regroup), through LDS with one barrier across waves (three
// stages of a 2^10 transform, four of a 2^11 one) -- which is the layout of the reference's narrow kernels
// (ntt/kernels/gs_mixed_radix_narrow.cu:58-118: shfl_bfly = ds_bpermute inside the wave, shared memory above).
// What differs: every twiddle of the lane (one per stage) is loaded at the top, together with the data and the coset
// powers, so that the whole transform pays ONE memory latency; the twiddles of a stage are one ROW of ntt_tables::inner
// (w_{2^R}^k for k < 2^R in consecutive entries), so that the lanes of a wave read consecutive entries; the single-word
// fields have one instance per size 2^8 ... 2^11, a straight line of stages that waits for each load where it is first
// used; and everything the driver would otherwise launch around the stages is folded into the load and the store: the
// bit-reversal permutations of the four orders (ntt/ntt.cuh:174-209), the coset powers g^k (ntt/kernels.cu:131-153), 1/n.
// The versions on the way, Goldilocks 2^8 ... 2^11, NR order, against the reference's build 2.9-3.1 / 3.5 / 4.4 / 11.2-11.5 us
// on the same box:
//   one butterfly per lane and stage, the array in LDS, a barrier per stage     3.9 / 4.4 / 5.2 / --    r05_ntt_small_first_version.log
//   this layout; twiddles w^(k 2^(lg-2-d)) gathered from the root table (up to 64 lines per wave and load, 2400 for a 2^10
//   transform), the size a run-time value (a branch and a select per stage, every stage behind ALL the loads)
//                                                                               2.7 / 3.2 / 4.7 / 7.0   r05_ntt_vs_reference_timing_before_sized.log
//   ... twiddle rows                                                            2.7 / 3.0 / 4.1 / 6.2   r05_ntt_small_sized_ab.log
//   ... and an instance per size                                                2.5 / 2.8 / 3.6 / 5.3   (2^8, 2^9: the host's issue rate)
// Measured on the way and NOT kept (same logs): two pairs per lane (n/4 lanes; gained 0.7 us at 2^11 before the instances per
// size, nothing after them, and loses 0.3 at 2^10); BabyBear's upper twiddles by squaring instead of loads (gained 0.3 us
// against the gathers, loses 0.1-0.2 against the rows); the exchanges at distance <= 8 as v_cndmask_b32 with a DPP source
// (16 instructions fewer per transform, 0-0.2 us slower).
// At 2^10 a work-group is ISSUE-bound on its one compute unit: eight waves x ~51 vector instructions per stage on four SIMDs.
// Same function of the array as the driver's general path; the emulation and the GPU tests hold both against the oracle.
enum { NTT_SMALL_GS = 1,            // GS / DIF stages (natural in -> bit-reversed out); else CT / DIT
       NTT_SMALL_PERM_IN = 2,       // the array is bit-reversed on the way in (NN)
       NTT_SMALL_PERM_OUT = 4,      // ... on the way out (RR)
       NTT_SMALL_BITREV = 8,        // the reference's |bitrev| flag: which index the coset powers follow
       NTT_SMALL_COSET_IN = 16,     // forward coset: x[p] *= g^(bitrev ? rev(p) : p) before the stages
       NTT_SMALL_COSET_OUT = 32 };  // inverse coset: x[p] *= g^-(bitrev ? p : rev(p)) after them

#if defined(SPPARK_HOST_EMULATION)
extern "C" void sppark_emu_barrier();          // (tests/emu: the lanes of a work-group are host threads)
#endif
SPPARK_DEVFN void ntt_wg_barrier()
{
#if defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
#elif defined(SPPARK_HOST_EMULATION)
    sppark_emu_barrier();
#endif
}
template<unsigned I, unsigned N> struct ntt_static_for {        // fn(I), ..., fn(N - 1) with a compile-time index
    template<class Fn> SPPARK_DEVFN static void run(Fn&& fn) { fn(std::integral_constant<unsigned, I>()); ntt_static_for<I + 1, N>::run(fn); }
};
template<unsigned N> struct ntt_static_for<N, N> { template<class Fn> SPPARK_DEVFN static void run(Fn&&) {} };

// After a stage the lanes l and l ^ 2^D regroup: the one with bit D clear keeps the two SUMS (its own and its partner's), the
// other one the two DIFFERENCES -- (x0, x1) = (sum, sum') resp. (dif', dif).
//   D = 5, 4   v_permlane32_swap / v_permlane16_swap (gfx950): "rows 2, 3 of the first operand are swapped with rows 0, 1
//              of the second" / "odd rows ... with even rows" -- applied to (sum, dif) that IS the regrouping, one VALU
//              instruction per 32-bit word, no select;
//   D = 3, 1, 0  the partner's value by a DPP move (row_ror:8, quad_perm [2,3,0,1] / [1,0,3,2]);  D = 2  ds_bpermute;
//   D >= 6 (across waves), and EVERY exchange of the host emulation: through LDS -- two buffers of |lanes| elements used in
//              turn, so that ONE barrier per exchange is enough (a lane can only write a buffer again after the barrier of
//              the exchange in between, which every lane reaches with its read of that buffer behind it).
template<class F> SPPARK_DEVFN F ntt_rx_pick(bool first, const F& a, const F& b) { return first ? a : b; }
template<class F, unsigned D>
SPPARK_DEVFN void ntt_rx_regroup(F& x0, F& x1, const F& sum, const F& dif, unsigned lane, F* lds, unsigned lanes, unsigned& par)
{
    const bool upper = ((lane >> D) & 1u) != 0;
    F recv;
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr unsigned W = sizeof(F) / 4;
    static_assert(sizeof(F) % 4 == 0, "whole words");
    if constexpr (D == 5 || D == 4) {
        u32 s[W], d[W];
        __builtin_memcpy(s, &sum, sizeof(F)); __builtin_memcpy(d, &dif, sizeof(F));
        #pragma unroll
        for (unsigned k = 0; k < W; k++) {
            if constexpr (D == 5) { auto r = __builtin_amdgcn_permlane32_swap(s[k], d[k], false, false); s[k] = r[0]; d[k] = r[1]; }
            else                  { auto r = __builtin_amdgcn_permlane16_swap(s[k], d[k], false, false); s[k] = r[0]; d[k] = r[1]; }
        }
        __builtin_memcpy(&x0, s, sizeof(F)); __builtin_memcpy(&x1, d, sizeof(F));
        return;
    } else if constexpr (D < 6) {
        const F send = ntt_rx_pick(upper, sum, dif);
        u32 w[W];
        __builtin_memcpy(w, &send, sizeof(F));
        #pragma unroll
        for (unsigned k = 0; k < W; k++) {
            if constexpr (D == 3)      w[k] = (u32)__builtin_amdgcn_update_dpp(0, (int)w[k], 0x128, 0xf, 0xf, false);   // row_ror:8
            else if constexpr (D == 1) w[k] = (u32)__builtin_amdgcn_update_dpp(0, (int)w[k], 0x4e, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
            else if constexpr (D == 0) w[k] = (u32)__builtin_amdgcn_update_dpp(0, (int)w[k], 0xb1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
            else                       w[k] = (u32)__builtin_amdgcn_ds_bpermute((int)(((lane & 63u) ^ (1u << D)) << 2), (int)w[k]);
        }
        __builtin_memcpy(&recv, w, sizeof(F));
    } else
#endif
    {
        F* buf = lds + (size_t)par * lanes;
        par ^= 1;
        ntt_lat_put(buf, lane, lanes, ntt_rx_pick(upper, sum, dif));
        ntt_wg_barrier();
        recv = ntt_lat_get(buf, lane ^ (1u << D), lanes);
    }
    x0 = ntt_rx_pick(upper, recv, sum); x1 = ntt_rx_pick(upper, dif, recv);
}

// Lane |l| of max(n/2, 64) carries the butterfly pair l of the n/2-lane network above (lanes beyond n/2 carry zeros through the
// same exchanges and touch no memory).
// LGC: the size the body is compiled for (0: any size up to the cap, read from the tables at run time).  With the size
// known, the stages are one straight line -- no per-stage branch, no select between the top twiddle and the others, and
// each stage waits for ITS twiddle instead of all the loads.
template<class F, bool INV, bool GS, unsigned LGC = 0>
SPPARK_DEVFN void ntt_rx_run(F* data, F* lds, const ntt_tables<F>& T, const ntt_tables<F>& G, unsigned flags,
                             unsigned l, unsigned lanes)
{
    constexpr unsigned MAXLG = ntt_small_cap<F>::value;
    static_assert(LGC <= MAXLG && (LGC == 0 || LGC >= 7), "a compiled-in size fills at least one wave");
    // (a one-element "transform" has no pair: every lane idle, nothing read or written -- the driver returns before it gets
    // here, ntt_engine::run, but the kernel is safe on its own)
    const unsigned lg = LGC ? LGC : T.lg_n, nh = lg ? 1u << (lg - 1) : 0u;
    const bool live = l < nh;
    const unsigned lq = live ? l : 0;                           // (idle lanes read the tables at valid indices)
    F x0 = F(), x1 = F(), sum, dif, g0 = F(), g1 = F(), w[MAXLG];
    // ---- every load of the transform, issued together ------------------------------------------------------------------
    // positions of the pair in the working array: GS (l, l + n/2) -> (2l, 2l + 1); CT (2l, 2l + 1) -> (l, l + n/2)
    const unsigned pin0 = GS ? lq : 2 * lq, pin1 = GS ? lq + nh : 2 * lq + 1;
    const unsigned pout0 = GS ? 2 * lq : lq, pout1 = GS ? 2 * lq + 1 : lq + nh;
    if (live) {
        x0 = data[(flags & NTT_SMALL_PERM_IN) ? bit_rev32(pin0, lg) : pin0];
        x1 = data[(flags & NTT_SMALL_PERM_IN) ? bit_rev32(pin1, lg) : pin1];
    }
    // w[d] is the twiddle of the stage with halves of 2^d elements, d >= 1 -- GS: stage lg-1-d, followed by the exchange at
    // distance 2^(d-1); CT: stage d, preceded by it -- and in both networks it is w_{2^(d+1)}^(l mod 2^d): entry l mod 2^d of
    // row d+1 of T.inner.  Issued in the order the stages use them.
    ntt_static_for<1, MAXLG>::run([&](auto K) {
        constexpr unsigned d = GS ? MAXLG - decltype(K)::value : decltype(K)::value;
        if (d < lg) w[d] = T.inner[(2u << d) + (lq & ((1u << d) - 1))];
    });
    if (flags & NTT_SMALL_COSET_IN) {
        g0 = G.lo[(flags & NTT_SMALL_BITREV) ? bit_rev32(pin0, lg) : pin0];
        g1 = G.lo[(flags & NTT_SMALL_BITREV) ? bit_rev32(pin1, lg) : pin1];
        x0 = x0 * g0; x1 = x1 * g1;
    } else if (flags & NTT_SMALL_COSET_OUT) {
        g0 = G.lo[(flags & NTT_SMALL_BITREV) ? pout0 : bit_rev32(pout0, lg)];
        g1 = G.lo[(flags & NTT_SMALL_BITREV) ? pout1 : bit_rev32(pout1, lg)];
    }
    // ---- the stages ----------------------------------------------------------------------------------------------------
    unsigned par = 0;
    if (GS) {
        // halves of 2^d: (x0 + x1, (x0 - x1) w), then the lanes l and l ^ 2^(d-1) regroup: the lower one keeps the sums, the
        // upper one the differences
        ntt_static_for<1, MAXLG>::run([&](auto K) {
            constexpr unsigned d = MAXLG - decltype(K)::value;                  // MAXLG - 1, ..., 1
            if (d < lg) {
                F::bfly(x0, x1, sum, dif); dif = dif * w[d];
                ntt_rx_regroup<F, d - 1>(x0, x1, sum, dif, l, lds, lanes, par);
            }
        });
        F::bfly(x0, x1, sum, dif);                              // the last stage: halves of 1, w^0
    } else {
        F::bfly(x0, x1, sum, dif);                              // stage 0: halves of 1, w^0
        ntt_static_for<1, MAXLG>::run([&](auto K) {
            constexpr unsigned d = decltype(K)::value;          // the exchange at distance 2^(d-1), then halves of 2^d
            if (d < lg) {
                ntt_rx_regroup<F, d - 1>(x0, x1, sum, dif, l, lds, lanes, par);
                F::bfly(x0, x1 * w[d], sum, dif);
            }
        });
    }
    x0 = sum; x1 = dif;
    // ---- the store -----------------------------------------------------------------------------------------------------
    if (INV) { x0 = x0 * T.scale; x1 = x1 * T.scale; }
    if (flags & NTT_SMALL_COSET_OUT) { x0 = x0 * g0; x1 = x1 * g1; }
    if (live) {
        data[(flags & NTT_SMALL_PERM_OUT) ? bit_rev32(pout0, lg) : pout0] = x0;
        data[(flags & NTT_SMALL_PERM_OUT) ? bit_rev32(pout1, lg) : pout1] = x1;
    }
}
template<class F, bool INV, unsigned LGC = 0>
__global__ __launch_bounds__(1u << ((LGC ? LGC : ntt_small_cap<F>::value) - 1))
void k_ntt_small(F* data, ntt_tables<F> T, ntt_tables<F> G, unsigned flags)
{
    extern __shared__ unsigned char ntt_lds[];
    F* lds = reinterpret_cast<F*>(ntt_lds);
    if (flags & NTT_SMALL_GS) ntt_rx_run<F, INV, true, LGC>(data, lds, T, G, flags, threadIdx.x, blockDim.x);    // (uniform over the launch)
    else                      ntt_rx_run<F, INV, false, LGC>(data, lds, T, G, flags, threadIdx.x, blockDim.x);
}
// the instances: X(INV, LGC).  Single-word fields: sizes 2^8 ... 2^11 compiled in, one kernel for everything below; the
// 256-bit fields (a stage is ~350 instructions, the branches are noise): the run-time form only.
#define SPPARK_NTT_SMALL_ALL_NARROW(X) X(false, 0) X(false, 8) X(false, 9) X(false, 10) X(false, 11) \
                                       X(true, 0) X(true, 8) X(true, 9) X(true, 10) X(true, 11)
#define SPPARK_NTT_SMALL_ALL_WIDE(X)   X(false, 0) X(true, 0)
// the flags of an (order, direction, type) call -- ntt/ntt.cuh:174-209: NN = bit_rev + CT, NR = GS, RN = CT, RR = GS + bit_rev
static inline unsigned ntt_small_flags(int order, bool inverse, bool coset)
{
    unsigned f = 0;
    switch (order) {
        case 0:  f = NTT_SMALL_PERM_IN | NTT_SMALL_BITREV; break;
        case 1:  f = NTT_SMALL_GS; break;
        case 2:  f = NTT_SMALL_BITREV; break;
        default: f = NTT_SMALL_GS | NTT_SMALL_BITREV | NTT_SMALL_PERM_OUT; break;
    }
    if (coset) f |= inverse ? NTT_SMALL_COSET_OUT : NTT_SMALL_COSET_IN;
    return f;
}

// (R1, R2) = (ceil(S/2), floor(S/2)); CALL(R1, R2) is expanded for the pass's S
#define SPPARK_NTT_DISPATCH_S(S, CALL)                                  \
    switch (S) {                                                        \
        case 1: CALL(1, 0); break; case 2: CALL(1, 1); break;           \
        case 3: CALL(2, 1); break; case 4: CALL(2, 2); break;           \
        case 5: CALL(3, 2); break; case 6: CALL(3, 3); break;           \
        case 7: CALL(4, 3); break; default: CALL(4, 4); break;          \
    }

// the same for at most 4 stages per pass (wide fields)
#define SPPARK_NTT_DISPATCH_S4(S, CALL)                                 \
    switch (S) {                                                        \
        case 1: CALL(1, 0); break; case 2: CALL(1, 1); break;           \
        case 3: CALL(2, 1); break; default: CALL(2, 2); break;          \
    }

// LDS elements a tile needs
static inline size_t ntt_lds_elems(const ntt_pass& P)
{
    unsigned R2 = P.S / 2;
    if (R2 == 0) return 0;
    return (size_t)1 << (P.lgG + P.S + P.lgC);
}

// bit-reversal permutation in place (NN and RR orders; ntt/ntt.cuh:44-79)
template<class F>
SPPARK_DEVFN void bitrev_item(F* data, unsigned lg_n, size_t i)
{
    size_t r = 0;
    r = bit_rev32((unsigned)i, lg_n);                        // lg_n <= 32: the index fits 32 bits
    if (i < r) { F t = data[i]; data[i] = data[r]; data[r] = t; }
}
template<class F>
__global__ __launch_bounds__(256) void k_bitrev(F* data, unsigned lg_n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ((size_t)1 << lg_n)) bitrev_item(data, lg_n, i);
}

// Tiled in-place bit reversal.  Index = (hi | mid | lo) with TB bits of hi and lo; its
// reversal is (rev lo | rev mid | rev hi), so the 2^TB x 2^TB tile of one `mid` maps onto the
// tile of rev(mid) with rows and columns exchanged: both tiles are read and written as rows of
// 2^TB consecutive elements (coalesced), the transposition happens in LDS.  One work-group per
// pair {mid, rev mid} (the larger one of a pair returns immediately).  The element-wise kernel
// above reads and writes single elements at bit-reversed addresses and costs as much as the
// whole transform at 2^24.
template<unsigned TB> SPPARK_DEVFN unsigned bitrev_lds_index(unsigned row, unsigned col)
{   return row * ((1u << TB) + 1) + col;   }          // one pad element per row: column reads spread over banks

template<class F, unsigned TB>
SPPARK_DEVFN void bitrev_tile_item(F* data, F* ldsA, F* ldsB, unsigned lg_n, size_t mid, unsigned tid, unsigned nt, int phase)
{
    const unsigned lg_mid = lg_n - 2 * TB, T = 1u << TB;
    size_t rmid = 0;
    rmid = bit_rev32((unsigned)mid, lg_mid);
    if (mid > rmid) return;
    const bool pair = mid != rmid;
    if (phase == 0) {                                   // rows of the two tiles -> LDS
        for (unsigned e = tid; e < T * T; e += nt) {
            const unsigned hi = e >> TB, lo = e & (T - 1);
            ldsA[bitrev_lds_index<TB>(hi, lo)] = data[((size_t)hi << (lg_n - TB)) | (mid << TB) | lo];
            if (pair) ldsB[bitrev_lds_index<TB>(hi, lo)] = data[((size_t)hi << (lg_n - TB)) | (rmid << TB) | lo];
        }
    } else {                                            // LDS -> rows of the partner tile, transposed + reversed
        for (unsigned e = tid; e < T * T; e += nt) {
            const unsigned r = e >> TB, q = e & (T - 1);                        // destination row / column
            const unsigned lo = bit_rev32(r, TB), hi = bit_rev32(q, TB);        // source coordinates
            data[((size_t)r << (lg_n - TB)) | (rmid << TB) | q] = ldsA[bitrev_lds_index<TB>(hi, lo)];
            if (pair) data[((size_t)r << (lg_n - TB)) | (mid << TB) | q] = ldsB[bitrev_lds_index<TB>(hi, lo)];
        }
    }
}
template<class F, unsigned TB>
__global__ __launch_bounds__(256) void k_bitrev_tiled(F* data, unsigned lg_n)
{
    extern __shared__ unsigned char bitrev_lds[];
    F* ldsA = reinterpret_cast<F*>(bitrev_lds);
    F* ldsB = ldsA + (((1u << TB) + 1) << TB);
    bitrev_tile_item<F, TB>(data, ldsA, ldsB, lg_n, blockIdx.x, threadIdx.x, blockDim.x, 0);
    __syncthreads();
    bitrev_tile_item<F, TB>(data, ldsA, ldsB, lg_n, blockIdx.x, threadIdx.x, blockDim.x, 1);
}
// The same with 16-byte global accesses, for elements of 4 and 8 bytes (round 4).  The kernel above moves one element
// per lane and access and waits for each load before it issues the next: a 64 x 64 tile of 4-byte elements is 16
// dependent rounds of 256-byte wave accesses per tile, and the BabyBear instance ran at 2.7 TB/s where the 8-byte one
// (4 rounds, 512 bytes per wave access) reached 6.7 (profiles/r03_ntt_bench.log).  Here a lane owns V = 16 / sizeof(F)
// CONSECUTIVE elements of a row: all its loads (both tiles of the pair) are issued before the first LDS write; in the
// second phase it collects the V consecutive elements of a DESTINATION row from V source rows of the LDS image
// (V 4-byte LDS reads) and stores them with one 16-byte access.  |data| must be 16-byte aligned.
template<class F, unsigned TB>
SPPARK_DEVFN void bitrev_tile_vec_item(F* data, F* ldsA, F* ldsB, unsigned lg_n, size_t mid, unsigned tid, int phase)
{
    constexpr unsigned V = 16 / sizeof(F), T = 1u << TB, NT = 256, ROW = T / V, PER = T * T / V / NT;
    static_assert(V >= 2 && (T * T / V) % NT == 0 && T % V == 0, "vector slots must tile the work-group");
    struct alignas(16) vec { F e[V]; };
    const unsigned lg_mid = lg_n - 2 * TB;
    const size_t rmid = bit_rev32((unsigned)mid, lg_mid);
    if (mid > rmid) return;
    const bool pair = mid != rmid;
    if (phase == 0) {
        vec a[PER], b[PER];
        #pragma unroll
        for (unsigned k = 0; k < PER; k++) {
            const unsigned s = tid + k * NT, hi = s / ROW, lo = (s % ROW) * V;
            a[k] = *reinterpret_cast<const vec*>(&data[((size_t)hi << (lg_n - TB)) | (mid << TB) | lo]);
            if (pair) b[k] = *reinterpret_cast<const vec*>(&data[((size_t)hi << (lg_n - TB)) | (rmid << TB) | lo]);
        }
        #pragma unroll
        for (unsigned k = 0; k < PER; k++) {
            const unsigned s = tid + k * NT, hi = s / ROW, lo = (s % ROW) * V;
            #pragma unroll
            for (unsigned j = 0; j < V; j++) {
                ldsA[bitrev_lds_index<TB>(hi, lo + j)] = a[k].e[j];
                if (pair) ldsB[bitrev_lds_index<TB>(hi, lo + j)] = b[k].e[j];
            }
        }
    } else {
        #pragma unroll
        for (unsigned k = 0; k < PER; k++) {
            const unsigned s = tid + k * NT, r = s / ROW, q = (s % ROW) * V;             // destination row / first column
            const unsigned lo = bit_rev32(r, TB);
            vec a, b;
            #pragma unroll
            for (unsigned j = 0; j < V; j++) {
                const unsigned hi = bit_rev32(q + j, TB);
                a.e[j] = ldsA[bitrev_lds_index<TB>(hi, lo)];
                if (pair) b.e[j] = ldsB[bitrev_lds_index<TB>(hi, lo)];
            }
            *reinterpret_cast<vec*>(&data[((size_t)r << (lg_n - TB)) | (rmid << TB) | q]) = a;
            if (pair) *reinterpret_cast<vec*>(&data[((size_t)r << (lg_n - TB)) | (mid << TB) | q]) = b;
        }
    }
}
template<class F, unsigned TB>
__global__ __launch_bounds__(256) void k_bitrev_tiled_vec(F* data, unsigned lg_n)
{
    extern __shared__ unsigned char bitrev_lds[];
    F* ldsA = reinterpret_cast<F*>(bitrev_lds);
    F* ldsB = ldsA + (((1u << TB) + 1) << TB);
    bitrev_tile_vec_item<F, TB>(data, ldsA, ldsB, lg_n, blockIdx.x, threadIdx.x, 0);
    __syncthreads();
    bitrev_tile_vec_item<F, TB>(data, ldsA, ldsB, lg_n, blockIdx.x, threadIdx.x, 1);
}
// tile edge (bits): rows of >= 256 bytes, two tiles within 64 KB of LDS
template<class F> struct bitrev_tile_bits { static constexpr unsigned value = sizeof(F) <= 4 ? 6 : sizeof(F) <= 8 ? 5 : 4; };

// coset scaling a[i] *= g^(bitrev ? rev(i) : i)   (LDE_distribute_powers, ntt/kernels.cu:131-153)
template<class F>
SPPARK_DEVFN void coset_item(F* data, const ntt_tables<F>& G, int bitrev, size_t i)
{
    size_t e = i;
    if (bitrev) e = bit_rev32((unsigned)i, G.lg_n);
    data[i] = data[i] * ntt_twiddle(G, e);
}
template<class F>
__global__ __launch_bounds__(256) void k_coset(F* data, ntt_tables<F> G, int bitrev)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ((size_t)1 << G.lg_n)) coset_item(data, G, bitrev, i);
}

// LDE spread (LDE_spread_distribute_powers, ntt/kernels.cu:155-237): the 2^lg_domain
// inputs are in bit-reversed order; out[idx << lg_blowup] = in[idx] * g^(rev(idx))
// (the coset shift, when |shift|), every other element of out is zero.  One work
// item per OUTPUT element so the stores are coalesced.  out and in either do not overlap or in is
// aligned to the END of out (the reference's in-place form, ntt/ntt.cuh:358-360); the driver then
// launches the kernel over output ranges that never contain an input still to be read
// (ntt_driver.hpp lde_spread).
template<class F>
SPPARK_DEVFN void lde_spread_item(F* out, const F* in, const ntt_tables<F>& G, unsigned lg_domain,
                                  unsigned lg_blowup, int shift, size_t o)
{
    const size_t idx = o >> lg_blowup;
    F r = F();
    if ((o & (((size_t)1 << lg_blowup) - 1)) == 0) {
        r = in[idx];
        if (shift) {
            size_t e = 0;
            e = bit_rev32((unsigned)idx, lg_domain);
            r = r * ntt_twiddle(G, e);
        }
    }
    out[o] = r;
}
// output elements [o_begin, o_end)
template<class F>
__global__ __launch_bounds__(256)
void k_lde_spread(F* out, const F* in, ntt_tables<F> G, unsigned lg_domain, unsigned lg_blowup, int shift,
                  size_t o_begin, size_t o_end)
{
    size_t o = o_begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o < o_end) lde_spread_item(out, in, G, lg_domain, lg_blowup, shift, o);
}
// out[rev(i)] = in[i]  (out-of-place bit reversal: the aux output of LDE_aux, ntt/ntt.cuh:312-315)
template<class F>
__global__ __launch_bounds__(256) void k_bitrev_copy(F* out, const F* in, unsigned lg_n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << lg_n)) return;
    size_t r = 0;
    r = bit_rev32((unsigned)i, lg_n);
    out[r] = in[i];
}

// table generation: lo[k] = base^k (k < 2^h), hi[k] = (base^(2^h))^k (k < 2^(lg_n-h)),
// inner[(1 << R) + k] = w_{2^R}^k = base^(k << (lg_n - R)) for R <= min(cap, lg_n) (ntt_inner_entries)
template<class F>
SPPARK_DEVFN void table_item(F* lo, F* hi, F* inner, F base, unsigned lg_n, unsigned h, size_t k)
{
    if (k < ((size_t)1 << h)) lo[k] = field_pow(base, k);
    if (k < ((size_t)1 << (lg_n - h))) { F b = base; for (unsigned s = 0; s < h; s++) b = b * b; hi[k] = field_pow(b, k); }
    if (inner && k >= 2 && k < ntt_inner_entries<F>::value) {
        unsigned R = 31 - __builtin_clz((unsigned)k), e = (unsigned)k - (1u << R);
        inner[k] = R <= lg_n ? field_pow(base, (u64)e << (lg_n - R)) : F::one();
    }
}
template<class F>
__global__ __launch_bounds__(256) void k_tables(F* lo, F* hi, F* inner, F base, unsigned lg_n, unsigned h)
{   table_item(lo, hi, inner, base, lg_n, h, (size_t)blockIdx.x * blockDim.x + threadIdx.x);   }

// pass_tw[(mid << lgQ) + col] = w_{n_cur}^(col * rev_S(mid)), n_cur = 2^lg_cur, lgQ = lg_cur - S
// |scaled|: times T.scale (1/n of the inverse transform: every element of a tabled pass is multiplied by its entry
// exactly once, so one pass's table can carry the scaling of the whole transform)
template<class F>
SPPARK_DEVFN void pass_table_item(F* tw, const ntt_tables<F>& T, unsigned lg_cur, unsigned S, size_t i, int scaled = 0)
{
    const unsigned lgQ = lg_cur - S;
    if (i >= ((size_t)1 << lg_cur)) return;
    const size_t col = i & (((size_t)1 << lgQ) - 1), mid = i >> lgQ;
    F t = ntt_twiddle(T, (col * bit_rev32((unsigned)mid, S)) << (T.lg_n - lg_cur));
    tw[i] = scaled ? t * T.scale : t;
}
template<class F>
__global__ __launch_bounds__(256) void k_pass_table(F* tw, ntt_tables<F> T, unsigned lg_cur, unsigned S, int scaled)
{   pass_table_item(tw, T, lg_cur, S, (size_t)blockIdx.x * blockDim.x + threadIdx.x, scaled);   }

// ---- planning (host) ---------------------------------------------------------
struct ntt_plan { ntt_pass pass[16]; unsigned npass; };

// GS/DIF order (pass 0 splits the whole transform).  |lgCmax| = log2 of the
// elements in one 128-byte line, |lg_tile| = log2 of the LDS tile capacity.
static inline ntt_plan make_ntt_plan(unsigned lg_n, unsigned lgCmax, unsigned lg_tile, unsigned Smax = 8)
{
    ntt_plan pl; pl.npass = 0;
    unsigned np = (lg_n + Smax - 1) / Smax;
    unsigned rem = lg_n;
    for (unsigned i = 0; i < np; i++) {
        unsigned left = np - i;
        unsigned S = (rem + left - 1) / left;                   // near-equal split, <= 8
        ntt_pass p; p.lg_cur = rem; p.S = S; p.apply_scale = 0;
        p.cmode = 0; p.crow = p.cg_lo = p.cg_hi = nullptr; p.cg_h = 0;
        unsigned lgQ = rem - S;
        if (lgQ >= lgCmax) { p.lgC = lgCmax; p.lgG = 0; }
        else {
            p.lgC = lgQ;
            unsigned room = lg_tile > S + lgQ ? lg_tile - S - lgQ : 0, nsub = lg_n - rem;  // log2 sub-problems available
            p.lgG = room < nsub ? room : nsub;
        }
        pl.pass[pl.npass++] = p;
        rem -= S;
    }
    return pl;
}

// The plan of the one-stage-per-round passes (k_ntt_pass_lat; 256-bit fields): passes of <= smax stages; tiles of
// 2^7 .. 2^10 elements, at least 256 of them from 2^16 elements on -- a small transform is latency-bound and
// wants every CU, a large one wants 128-byte rows (four 32-byte elements).  lgc / lgtile >= 0 override the shape
// (SPPARK_NTT_LAT_LGC / _LGTILE; profiles/r04_ntt_wide_lat_planes.log).
static inline ntt_plan make_ntt_lat_plan(unsigned lg_n, unsigned smax, int lgc = -1, int lgtile = -1)
{
    const unsigned np = (lg_n + smax - 1) / smax, s0 = (lg_n + np - 1) / np;
    const unsigned tile_lg = lgtile >= 0 ? (unsigned)lgtile : lg_n < 15 ? 7u : lg_n > 18 ? 10u : lg_n - 8;
    const unsigned c_lg = lgc >= 0 ? (unsigned)lgc : tile_lg > s0 ? (tile_lg - s0 < 2u ? tile_lg - s0 : 2u) : 0u;
    return make_ntt_plan(lg_n, c_lg, tile_lg > s0 + c_lg ? tile_lg : s0 + c_lg, smax);
}

} // namespace sppark_amd
