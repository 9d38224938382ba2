// MI355X NTT kernels for single-word fields (Goldilocks, BabyBear).
//
// Same observable semantics as the reference's driver (ntt/ntt.cuh:161-213:
// NN = bit_rev + CT, NR = GS, RN = CT, RR = GS + bit_rev; coset scaling of
// ntt/kernels.cu:131-153) but a different decomposition: the transform is cut
// into PASSES, each pass is a batch of independent 2^S-point transforms done
// entirely in LDS on a tile of [2^S rows] x [C adjacent columns] (C*sizeof(F)
// = one 128-byte line, so the strided passes stay coalesced), followed (GS) or
// preceded (CT) by ONE diagonal twiddle multiplication per element:
//
//   GS/DIF pass on a sub-problem of size n_cur = 2^S * Q, element (mid, lo):
//        y[mid][lo] = DIF_{2^S}(x[.][lo])[mid] * w_{n_cur}^(lo * rev_S(mid))
//   and the next pass works on the Q-sized rows independently.  CT/DIT is the
//   transposed network: passes in reverse order, twiddle first.
//
// Twiddles w_n^e come from two tables of <= 2^12 entries each (w^e_lo, w^(e_hi<<h))
// instead of the reference's four 7-bit windows (ntt/parameters.cuh:72-145).
//
// Each phase is a SPPARK_DEVFN function of (tid, nthreads) so that the host
// emulation harness (tests/emu) can run the same index math on the CPU.
#pragma once
#include "../ff/small_fields_dev.hpp"

namespace sppark_amd {

template<class F> struct ntt_tables {
    const F* lo;        // w^k,            k < 2^h
    const F* hi;        // w^(k << h),     k < 2^(lg_n - h)
    unsigned lg_n, h;
    F scale;            // 1/n for the inverse transform (Montgomery form where applicable)
};

struct ntt_pass {
    unsigned lg_cur;    // log2 of the sub-problem size this pass splits
    unsigned S;         // stages done in LDS
    unsigned lgC;       // log2 columns (lo) per tile
    unsigned lgG;       // log2 sub-problems per tile (only when the tile spans all Q columns)
    int apply_scale;    // multiply by tables.scale when storing (inverse, last executed pass)
};

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

// inner[k] = w_{2^S}^k, k < 2^(S-1)
template<class F>
SPPARK_DEVFN void ntt_phase_inner_table(F* inner, const ntt_tables<F>& T, unsigned S, unsigned tid, unsigned nt)
{
    for (unsigned k = tid; k < (1u << (S - 1)); k += nt)
        inner[k] = ntt_twiddle(T, (size_t)k << (T.lg_n - S));
}

template<class F, bool DIF>
SPPARK_DEVFN void ntt_phase_load(F* tile, const F* data, const ntt_tables<F>& T, const ntt_pass& P,
                                 size_t tile_id, unsigned tid, unsigned nt)
{
    const unsigned lgQ = P.lg_cur - P.S, S = P.S;
    const unsigned E = 1u << (P.lgG + S + P.lgC), C = 1u << P.lgC;
    size_t row0; unsigned c0;                           // first (sub*2^S + mid) row, first column
    if (P.lgG) { row0 = (tile_id << P.lgG) << S; c0 = 0; }
    else { size_t tiles_per_sub = (size_t)1 << (lgQ - P.lgC); row0 = (tile_id / tiles_per_sub) << S; c0 = (unsigned)(tile_id % tiles_per_sub) << P.lgC; }
    for (unsigned e = tid; e < E; e += nt) {
        unsigned c = e & (C - 1), gm = e >> P.lgC;
        F v = data[((row0 + gm) << lgQ) + c0 + c];
        if (!DIF && lgQ) {                              // CT: diagonal twiddle before the local transform
            unsigned mid = gm & ((1u << S) - 1);
            size_t ex = (size_t)(c0 + c) * bit_rev32(mid, S);
            v = v * ntt_twiddle(T, ex << (T.lg_n - P.lg_cur));
        }
        tile[e] = v;
    }
}

template<class F, bool DIF>
SPPARK_DEVFN void ntt_phase_stage(F* tile, const F* inner, const ntt_pass& P, unsigned t, unsigned tid, unsigned nt)
{
    const unsigned S = P.S, C = 1u << P.lgC;
    const unsigned nb = 1u << (P.lgG + S - 1 + P.lgC);
    for (unsigned bf = tid; bf < nb; bf += nt) {
        unsigned c = bf & (C - 1), r = bf >> P.lgC;
        unsigned g = r >> (S - 1), j = r & ((1u << (S - 1)) - 1);
        unsigned lgh = DIF ? S - 1 - t : t;             // log2 of the butterfly span
        unsigned half = 1u << lgh, off = j & (half - 1), blk = j >> lgh;
        unsigned m0 = (blk << (lgh + 1)) + off, m1 = m0 + half;
        unsigned i0 = (((g << S) + m0) << P.lgC) + c, i1 = (((g << S) + m1) << P.lgC) + c;
        F w = inner[off << (S - 1 - lgh)];
        F a = tile[i0], b = tile[i1];
        if (DIF) { tile[i0] = a + b; tile[i1] = (a - b) * w; }
        else     { F bw = b * w; tile[i0] = a + bw; tile[i1] = a - bw; }
    }
}

template<class F, bool DIF>
SPPARK_DEVFN void ntt_phase_store(F* data, const F* tile, const ntt_tables<F>& T, const ntt_pass& P,
                                  size_t tile_id, unsigned tid, unsigned nt)
{
    const unsigned lgQ = P.lg_cur - P.S, S = P.S;
    const unsigned E = 1u << (P.lgG + S + P.lgC), C = 1u << P.lgC;
    size_t row0; unsigned c0;
    if (P.lgG) { row0 = (tile_id << P.lgG) << S; c0 = 0; }
    else { size_t tiles_per_sub = (size_t)1 << (lgQ - P.lgC); row0 = (tile_id / tiles_per_sub) << S; c0 = (unsigned)(tile_id % tiles_per_sub) << P.lgC; }
    for (unsigned e = tid; e < E; e += nt) {
        unsigned c = e & (C - 1), gm = e >> P.lgC;
        F v = tile[e];
        if (DIF && lgQ) {                               // GS: diagonal twiddle after the local transform
            unsigned mid = gm & ((1u << S) - 1);
            size_t ex = (size_t)(c0 + c) * bit_rev32(mid, S);
            v = v * ntt_twiddle(T, ex << (T.lg_n - P.lg_cur));
        }
        if (P.apply_scale) v = v * T.scale;
        data[((row0 + gm) << lgQ) + c0 + c] = v;
    }
}

template<class F, bool DIF>
__global__ __launch_bounds__(256)
void k_ntt_pass(F* data, ntt_tables<F> T, ntt_pass P)
{
    extern __shared__ unsigned char ntt_lds[];
    F* tile = reinterpret_cast<F*>(ntt_lds);
    F* inner = tile + ((size_t)1 << (P.lgG + P.S + P.lgC));
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    ntt_phase_inner_table(inner, T, P.S, tid, nt);
    ntt_phase_load<F, DIF>(tile, data, T, P, blockIdx.x, tid, nt);
    __syncthreads();
    for (unsigned t = 0; t < P.S; t++) {
        ntt_phase_stage<F, DIF>(tile, inner, P, t, tid, nt);
        __syncthreads();
    }
    ntt_phase_store<F, DIF>(data, tile, T, P, blockIdx.x, tid, nt);
}

// bit-reversal permutation in place (NN and RR orders; ntt/ntt.cuh:44-79)
template<class F>
SPPARK_DEVFN void bitrev_item(F* data, unsigned lg_n, size_t i)
{
    size_t r = 0;
    for (unsigned k = 0; k < lg_n; k++) r |= ((i >> k) & 1) << (lg_n - 1 - k);
    if (i < r) { F t = data[i]; data[i] = data[r]; data[r] = t; }
}
template<class F>
__global__ __launch_bounds__(256) void k_bitrev(F* data, unsigned lg_n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ((size_t)1 << lg_n)) bitrev_item(data, lg_n, i);
}

// coset scaling a[i] *= g^(bitrev ? rev(i) : i)   (LDE_distribute_powers, ntt/kernels.cu:131-153)
template<class F>
SPPARK_DEVFN void coset_item(F* data, const ntt_tables<F>& G, int bitrev, size_t i)
{
    size_t e = i;
    if (bitrev) { e = 0; for (unsigned k = 0; k < G.lg_n; k++) e |= ((i >> k) & 1) << (G.lg_n - 1 - k); }
    data[i] = data[i] * ntt_twiddle(G, e);
}
template<class F>
__global__ __launch_bounds__(256) void k_coset(F* data, ntt_tables<F> G, int bitrev)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ((size_t)1 << G.lg_n)) coset_item(data, G, bitrev, i);
}

// table generation: lo[k] = base^k (k < 2^h), hi[k] = (base^(2^h))^k (k < 2^(lg_n-h))
template<class F>
SPPARK_DEVFN void table_item(F* lo, F* hi, F base, unsigned lg_n, unsigned h, size_t k)
{
    if (k < ((size_t)1 << h)) lo[k] = field_pow(base, k);
    if (k < ((size_t)1 << (lg_n - h))) { F b = base; for (unsigned s = 0; s < h; s++) b = b * b; hi[k] = field_pow(b, k); }
}
template<class F>
__global__ __launch_bounds__(256) void k_tables(F* lo, F* hi, F base, unsigned lg_n, unsigned h)
{   table_item(lo, hi, base, lg_n, h, (size_t)blockIdx.x * blockDim.x + threadIdx.x);   }

// ---- planning (host) ---------------------------------------------------------
struct ntt_plan { ntt_pass pass[8]; unsigned npass; };

// GS/DIF order (pass 0 splits the whole transform).  |lgCmax| = log2 of the
// elements in one 128-byte line, |lg_tile| = log2 of the LDS tile capacity.
static inline ntt_plan make_ntt_plan(unsigned lg_n, unsigned lgCmax, unsigned lg_tile)
{
    ntt_plan pl; pl.npass = 0;
    const unsigned Smax_strided = lg_tile - lgCmax, Smax_last = lg_tile;
    unsigned np = 1;
    if (lg_n > Smax_last) { np = 2; while ((np - 1) * Smax_strided + Smax_last < lg_n) np++; }
    unsigned rem = lg_n;
    for (unsigned i = 0; i < np; i++) {
        unsigned left = np - i;
        unsigned S = (rem + left - 1) / left;                   // near-equal split
        if (left > 1 && S > Smax_strided) S = Smax_strided;
        if (left == 1) S = rem;
        ntt_pass p; p.lg_cur = rem; p.S = S; p.apply_scale = 0;
        unsigned lgQ = rem - S;
        if (lgQ >= lgCmax) { p.lgC = lgCmax; p.lgG = 0; }
        else {
            p.lgC = lgQ;
            unsigned room = lg_tile - S - lgQ, nsub = lg_n - rem;  // log2 sub-problems available
            p.lgG = room < nsub ? room : nsub;
        }
        pl.pass[pl.npass++] = p;
        rem -= S;
    }
    return pl;
}

} // namespace sppark_amd
