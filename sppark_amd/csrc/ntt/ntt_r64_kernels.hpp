// Radix-64 plan of the single-word NTT (Goldilocks, BabyBear) for transforms of >= 2^12 elements.
//
// Why.  The 8-stage passes of ntt_kernels.hpp (radix-16 x radix-16 in registers) need a GENERAL
// twiddle inside every pass and one between passes: 2^24 = 8 + 8 + 8 has five twiddle layers, each a
// 64x64-bit product + 128-bit reduction per element for Goldilocks (~28 vector instructions; they
// were half of the pass's instruction count, profiles/r02_ntt_gl64_pmc.txt).  Every Goldilocks root
// of order <= 64 is a power of two (w_64 = 2^39, ntt/parameters/goldilocks.h:86-93 in the reference),
// so a 64-point block -- radix-8 in registers, one trip through LDS, radix-8 again -- needs no general
// product at all: the diagonal w_64^(b*rev(a)) between its two rounds is a shift-and-fold too.  With
// blocks of 6 stages 2^24 = 6 + 6 + (6 + 6) has THREE general layers.  The last 12 stages stay in one
// launch (a 4096-element sub-problem is contiguous: 32 KB of LDS for Goldilocks), so HBM is still read
// and written three times per transform; the reference's narrow kernels do one radix-2 stage per
// shared-memory step (ntt/kernels/gs_mixed_radix_narrow.cu:5-185).
//
//   k_ntt6   6 stages of a sub-problem of 2^lg_cur = 64 * Q elements (Q >= 64): tile = [64 rows at
//            stride Q] x [64 adjacent columns] (512-byte row segments), 512 lanes x 8 elements;
//            round "high" keeps the 8 values of a (row = 8a + b) in registers, round "low" those of b;
//            the inter-pass twiddle w_{n_cur}^(col * rev6(row)) comes from one table when that is small
//            (sub-problems of <= 2^20 elements: L2 / Infinity-Cache resident) and otherwise from two
//            small ones, W^(c0*r) (uniform over the work-group: scalar loads) x W^(c*r) (4096 entries),
//            c0 = first column of the tile -- one product more per element instead of a table as large
//            as the data (which would make the pass HBM-bound);
//   k_ntt12  the last 12 stages on contiguous 4096-element sub-problems: four radix-8 rounds over the
//            octal digits (a1, b1, a2, b2) of the position, three trips through LDS, shift diagonals
//            after a1 and a2, ONE general twiddle w_4096^(lo * rev6(hi)) after b1 (a 4096-entry table
//            read with the data's own addressing).  The LDS image is XOR-swizzled so that all four
//            access patterns are conflict-free.
//
// The lane-variable factor of a shift diagonal is made WAVE-UNIFORM by the lane mapping (b = tid >> 6)
// and dispatched by a switch, so that every shift amount is a compile-time constant: a shift by a
// lane-variable amount costs as much as a general product.  The scale 1/n of the inverse transform is
// folded into the table of the last executed pass.
//
// Index conventions are those of ntt_kernels.hpp (in-place GS/DIF: natural in, bit-reversed out;
// CT/DIT is the transposed network, twiddle first); a k_ntt12 launch equals two consecutive 6-stage
// passes of that file, which is what tests/emu and the GPU tests check it against.
#pragma once
#include "ntt_kernels.hpp"

namespace sppark_amd {

template<class F> struct ntt_r64_args {
    const F* inner;     // ntt_tables::inner (in-register roots of fields without shift roots)
    const F* tw;        // k_ntt6: tw[(row << lgQ) + col] = W^(col * rev6(row)) or null; k_ntt12: tw12[pos]
    const F* t1;        // k_ntt6 without tw: t1[c0 + row] = W^(c0 * rev6(row)), c0 = col & ~63
    const F* t2;        //                    t2[(row << 6) + c] = W^(c * rev6(row)), c = col & 63
    unsigned lg_cur;    // k_ntt6: log2 of the sub-problem size (>= 12)
    // coset transforms folded into the passes (r64_coset_mode below), or null: 64 constants g^(x << (lg_n - 6)).
    // k_ntt6 (the step on the whole transform): cz[row], applied to the rows when they are loaded (DIF) / stored (DIT);
    // k_ntt12: cz[lo] = g^(rev6(lo) << (lg_n - 6)), lo = position & 63, applied when the block is stored (DIF) / loaded (DIT).
    const F* cz;
    // k_ntt12, first step of a forward RN transform inside sppark_lde (LDE template flag): the block is read from the COMPACT
    // 2^lde_lgd coefficients (bit-reversed order) instead of the spread array -- position p holds lde_src[p >> lde_lgb] times
    // g^(rev(p >> lde_lgb)) when its low lde_lgb <= 3 bits are zero, and zero otherwise (k_lde_spread's output, never written)
    F* out = nullptr;   // k_ntt12 as the LAST step (DIF): the blocks are stored here instead of in place (sppark_lde's inverse transform leaves
                        // the coefficients in its scratch buffer without a copy before it)
    const F* lde_src = nullptr;
    const F* lde_glo = nullptr; const F* lde_ghi = nullptr;     // the powers of g: ntt_tables lo / hi / h of the domain's size
    unsigned lde_gh = 0, lde_lgd = 0, lde_lgb = 0;
};

SPPARK_DEVFN unsigned wave_uniform(unsigned v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
}

SPPARK_DEVFN constexpr unsigned rev3c(unsigned a) { return ((a & 1u) << 2) | (a & 2u) | (a >> 2); }

// x[a] *= w_64^(B * rev3(a)): the diagonal between the two radix-8 rounds of a 64-point block
template<class F, bool INV, unsigned B>
SPPARK_DEVFN void tw64_mul(F* x, const F* inner)
{
    #pragma unroll
    for (unsigned a = 1; a < 8; a++)
        x[a] = F::template mul_root_full<INV>(x[a], 6, B * rev3c(a), inner);
}
template<class F, bool INV>
SPPARK_DEVFN void tw64_layer(F* x, unsigned b, const F* inner)      // b uniform over the wave
{
    switch (b) {
        case 1: tw64_mul<F, INV, 1>(x, inner); break;
        case 2: tw64_mul<F, INV, 2>(x, inner); break;
        case 3: tw64_mul<F, INV, 3>(x, inner); break;
        case 4: tw64_mul<F, INV, 4>(x, inner); break;
        case 5: tw64_mul<F, INV, 5>(x, inner); break;
        case 6: tw64_mul<F, INV, 6>(x, inner); break;
        case 7: tw64_mul<F, INV, 7>(x, inner); break;
        default: break;
    }
}

// ---- k_ntt6 -----------------------------------------------------------------------------------
struct ntt6_geom { size_t row0; unsigned c0, lgQ; };
SPPARK_DEVFN ntt6_geom ntt6_tile(unsigned lg_cur, size_t tile_id)
{
    ntt6_geom g; g.lgQ = lg_cur - 6;
    const unsigned lgT = g.lgQ - 6;                                 // log2 tiles per sub-problem
    g.row0 = (tile_id >> lgT) << 6;
    g.c0 = (unsigned)(tile_id & (((size_t)1 << lgT) - 1)) << 6;
    return g;
}

// lanes (c, b), registers over a.  DIF: HBM -> DIF8 -> * w_64^(b*rev3(a)) -> LDS;  DIT: LDS -> * w_64 -> DIT8 -> HBM
template<class F, bool DIF, bool INV>
SPPARK_DEVFN void ntt6_high(F* data, F* tile, const ntt_r64_args<F>& A, size_t tile_id, unsigned tid)
{
    const ntt6_geom g = ntt6_tile(A.lg_cur, tile_id);
    const unsigned c = tid & 63, b = wave_uniform(tid >> 6);
    F* base = data + ((g.row0 + b) << g.lgQ) + g.c0 + c;
    const size_t stride = (size_t)8 << g.lgQ;
    F x[8];
    #pragma unroll
    for (unsigned a = 0; a < 8; a++) x[a] = DIF ? base[a * stride] : tile[(((a << 3) + b) << 6) + c];
    if (DIF) {
        if (A.cz != nullptr) {                                      // coset: row (8a + b) times g^(row * Q); b is wave-uniform: scalar loads
            const F* cz = A.cz + b;
            #pragma unroll
            for (unsigned a = 0; a < 8; a++) x[a] = x[a] * cz[a << 3];
        }
        radix_dif<F, INV, 3>(x, A.inner);
        tw64_layer<F, INV>(x, b, A.inner);
        #pragma unroll
        for (unsigned a = 0; a < 8; a++) tile[(((a << 3) + b) << 6) + c] = x[a];
    } else {
        tw64_layer<F, INV>(x, b, A.inner);
        radix_dit<F, INV, 3>(x, A.inner);
        if (A.cz != nullptr) {
            const F* cz = A.cz + b;
            #pragma unroll
            for (unsigned a = 0; a < 8; a++) x[a] = x[a] * cz[a << 3];
        }
        #pragma unroll
        for (unsigned a = 0; a < 8; a++) base[a * stride] = x[a];
    }
}

// lanes (c, a), registers over b.  DIF: LDS -> DIF8 -> * W^(col*rev6(row)) -> HBM;  DIT: HBM -> * W -> DIT8 -> LDS
template<class F, bool DIF, bool INV>
SPPARK_DEVFN void ntt6_low(F* data, F* tile, const ntt_r64_args<F>& A, size_t tile_id, unsigned tid)
{
    const ntt6_geom g = ntt6_tile(A.lg_cur, tile_id);
    const unsigned c = tid & 63, a = wave_uniform(tid >> 6);
    F* base = data + ((g.row0 + (a << 3)) << g.lgQ) + g.c0 + c;
    const size_t stride = (size_t)1 << g.lgQ;
    // (the entry of row 0 is not skipped: it carries the scale 1/n when this is an inverse transform's last pass)
    F tw[8];
    if (A.tw != nullptr) {                                          // uniform over the launch
        const F* twp = A.tw + ((size_t)(a << 3) << g.lgQ) + g.c0 + c;
        #pragma unroll
        for (unsigned b = 0; b < 8; b++) tw[b] = twp[b * stride];
    } else {
        const F* t1p = A.t1 + g.c0 + (a << 3);                      // uniform address: scalar loads
        const F* t2p = A.t2 + ((a << 3) << 6) + c;
        #pragma unroll
        for (unsigned b = 0; b < 8; b++) tw[b] = t1p[b] * t2p[b << 6];
    }
    F x[8];
    #pragma unroll
    for (unsigned b = 0; b < 8; b++) x[b] = DIF ? tile[(((a << 3) + b) << 6) + c] : base[b * stride] * tw[b];
    if (DIF) {
        radix_dif<F, INV, 3>(x, A.inner);
        #pragma unroll
        for (unsigned b = 0; b < 8; b++) base[b * stride] = x[b] * tw[b];
    } else {
        radix_dit<F, INV, 3>(x, A.inner);
        #pragma unroll
        for (unsigned b = 0; b < 8; b++) tile[(((a << 3) + b) << 6) + c] = x[b];
    }
}

template<class F, bool DIF, bool INV>
__global__ __launch_bounds__(512)
void k_ntt6(F* data, ntt_r64_args<F> A)
{
    extern __shared__ unsigned char ntt_lds[];
    F* tile = reinterpret_cast<F*>(ntt_lds);
    if (DIF) {
        ntt6_high<F, true, INV>(data, tile, A, blockIdx.x, threadIdx.x);
        __syncthreads();
        ntt6_low<F, true, INV>(data, tile, A, blockIdx.x, threadIdx.x);
    } else {
        ntt6_low<F, false, INV>(data, tile, A, blockIdx.x, threadIdx.x);
        __syncthreads();
        ntt6_high<F, false, INV>(data, tile, A, blockIdx.x, threadIdx.x);
    }
}

// ---- k_ntt12 ----------------------------------------------------------------------------------
// position p = a1 | b1 | a2 | b2 (three bits each, a1 on top).  Physical LDS slot: the low six bits
// XORed with the high six, so that the rounds whose lanes differ in the HIGH bits (a2, b2) also hit
// 32 different 8-byte banks per half-wave.
SPPARK_DEVFN unsigned ntt12_phys(unsigned p) { return p ^ ((p >> 6) & 63u); }

enum { R12_A1 = 0, R12_B1 = 1, R12_A2 = 2, R12_B2 = 3 };
// position of register r of lane |lane| in the round that keeps digit D in registers
template<int D> SPPARK_DEVFN unsigned ntt12_pos(unsigned lane, unsigned r)
{
    if (D == R12_A1) return (r << 9) | lane;                                        // b1 = lane >> 6: wave-uniform
    if (D == R12_B1) return ((lane >> 6) << 9) | (r << 6) | (lane & 63);
    if (D == R12_A2) return ((lane & 63) << 6) | (r << 3) | (lane >> 6);            // b2 = lane >> 6: wave-uniform
    // b2 in registers: a lane owns 8 consecutive positions; the lane bits are spread over p >> 3 so
    // that a half-wave touches 32 different banks (p[4:3] and p[8:6] vary within it)
    const unsigned q = ((lane >> 6) << 6) | (((lane >> 2) & 7u) << 3) | (((lane >> 5) & 1u) << 2) | (lane & 3u);
    return (q << 3) | r;
}

template<class F> struct alignas(16) ntt_vec16 { F v[16 / sizeof(F)]; };

// One round.  DIF runs A1, B1, A2, B2 (HBM -> ... -> HBM); DIT runs them in the opposite order with
// the diagonal applied BEFORE the butterflies.
template<class F, bool DIF, bool INV, int D, bool LDE = false>
SPPARK_DEVFN void ntt12_round(F* sub, F* tile, const ntt_r64_args<F>& A, unsigned lane, size_t blk = 0, F* osub = nullptr)
{
    if (osub == nullptr) osub = sub;                                // where a round that stores to memory puts the block
    constexpr bool from_hbm = DIF ? D == R12_A1 : D == R12_B2;
    constexpr bool to_hbm   = DIF ? D == R12_B2 : D == R12_A1;
    constexpr unsigned PER = 16 / sizeof(F);                        // elements per 16-byte access
    const unsigned u = wave_uniform(lane >> 6);
    F x[8];
    if (LDE && D == R12_B2 && from_hbm) {                           // the spread array as it would have been (8 consecutive positions)
        const size_t p0 = (blk << 12) + ntt12_pos<D>(lane, 0);
        const unsigned B = 1u << A.lde_lgb;
        #pragma unroll
        for (unsigned r = 0; r < 8; r++) {
            x[r] = F();
            if ((r & (B - 1)) == 0) {
                const size_t idx = (p0 + r) >> A.lde_lgb;
                const size_t e = bit_rev32((unsigned)idx, A.lde_lgd);
                x[r] = A.lde_src[idx] * (A.lde_glo[e & (((size_t)1 << A.lde_gh) - 1)] * A.lde_ghi[e >> A.lde_gh]);
            }
        }
    } else if (D == R12_B2 && from_hbm) {                           // 8 consecutive elements: 16-byte loads
        const ntt_vec16<F>* src = reinterpret_cast<const ntt_vec16<F>*>(sub + ntt12_pos<D>(lane, 0));
        #pragma unroll
        for (unsigned k = 0; k < 8 / PER; k++) {
            ntt_vec16<F> q = src[k];
            #pragma unroll
            for (unsigned j = 0; j < PER; j++) x[k * PER + j] = q.v[j];
        }
    } else {
        #pragma unroll
        for (unsigned r = 0; r < 8; r++) {
            const unsigned p = ntt12_pos<D>(lane, r);
            x[r] = from_hbm ? sub[p] : tile[ntt12_phys(p)];
        }
    }
    if (D == R12_B2 && from_hbm && A.cz != nullptr) {               // coset, DIT: the block's input times g^(rev6(lo) << (lg_n - 6))
        const F* cz = A.cz + (ntt12_pos<D>(lane, 0) & 63u);
        #pragma unroll
        for (unsigned r = 0; r < 8; r++) x[r] = x[r] * cz[r];
    }
    if (!DIF) {
        if (D == R12_A1 || D == R12_A2) tw64_layer<F, INV>(x, u, A.inner);
        if (D == R12_B1) {
            #pragma unroll
            for (unsigned r = 0; r < 8; r++) x[r] = x[r] * A.tw[ntt12_pos<D>(lane, r)];
        }
        radix_dit<F, INV, 3>(x, A.inner);
    } else {
        radix_dif<F, INV, 3>(x, A.inner);
        if (D == R12_A1 || D == R12_A2) tw64_layer<F, INV>(x, u, A.inner);
        if (D == R12_B1) {
            #pragma unroll
            for (unsigned r = 0; r < 8; r++) x[r] = x[r] * A.tw[ntt12_pos<D>(lane, r)];
        }
    }
    if (D == R12_B2 && to_hbm && A.cz != nullptr) {                 // coset, DIF: the block's output likewise
        const F* cz = A.cz + (ntt12_pos<D>(lane, 0) & 63u);
        #pragma unroll
        for (unsigned r = 0; r < 8; r++) x[r] = x[r] * cz[r];
    }
    if (D == R12_B2 && to_hbm) {
        ntt_vec16<F>* dst = reinterpret_cast<ntt_vec16<F>*>(osub + ntt12_pos<D>(lane, 0));
        #pragma unroll
        for (unsigned k = 0; k < 8 / PER; k++) {
            ntt_vec16<F> q;
            #pragma unroll
            for (unsigned j = 0; j < PER; j++) q.v[j] = x[k * PER + j];
            dst[k] = q;
        }
    } else {
        #pragma unroll
        for (unsigned r = 0; r < 8; r++) {
            const unsigned p = ntt12_pos<D>(lane, r);
            if (to_hbm) osub[p] = x[r]; else tile[ntt12_phys(p)] = x[r];
        }
    }
}

template<class F, bool DIF, bool INV, bool LDE = false>
__global__ __launch_bounds__(512)
void k_ntt12(F* data, ntt_r64_args<F> A)
{
    extern __shared__ unsigned char ntt_lds[];
    F* tile = reinterpret_cast<F*>(ntt_lds);
    F* sub = data + ((size_t)blockIdx.x << 12);
    const unsigned lane = threadIdx.x;
    if (LDE) {                                                      // (forward DIT only: the driver's sppark_lde path)
        ntt12_round<F, false, INV, R12_B2, true>(sub, tile, A, lane, blockIdx.x); __syncthreads();
        ntt12_round<F, false, INV, R12_A2>(sub, tile, A, lane); __syncthreads();
        ntt12_round<F, false, INV, R12_B1>(sub, tile, A, lane); __syncthreads();
        ntt12_round<F, false, INV, R12_A1>(sub, tile, A, lane);
    } else if (DIF) {
        ntt12_round<F, true, INV, R12_A1>(sub, tile, A, lane); __syncthreads();
        ntt12_round<F, true, INV, R12_B1>(sub, tile, A, lane); __syncthreads();
        ntt12_round<F, true, INV, R12_A2>(sub, tile, A, lane); __syncthreads();
        ntt12_round<F, true, INV, R12_B2>(sub, tile, A, lane, 0, A.out ? A.out + ((size_t)blockIdx.x << 12) : nullptr);
    } else {
        ntt12_round<F, false, INV, R12_B2>(sub, tile, A, lane); __syncthreads();
        ntt12_round<F, false, INV, R12_A2>(sub, tile, A, lane); __syncthreads();
        ntt12_round<F, false, INV, R12_B1>(sub, tile, A, lane); __syncthreads();
        ntt12_round<F, false, INV, R12_A1>(sub, tile, A, lane);
    }
}

// ---- tables ------------------------------------------------------------------------------------
// kind 0: tw[(row << lgQ) + col] = W^(col * rev6(row))         (2^lg_cur entries; k_ntt12: lg_cur = 12)
// kind 1: t1[c0 + row]           = W^(c0 * rev6(row)), c0 = 64-aligned column   (2^(lg_cur - 6) entries)
// kind 2: t2[(row << 6) + c]     = W^(c * rev6(row))                            (4096 entries)
// W = w_n^(n / n_cur); |scaled|: times 1/n (the inverse transform's last executed pass).
//
// Coset transforms (ntt/ntt.cuh:197-207: the powers of the coset generator as a separate kernel before a forward / after an
// inverse transform, LDE_distribute_powers ntt/kernels.cu:131-153) are FOLDED into the passes where the plan is made of k_ntt6 /
// k_ntt12 steps only.  The exponent of g splits along the index digits the passes work on, and all but 64 constants of it
// land in twiddle tables that are multiplied anyway:
//   cmode 1, "natural exponents" -- forward DIF (x_i g^i on the natural input) and inverse DIT (X_k g^-k on the natural
//     output): with i = row Q + col on the step over the whole transform, g^(row Q) is one of 64 constants per ROW (cz, the
//     one product per element this costs) and g^col joins that step's inter-pass twiddle W^(col rev6(row)): kind 0 times
//     g^col, kind 1 times g^c0, kind 2 times g^c.  The later steps are plain transforms.
//   cmode 2, "bit-reversed exponents" -- inverse DIF (the output position p holds index rev(p)) and forward DIT (the input
//     position likewise): rev(p) = sum over the steps of rev6(row_s) << (lg_n - lg_cur_s) plus, inside a 4096-block,
//     rev6(hi) << (lg_n - 12) and rev6(lo) << (lg_n - 6): every step's table (kind 0 and 1; kind 2 has no row-only part)
//     takes g^(rev6(row) << (lg_n - lg_cur)), k_ntt12's table the same with row = hi, and the lo part is 64 constants per
//     block position (cz), one product per element in k_ntt12.
// G: the powers of g (of 1/g for an inverse transform), as k_coset uses them.
template<class F>
SPPARK_DEVFN void r64_table_item(F* out, const ntt_tables<F>& T, unsigned kind, unsigned lg_cur, int scaled, size_t i,
                                 unsigned cmode = 0, const ntt_tables<F>* G = nullptr)
{
    const unsigned lgQ = lg_cur - 6;
    size_t count, col; unsigned row;
    if (kind == 0)      { count = (size_t)1 << lg_cur; row = (unsigned)(i >> lgQ); col = i & (((size_t)1 << lgQ) - 1); }
    else if (kind == 1) { count = (size_t)1 << lgQ;    row = (unsigned)(i & 63);   col = i & ~(size_t)63; }
    else                { count = 4096;                row = (unsigned)(i >> 6);   col = i & 63; }
    if (i >= count) return;
    F w = ntt_twiddle(T, (col * bit_rev32(row, 6)) << (T.lg_n - lg_cur));
    if (scaled) w = w * T.scale;
    if (cmode == 1) w = w * ntt_twiddle(*G, col);
    if (cmode == 2 && kind != 2) w = w * ntt_twiddle(*G, (size_t)bit_rev32(row, 6) << (T.lg_n - lg_cur));
    out[i] = w;
}
template<class F>
__global__ __launch_bounds__(256) void k_r64_table(F* out, ntt_tables<F> T, unsigned kind, unsigned lg_cur, int scaled, unsigned cmode, ntt_tables<F> G)
{   r64_table_item(out, T, kind, lg_cur, scaled, (size_t)blockIdx.x * blockDim.x + threadIdx.x, cmode, &G);   }

// the 64 constants of a folded coset transform: cz[x] = g^(x << (lg_n - 6)) (cmode 1) / g^(rev6(x) << (lg_n - 6)) (cmode 2)
template<class F>
SPPARK_DEVFN void r64_cz_item(F* out, const ntt_tables<F>& G, unsigned cmode, unsigned x)
{
    if (x >= 64) return;
    out[x] = ntt_twiddle(G, (size_t)(cmode == 2 ? bit_rev32(x, 6) : x) << (G.lg_n - 6));
}
template<class F>
__global__ __launch_bounds__(64) void k_r64_cz(F* out, ntt_tables<F> G, unsigned cmode)
{   r64_cz_item(out, G, cmode, threadIdx.x);   }
// the 2^S row constants of a generic top pass (ntt_pass::crow): g^(row << (lg_n - S)) (cmode 1) / g^(rev_S(row)) (cmode 2)
template<class F>
SPPARK_DEVFN void pass_crow_item(F* out, const ntt_tables<F>& G, unsigned cmode, unsigned S, unsigned row)
{
    if (row >= (1u << S)) return;
    out[row] = ntt_twiddle(G, cmode == 2 ? (size_t)bit_rev32(row, S) : (size_t)row << (G.lg_n - S));
}
template<class F>
__global__ __launch_bounds__(256) void k_pass_crow(F* out, ntt_tables<F> G, unsigned cmode, unsigned S)
{   pass_crow_item(out, G, cmode, S, threadIdx.x);   }

// ---- planning (host) -----------------------------------------------------------------------------
// GS/DIF order (step 0 splits the whole transform); CT/DIT executes the steps in reverse.
struct r64_step { int kind; unsigned lg_cur, S; };                  // kind 0: k_ntt_pass (S <= 8), 1: k_ntt6, 2: k_ntt12
struct r64_plan { r64_step step[8]; unsigned nsteps; };

static inline r64_plan make_r64_plan(unsigned lg_n)                  // lg_n >= 12
{
    r64_plan pl; pl.nsteps = 0;
    unsigned rem = lg_n - 12, sizes[8], np = (rem + 7) / 8;
    if (np) {
        const unsigned last = rem - 6 * (np - 1);                   // all but one pass of 6 stages, if what is left fits one pass
        if (rem >= 6 * (np - 1) + 1 && last <= 8) { sizes[0] = last; for (unsigned i = 1; i < np; i++) sizes[i] = 6; }
        else { unsigned r = rem; for (unsigned i = 0; i < np; i++) { sizes[i] = (r + (np - i) - 1) / (np - i); r -= sizes[i]; } }
    }
    unsigned cur = lg_n;
    for (unsigned i = 0; i < np; i++) {
        pl.step[pl.nsteps++] = r64_step{sizes[i] == 6 ? 1 : 0, cur, sizes[i]};
        cur -= sizes[i];
    }
    pl.step[pl.nsteps++] = r64_step{2, 12, 12};
    return pl;
}

// How a coset transform runs on this plan: 0 = the separate scaling launch (the reference's RR order, whose exponents follow
// neither index; 2^12 with natural exponents), 1 / 2 = folded (r64_table_item above; a generic pass on top of the plan takes
// its share as row constants and, for the natural exponents, g^col per work item: ntt_pass::cmode, ntt_kernels.hpp).
// |gs|: the DIF network; |foldable|: a coset transform in the NN / NR / RN order.
static inline unsigned r64_coset_mode(const r64_plan& pl, bool gs, bool inverse, bool foldable)
{
    if (!foldable || pl.nsteps == 0) return 0;
    for (unsigned i = 1; i < pl.nsteps; i++) if (pl.step[i].kind == 0) return 0;      // (a generic pass is only ever the top one)
    const unsigned mode = gs != inverse ? 1 : 2;
    if (mode == 1 && pl.nsteps < 2) return 0;
    return mode;
}

} // namespace sppark_amd
