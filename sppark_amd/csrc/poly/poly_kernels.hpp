// Polynomial primitives over the NTT fields: the role of the reference's polynomial/ directory
//
//   prefix_op<Add|Multiply>     polynomial/prefix_op.cuh:17-47,324-396   inclusive scan
//   evaluate                    polynomial/evaluate.cuh:307-412          ret[j] = sum_i c_i x_j^i
//   div_by_x_minus_z<rotate>    polynomial/div_by_x_minus_z.cuh:447-486  synthetic division
//
// The reference runs each as ONE cooperative kernel whose work-groups hand carries to each other
// through global memory between grid-wide syncs, with warp shuffles of whole field elements
// inside a block.  Here all three are instances of one three-launch scan over a monoid
//
//   reduce    every tile of NT*E elements -> one aggregate          (1 read of the array)
//   spine     exclusive scan of the tile aggregates, one work-group
//   apply     every tile rescans its elements seeded with its carry (1 read + 1 write)
//
// with kernel boundaries as the only grid-wide synchronisation (as everywhere in this library)
// and LDS, not cross-lane shuffles of 8-word values, for the step across the lanes of a tile.
//
//   Add       M = F,       combine(l, r) = l + r
//   Multiply  M = F,       combine(l, r) = l * r
//   Horner    M = (m, a),  the affine map x -> m*x + a; element c is (z, c):  b_i = c_i + z*b_{i-1}
//
// Synthetic division by (x - z) is the Horner scan from the TOP coefficient down:
// B_k = sum_{m >= k} c_m z^(m-k); B_0 is the remainder p(z), B_1.. the quotient
// (div_by_x_minus_z.cuh:132-157).  A lane's sequential part is plain Horner (one product per
// element); only the steps across lanes and tiles compose (m, a) pairs, and the m's of whole
// lanes / tiles are the fixed powers z^E, z^(NT*E) (k_horner_setup).
#pragma once
#include "../ff/small_fields_dev.hpp"
#include "../ff/fr256_dev.hpp"

namespace sppark_amd {

static constexpr unsigned POLY_NT = 256;                                       // lanes per tile
template<class F> struct poly_geom { static constexpr unsigned E = sizeof(F) <= 8 ? 8 : 4; };      // (16-byte bb31_4: 4)     // elements per lane

template<class F> SPPARK_DEVFN F poly_zero() { F r; memset(&r, 0, sizeof(r)); return r; }         // all three wire formats: zero bits

// ---- the three monoids ---------------------------------------------------------------------
template<class F> struct op_add {
    typedef F elem; typedef F M;
    static constexpr bool REVERSED = false;
    SPPARK_DEVFN M identity() const { return poly_zero<F>(); }
    SPPARK_DEVFN M combine(const M& l, const M& r) const { return l + r; }
    // inclusive scan of x[0..cnt) in place, seeded with the exclusive prefix |seed|; returns the total
    SPPARK_DEVFN M scan(F* x, unsigned cnt, const M& seed) const
    {   M run = seed; for (unsigned i = 0; i < cnt; i++) { run = run + x[i]; x[i] = run; } return run;   }
    SPPARK_DEVFN M lane_total(const F* x, unsigned cnt) const
    {   M run = poly_zero<F>(); for (unsigned i = 0; i < cnt; i++) run = run + x[i]; return run;   }
    SPPARK_DEVFN M tile_total(const M& t) const { return t; }
};
template<class F> struct op_mul {
    typedef F elem; typedef F M;
    static constexpr bool REVERSED = false;
    SPPARK_DEVFN M identity() const { return F::one(); }
    SPPARK_DEVFN M combine(const M& l, const M& r) const { return l * r; }
    SPPARK_DEVFN M scan(F* x, unsigned cnt, const M& seed) const
    {   M run = seed; for (unsigned i = 0; i < cnt; i++) { run = run * x[i]; x[i] = run; } return run;   }
    SPPARK_DEVFN M lane_total(const F* x, unsigned cnt) const
    {   M run = F::one(); for (unsigned i = 0; i < cnt; i++) run = run * x[i]; return run;   }
    SPPARK_DEVFN M tile_total(const M& t) const { return t; }
};
template<class F> struct horner_pair { F m, a; };
template<class F> struct op_horner {
    typedef F elem; typedef horner_pair<F> M;
    static constexpr bool REVERSED = true;          // logical element i is array element len-1-i
    const F* zp;                                    // device: { z, z^E, z^(NT*E) }  (k_horner_setup)
    SPPARK_DEVFN M identity() const { return M{F::one(), poly_zero<F>()}; }
    // l first, then r:  x -> r.m*(l.m*x + l.a) + r.a
    SPPARK_DEVFN M combine(const M& l, const M& r) const { return M{l.m * r.m, r.a + r.m * l.a}; }
    SPPARK_DEVFN M scan(F* x, unsigned cnt, const M& seed) const
    {   const F z = zp[0]; F run = seed.a; for (unsigned i = 0; i < cnt; i++) { run = x[i] + z * run; x[i] = run; } return M{zp[1], run};   }
    // (the m of a short lane / tile is never used: short ones are last in scan order)
    SPPARK_DEVFN M lane_total(const F* x, unsigned cnt) const
    {   const F z = zp[0]; F run = poly_zero<F>(); for (unsigned i = 0; i < cnt; i++) run = x[i] + z * run; return M{zp[1], run};   }
    SPPARK_DEVFN M tile_total(const M& t) const { return M{zp[2], t.a}; }
};
template<class F>
__global__ void k_horner_setup(F* zp, F z)
{
    F zl = field_pow(z, (u64)poly_geom<F>::E);
    zp[0] = z; zp[1] = zl; zp[2] = field_pow(zl, (u64)POLY_NT);
}

// physical index of logical element i
template<bool REV> SPPARK_DEVFN size_t poly_index(size_t i, size_t len) { return REV ? len - 1 - i : i; }

// Inclusive scan of one value per lane across the NT lanes of a tile through LDS
// (Hillis-Steele; lds: 2*NT values of M).
#if defined(__HIP_DEVICE_COMPILE__)
template<class Op>
__device__ __forceinline__ typename Op::M poly_lane_scan(const Op& op, typename Op::M v, typename Op::M* lds, unsigned t)
{
    typedef typename Op::M M;
    M* cur = lds; M* nxt = lds + POLY_NT;
    cur[t] = v;
    __syncthreads();
    #pragma unroll 1
    for (unsigned d = 1; d < POLY_NT; d <<= 1) {
        M mine = cur[t];
        if (t >= d) mine = op.combine(cur[t - d], mine);
        nxt[t] = mine;
        __syncthreads();
        M* s = cur; cur = nxt; nxt = s;
    }
    return cur[t];
}
#endif

// ---- kernels ------------------------------------------------------------------------------------
// reduce: agg[tile] = combination of the tile's elements
template<class Op>
__global__ __launch_bounds__(POLY_NT)
void k_poly_reduce(typename Op::M* __restrict__ agg, const typename Op::elem* __restrict__ inp, size_t len, Op op)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename Op::elem F; typedef typename Op::M M;
    constexpr unsigned E = poly_geom<F>::E;
    extern __shared__ unsigned char poly_lds[];
    M* lds = reinterpret_cast<M*>(poly_lds);
    const unsigned t = threadIdx.x;
    const size_t base = ((size_t)blockIdx.x * POLY_NT + t) * E;
    F x[E];
    unsigned cnt = 0;
    #pragma unroll
    for (unsigned k = 0; k < E; k++) if (base + k < len) { x[k] = inp[poly_index<Op::REVERSED>(base + k, len)]; cnt = k + 1; }
    M v = cnt ? op.lane_total(x, cnt) : op.identity();
    M incl = poly_lane_scan(op, v, lds, t);
    if (t == POLY_NT - 1) agg[blockIdx.x] = op.tile_total(incl);
#endif
}

// spine: agg[0..ntiles) -> exclusive prefixes, in place, one work-group of NT lanes
template<class Op>
__global__ __launch_bounds__(POLY_NT)
void k_poly_spine(typename Op::M* __restrict__ agg, size_t ntiles, Op op)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename Op::M M;
    extern __shared__ unsigned char poly_lds[];
    M* lds = reinterpret_cast<M*>(poly_lds);
    __shared__ M carry_slot;
    const unsigned t = threadIdx.x;
    const size_t per = (ntiles + POLY_NT - 1) / POLY_NT;            // consecutive tiles per lane
    const size_t lo = (size_t)t * per, hi = lo + per < ntiles ? lo + per : ntiles;
    M tot = op.identity();
    for (size_t i = lo; i < hi; i++) tot = op.combine(tot, agg[i]);
    M incl = poly_lane_scan(op, tot, lds, t);
    // exclusive prefix of this lane = inclusive of the previous one
    if (t == 0) carry_slot = op.identity();
    __syncthreads();
    lds[t] = incl;
    __syncthreads();
    M run = t ? lds[t - 1] : carry_slot;
    for (size_t i = lo; i < hi; i++) { M a = agg[i]; agg[i] = run; run = op.combine(run, a); }
#endif
}

// apply: out[i] = scan value of logical element i.  SHIFT (division with rotate): the value of logical
// element i goes to the physical slot of logical element i+1; the one value per tile that would land in
// the next tile's input range is parked in edge[tile] and written by k_poly_edges afterwards, and the
// last logical element's value (the remainder) goes to the slot of logical element 0.
template<class Op, bool SHIFT>
__global__ __launch_bounds__(POLY_NT)
void k_poly_apply(typename Op::elem* __restrict__ out, const typename Op::elem* __restrict__ inp,
                  const typename Op::M* __restrict__ carry, typename Op::elem* __restrict__ edge, size_t len, Op op)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename Op::elem F; typedef typename Op::M M;
    constexpr unsigned E = poly_geom<F>::E;
    extern __shared__ unsigned char poly_lds[];
    M* lds = reinterpret_cast<M*>(poly_lds);
    const unsigned t = threadIdx.x;
    const size_t base = ((size_t)blockIdx.x * POLY_NT + t) * E;
    F x[E];
    unsigned cnt = 0;
    #pragma unroll
    for (unsigned k = 0; k < E; k++) if (base + k < len) { x[k] = inp[poly_index<Op::REVERSED>(base + k, len)]; cnt = k + 1; }
    M v = cnt ? op.lane_total(x, cnt) : op.identity();
    M incl = poly_lane_scan(op, v, lds, t);
    __syncthreads();
    lds[t] = incl;
    __syncthreads();                                    // (also: every lane of the tile has loaded its inputs)
    M seed = carry[blockIdx.x];
    if (t) seed = op.combine(seed, lds[t - 1]);
    if (cnt) op.scan(x, cnt, seed);
    #pragma unroll
    for (unsigned k = 0; k < E; k++) {
        if (k >= cnt) break;
        const size_t i = base + k;
        if (!SHIFT) out[poly_index<Op::REVERSED>(i, len)] = x[k];
        else if (i + 1 == len || (t == POLY_NT - 1 && k == E - 1)) edge[blockIdx.x] = x[k];     // tile's last element
        else out[poly_index<Op::REVERSED>(i + 1, len)] = x[k];
    }
#endif
}

template<class F, bool REV>
__global__ __launch_bounds__(POLY_NT)
void k_poly_edges(F* __restrict__ out, const F* __restrict__ edge, size_t ntiles, size_t len)
{
    const size_t tile = (size_t)blockIdx.x * POLY_NT + threadIdx.x;
    if (tile >= ntiles) return;
    const size_t tile_sz = (size_t)POLY_NT * poly_geom<F>::E;
    const size_t last = (tile + 1) * tile_sz < len ? (tile + 1) * tile_sz - 1 : len - 1;       // the tile's last logical element
    out[poly_index<REV>(last + 1 == len ? 0 : last + 1, len)] = edge[tile];
}

// ---- evaluation --------------------------------------------------------------------------------
// part[tile*n + j] = x_j^(tile_start) * sum_{i in tile} c_i x_j^(i - tile_start)
template<class F>
__global__ __launch_bounds__(POLY_NT)
void k_poly_eval(F* __restrict__ part, const F* __restrict__ xs, unsigned n, const F* __restrict__ coeffs, size_t len)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr unsigned E = poly_geom<F>::E * 2;
    __shared__ F red[POLY_NT];
    const unsigned t = threadIdx.x;
    const size_t tile_start = (size_t)blockIdx.x * POLY_NT * E, base = tile_start + (size_t)t * E;
    F c[E];                                                             // (E = 2x the scan's: evaluation keeps no second copy)
    unsigned cnt = 0;
    #pragma unroll
    for (unsigned k = 0; k < E; k++) if (base + k < len) { c[k] = coeffs[base + k]; cnt = k + 1; }
    for (unsigned j = 0; j < n; j++) {
        const F x = xs[j];
        F h = poly_zero<F>();
        for (unsigned k = cnt; k--;) h = c[k] + x * h;                  // Horner over the lane's coefficients
        h = h * field_pow(field_pow(x, (u64)E), (u64)t);                // * x^(E*lane)
        red[t] = h;
        __syncthreads();
        for (unsigned d = POLY_NT / 2; d; d >>= 1) {
            if (t < d) red[t] = red[t] + red[t + d];
            __syncthreads();
        }
        if (t == 0) part[(size_t)blockIdx.x * n + j] = red[0] * field_pow(x, (u64)tile_start);
        __syncthreads();
    }
#endif
}
// ret[j] = sum over tiles of part[tile*n + j]; one work-group per j
template<class F>
__global__ __launch_bounds__(POLY_NT)
void k_poly_eval_sum(F* __restrict__ ret, const F* __restrict__ part, unsigned n, size_t ntiles)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ F red[POLY_NT];
    const unsigned t = threadIdx.x, j = blockIdx.x;
    F s = poly_zero<F>();
    for (size_t i = t; i < ntiles; i += POLY_NT) s = s + part[i * n + j];
    red[t] = s;
    __syncthreads();
    for (unsigned d = POLY_NT / 2; d; d >>= 1) {
        if (t < d) red[t] = red[t] + red[t + d];
        __syncthreads();
    }
    if (t == 0) ret[j] = red[0];
#endif
}

} // namespace sppark_amd
