// Host driver of the polynomial primitives (polynomial/prefix_op.cuh:324-396,
// polynomial/evaluate.cuh:307-412, polynomial/div_by_x_minus_z.cuh:447-486): three plain
// launches per scan on one stream, scratch from the per-library pool (util/runtime.hpp).
#pragma once
#include "poly_kernels.hpp"
#include "../util/runtime.hpp"

namespace sppark_amd {

template<class F> struct poly_engine {
    static constexpr size_t TILE = (size_t)POLY_NT * poly_geom<F>::E;

    typedef pooled_scratch scratch;     // util/runtime.hpp: kept between calls; freed instead when the call ends in an exception

    template<class Op, bool SHIFT>
    static void scan(F* d_out, const F* d_inp, size_t len, const Op& op, hipStream_t stream)
    {
        typedef typename Op::M M;
        if (len == 0) return;
        const size_t ntiles = (len + TILE - 1) / TILE;
        if (ntiles > 0x7fffffffu) HIP_OK(hipErrorInvalidValue);
        scratch agg(ntiles * sizeof(M)), edge(SHIFT ? ntiles * sizeof(F) : 0);
        const size_t lds = 2 * POLY_NT * sizeof(M);
        hipLaunchKernelGGL((k_poly_reduce<Op>), dim3((unsigned)ntiles), dim3(POLY_NT), lds, stream, (M*)agg.p, d_inp, len, op);
        hipLaunchKernelGGL((k_poly_spine<Op>), dim3(1), dim3(POLY_NT), lds, stream, (M*)agg.p, ntiles, op);
        hipLaunchKernelGGL((k_poly_apply<Op, SHIFT>), dim3((unsigned)ntiles), dim3(POLY_NT), lds, stream,
                           d_out, d_inp, (const M*)agg.p, (F*)edge.p, len, op);
        if (SHIFT)
            hipLaunchKernelGGL((k_poly_edges<F, Op::REVERSED>), dim3((unsigned)((ntiles + POLY_NT - 1) / POLY_NT)), dim3(POLY_NT), 0, stream,
                               d_out, (const F*)edge.p, ntiles, len);
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(stream));           // the scratch goes back to the pool with this frame
        agg.done(); edge.done();
    }

    // out[i] = inp[0] (op) ... (op) inp[i];  op 0 = Add, 1 = Multiply; d_out may alias d_inp
    static void prefix_op(F* d_out, const F* d_inp, size_t len, int which, hipStream_t stream)
    {
        if (which == 0)      scan<op_add<F>, false>(d_out, d_inp, len, op_add<F>(), stream);
        else if (which == 1) scan<op_mul<F>, false>(d_out, d_inp, len, op_mul<F>(), stream);
        else HIP_OK(hipErrorInvalidValue);
    }

    // In-place division of sum_i c_i x^i by (x - z): B_k = sum_{m >= k} c_m z^(m-k).
    // rotate == false: inout[k] = B_k (remainder first, then the quotient);
    // rotate == true : inout[k-1] = B_k for k >= 1 and inout[len-1] = B_0 (quotient first, remainder last).
    static void div_by_x_minus_z(F* d_inout, size_t len, const F& z, bool rotate, hipStream_t stream)
    {
        if (len == 0) return;
        scratch zp(3 * sizeof(F));
        hipLaunchKernelGGL(k_horner_setup<F>, dim3(1), dim3(1), 0, stream, (F*)zp.p, z);
        HIP_OK(hipGetLastError());
        op_horner<F> op; op.zp = (const F*)zp.p;
        if (rotate) scan<op_horner<F>, true>(d_inout, d_inout, len, op, stream);
        else        scan<op_horner<F>, false>(d_inout, d_inout, len, op, stream);
        zp.done();                                      // (scan() synchronised the stream)
    }

    // ret[j] = sum_i coeffs[i] * x[j]^i for j < n; all device pointers
    static void evaluate(F* d_ret, const F* d_x, size_t n, const F* d_coeffs, size_t len, hipStream_t stream)
    {
        if (n == 0) return;
        if (n > 0x7fffffffu) HIP_OK(hipErrorInvalidValue);
        const size_t tile = TILE * 2, ntiles = len ? (len + tile - 1) / tile : 1;
        scratch part(ntiles * n * sizeof(F));
        hipLaunchKernelGGL(k_poly_eval<F>, dim3((unsigned)ntiles), dim3(POLY_NT), 0, stream, (F*)part.p, d_x, (unsigned)n, d_coeffs, len);
        hipLaunchKernelGGL(k_poly_eval_sum<F>, dim3((unsigned)n), dim3(POLY_NT), 0, stream, d_ret, (const F*)part.p, (unsigned)n, ntiles);
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(stream));
        part.done();
    }
};

} // namespace sppark_amd
