// ORACLE / TEST INFRASTRUCTURE ONLY -- never linked into, loaded by or shipped with the product.
// NOT part of the default reference build (`make -C oracle ref`): `make -C oracle ref_poly` builds
// oracle/_ref/libref_poly_<field>.so, and tools/gpu_poly_vs_reference.py runs it, ONE call per process under a timeout.
// Status (end of round 4): built, first GPU run did not return (a cooperative-grid launch of the reference presumably
// never completes on this part); which call blocks is what the tool is there to find out.  No test depends on it.
//
// The reference's own polynomial primitives (polynomial/prefix_op.cuh, div_by_x_minus_z.cuh: C++ templates with no
// exported symbol) instantiated over the library's field and built for gfx950 by the reference's HIP path.  Forwarders
// only: device buffers in, the reference's kernels on the reference's stream, synchronous.
// What does NOT build with this image's hipcc, and is therefore not here (no stand-ins are written for it):
//   * polynomial/evaluate.cuh -- `__launch_bounds__(d_evaluate_bsize<N, fr_t>())` (:23): HIP's __launch_bounds__ is a
//     macro and the comma of the template argument list splits its argument;
//   * polynomial/div_by_x_minus_z.cuh for the 256-bit fields -- `__launch_bounds__(BSZ)` with BSZ = 0 (:18,453):
//     amdgpu_flat_work_group_size rejects a zero maximum.  The single-word fields (BSZ = 1024) build.
#if defined(FEATURE_BLS12_381)
# include <ff/bls12-381.hpp>
#elif defined(FEATURE_BN254)
# include <ff/alt_bn128.hpp>
#elif defined(FEATURE_GOLDILOCKS)
# include <ff/goldilocks.hpp>
#elif defined(FEATURE_BABY_BEAR)
# include <ff/baby_bear.hpp>
#else
# error "no FEATURE"
#endif
#include <util/gpu_t.cuh>
#include <polynomial/prefix_op.cuh>
#if defined(FEATURE_GOLDILOCKS) || defined(FEATURE_BABY_BEAR)
# include <polynomial/div_by_x_minus_z.cuh>
# define REF_HAS_DIV 1
#endif

#define REF_FFI extern "C" __attribute__((visibility("default")))

// polynomial/prefix_op.cuh:324-396; op: 0 = Add, 1 = Multiply; out may equal inp
REF_FFI int ref_prefix_op(fr_t* d_out, const fr_t* d_inp, size_t len, int op)
{
    try {
        auto& gpu = select_gpu(0);
        stream_t& s = gpu;
        if (op == 0) prefix_op<Add<fr_t>>(d_out, d_inp, len, s);
        else         prefix_op<Multiply<fr_t>>(d_out, d_inp, len, s);
        s.sync();
        return 0;
    } catch (const cuda_error& e) { return e.code() ? e.code() : -1; }
}
// polynomial/div_by_x_minus_z.cuh:446-486; z: ONE field element in host memory
REF_FFI int ref_div_by_x_minus_z(fr_t* d_inout, size_t len, const fr_t* z, int rotate)
{
#ifndef REF_HAS_DIV
    (void)d_inout; (void)len; (void)z; (void)rotate;
    return -38;                                                 // ENOSYS: not buildable for this field (see above)
#else
    try {
        auto& gpu = select_gpu(0);
        stream_t& s = gpu;
        if (rotate) div_by_x_minus_z<true>(d_inout, len, *z, s);
        else        div_by_x_minus_z<false>(d_inout, len, *z, s);
        s.sync();
        return 0;
    } catch (const cuda_error& e) { return e.code() ? e.code() : -1; }
#endif
}
