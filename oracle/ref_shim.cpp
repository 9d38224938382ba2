// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Compiles the REFERENCE's own CPU Pippenger where it lies under
// /root/reference (nothing is copied into this repo):
//     msm/pippenger.hpp, ec/xyzz_t.hpp, ec/jacobian_t.hpp, ec/affine_t.hpp,
//     util/thread_pool_t.hpp, util/slice_t.hpp
// These are header-only templates parameterised on the field type
// (msm/pippenger.hpp:216-221).  The reference's own host field comes from blst
// (ff/bls12-381.hpp:90-91), which is not vendored; the template argument used
// here is the oracle's mont_t (oracle/ff.hpp), whose arithmetic is pinned
// against Python big-ints.  Output -> oracle/_ref/libref_msm.so, used in the
// build container to validate oracle/msm.hpp and to generate tests/golden/.
#include <cstdint>
#include <cstring>
#include <atomic>
#include "ff.hpp"

#include <ec/jacobian_t.hpp>
#include <ec/xyzz_t.hpp>
#include <msm/pippenger.hpp>

namespace {

template<class FP, class FR>
int ref_msm(unsigned char* out_affine, const unsigned char* points, size_t stride,
            size_t npoints, const unsigned char* scalars, int mont, unsigned nthreads)
{
    typedef jacobian_t<FP> point_t;
    typedef xyzz_t<FP> bucket_t;
    typedef typename bucket_t::affine_t affine_t;
    const size_t fb = sizeof(FP);

    std::vector<affine_t> pts;
    pts.reserve(npoints);
    for (size_t i = 0; i < npoints; i++) {
        FP x, y;
        memcpy((void*)&x, points + i * stride, fb);
        memcpy((void*)&y, points + i * stride + fb, fb);
        if (stride > 2 * fb && (points[i * stride + 2 * fb] & 1)) { x.zero(); y.zero(); }
        pts.push_back(affine_t{x, y});
    }
    std::vector<FR> sc(npoints);
    memcpy((void*)sc.data(), scalars, npoints * sizeof(FR));

    point_t ret;
    if (nthreads >= 2) {
        thread_pool_t pool(nthreads);
        mult_pippenger<bucket_t>(ret, pts.data(), npoints, sc.data(), mont != 0, &pool);
    } else {
        mult_pippenger<bucket_t>(ret, pts.data(), npoints, sc.data(), mont != 0, nullptr);
    }
    affine_t a = ret;                       // jacobian_t::operator affine_t
    static_assert(sizeof(affine_t) == 2 * sizeof(FP), "affine layout");
    if (ret.is_inf()) memset(out_affine, 0, 2 * fb);
    else              memcpy(out_affine, &a, 2 * fb);
    return 0;
}

} // namespace

extern "C"
int ref_mult_pippenger(int curve, unsigned char* out_affine, const unsigned char* points,
                       size_t stride, size_t npoints, const unsigned char* scalars,
                       int mont, unsigned nthreads)
{
    using namespace oracle;
    switch (curve) {
        case 0: return ref_msm<bls12_381_fp, bls12_381_fr>(out_affine, points, stride, npoints, scalars, mont, nthreads);
        case 1: return ref_msm<alt_bn128_fp, alt_bn128_fr>(out_affine, points, stride, npoints, scalars, mont, nthreads);
        // G2: the same reference templates over the oracle's Fp2 (jacobian_t<fp2_t>, xyzz_t<fp2_t>,
        // as poc/msm-cuda/cuda/pippenger_inf.cu:36-47 instantiates them on the device)
        case 2: return ref_msm<bls12_381_fp2, bls12_381_fr>(out_affine, points, stride, npoints, scalars, mont, nthreads);
        case 3: return ref_msm<alt_bn128_fp2, alt_bn128_fr>(out_affine, points, stride, npoints, scalars, mont, nthreads);
        // BLS12-377 (poc/msm-cuda/cuda/pippenger_inf.cu:9-10), G1 and G2
        case 4: return ref_msm<bls12_377_fp, bls12_377_fr>(out_affine, points, stride, npoints, scalars, mont, nthreads);
        case 5: return ref_msm<bls12_377_fp2, bls12_377_fr>(out_affine, points, stride, npoints, scalars, mont, nthreads);
        // the Pasta cycle (ff/pasta.hpp:93-104)
        case 6: return ref_msm<pasta_p, pasta_q>(out_affine, points, stride, npoints, scalars, mont, nthreads);
        case 7: return ref_msm<pasta_q, pasta_p>(out_affine, points, stride, npoints, scalars, mont, nthreads);
    }
    return -1;
}
