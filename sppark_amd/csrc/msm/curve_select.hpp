// Curve selection by -DFEATURE_* (same macro names as the reference's build
// drivers: poc/msm-cuda/build.rs, ff/bls12-381.hpp:154-156, ff/alt_bn128.hpp:147-149).
#pragma once
#include "../ff/params.hpp"
#include "../ff/mont_dev.hpp"
#include "../ec/xyzz_dev.hpp"

namespace sppark_amd {
#if defined(FEATURE_BLS12_381)
typedef bls12_381_g1_p curve_p;
#elif defined(FEATURE_BN254)
typedef alt_bn128_g1_p curve_p;
#else
# error "no FEATURE"
#endif
typedef mont_dev<curve_p::fp> fp_d;
typedef mont_dev<curve_p::fr> fr_d;
typedef xyzz_dev<fp_d> bucket_d;
}
