// Curve selection by -DFEATURE_* (same macro names as the reference's build
// drivers: poc/msm-cuda/build.rs, ff/bls12-381.hpp:154-156, ff/alt_bn128.hpp:147-149).
#pragma once
#include "../ff/params.hpp"
#include "../ff/mont_dev.hpp"
#include "../ff/mont30_dev.hpp"
#include "../ff/fp2_dev.hpp"
#include "../ec/xyzz_dev.hpp"

namespace sppark_amd {
#if defined(FEATURE_BLS12_381)
typedef bls12_381_g1_p curve_p;
#elif defined(FEATURE_BN254)
typedef alt_bn128_g1_p curve_p;
#else
# error "no FEATURE"
#endif
// base-field register class of the bucket kernels: 32-bit limbs (mont_dev.hpp).
// -DSPPARK_FP30LIMB selects the reduced-radix class (mont30_dev.hpp), kept for A/B
// measurements: on MI355X it multiplies 5 % faster and squares 34 % faster but its
// carry-free additions cost more, and the mixed addition comes out 2 % SLOWER
// (profiles/r01_mont30_vs_mont32.log), so it is not the default.
#ifdef SPPARK_FP30LIMB
template<class P> using fp_class = mont30_dev<P>;
#else
template<class P> using fp_class = mont_dev<P>;
#endif
typedef fp_class<curve_p::fp> fp_d;
typedef mont_dev<curve_p::fr> fr_d;
typedef xyzz_dev<fp_d> bucket_d;           // register type
typedef bucket_d::mem_t bucket_m;           // memory image (wire format)
// G2: same pipeline over the quadratic extension (ff/fp2_dev.hpp).  The kernel
// translation units are compiled a second time with -DSPPARK_G2 to instantiate it.
typedef fp2_dev<curve_p::fp> fp2_d;
typedef xyzz_dev<fp2_d> bucket2_d;
typedef bucket2_d::mem_t bucket2_m;
#ifdef SPPARK_G2
typedef fp2_d inst_fp;
#else
typedef fp_d inst_fp;
#endif
typedef xyzz_dev<inst_fp>::mem_t inst_m;    // what the k_*.hip units instantiate
}
