// NTT element type by -DFEATURE_* (poc/ntt-cuda/cuda/ntt_api.cu:7-21 selects the field header
// the same way).  The curve features pick the curve's SCALAR field (256-bit, Montgomery).
#pragma once
#include "../ff/params.hpp"
#include "../ff/small_fields_dev.hpp"
#include "../ff/fr256_dev.hpp"
#include "ntt_kernels.hpp"

namespace sppark_amd {
#if defined(FEATURE_GOLDILOCKS)
typedef gl64_dev ntt_fr_t;
#elif defined(FEATURE_BABY_BEAR)
typedef bb31_dev ntt_fr_t;
#elif defined(FEATURE_BLS12_381)       // ntt_api.cu:7-8 -> ff/bls12-381.hpp fr_t
typedef fr256_dev<bls12_381_fr_p> ntt_fr_t;
#elif defined(FEATURE_BN254)           // ntt_api.cu:15-16 -> ff/alt_bn128.hpp fr_t
typedef fr256_dev<alt_bn128_fr_p> ntt_fr_t;
#elif defined(FEATURE_BLS12_377)       // ntt_api.cu:7-8 -> ff/bls12-377.hpp fr_t (2-adicity 47)
typedef fr256_dev<bls12_377_fr_p> ntt_fr_t;
#elif defined(FEATURE_PALLAS)          // ntt_api.cu:9-10; "Fr for Pallas curve is Vesta" (ntt/parameters.cuh:54-55)
typedef fr256_dev<pasta_q_p> ntt_fr_t;
#elif defined(FEATURE_VESTA)           // ntt_api.cu:11-12; "Fr for Vesta curve is Pallas" (ntt/parameters.cuh:56-57)
typedef fr256_dev<pasta_p_p> ntt_fr_t;
#elif defined(FEATURE_MERSENNE31)      // ff/mersenne31.hpp: a field type only (no NTT parameters): polynomial primitives
typedef mrs31_dev ntt_fr_t;
#elif defined(FEATURE_BABY_BEAR_X4)    // ff/baby_bear.hpp:70-446 bb31_4_t: likewise
typedef bb31_4_dev ntt_fr_t;
#else
# error "no FEATURE"
#endif

// The pass kernels are instantiated in their own translation units (ntt/k_ntt_pass.hip,
// compiled once with -DSPPARK_NTT_DIF=1 and once with =0) so that the library builds in
// parallel: for the 256-bit fields one unit with all of them takes minutes.  The 256-bit fields
// use at most 4 stages per pass (radix-4 x radix-4 in registers), the single-word fields up to 8.
#define SPPARK_NTT_PASS_SET(X, DIF, INV) X(DIF, INV, 1, 0) X(DIF, INV, 1, 1) X(DIF, INV, 2, 1) X(DIF, INV, 2, 2)
#define SPPARK_NTT_PASS_SET_BIG(X, DIF, INV) X(DIF, INV, 3, 2) X(DIF, INV, 3, 3) X(DIF, INV, 4, 3) X(DIF, INV, 4, 4)
#define SPPARK_NTT_PASS_ALL(X, DIF)                                                               \
    SPPARK_NTT_PASS_SET(X, DIF, false) SPPARK_NTT_PASS_SET(X, DIF, true)
#define SPPARK_NTT_PASS_ALL_BIG(X, DIF)                                                           \
    SPPARK_NTT_PASS_SET_BIG(X, DIF, false) SPPARK_NTT_PASS_SET_BIG(X, DIF, true)
}
