// Curve selection by -DFEATURE_* (same macro names as the reference's build
// drivers: poc/msm-cuda/build.rs, ff/bls12-381.hpp:154-156, ff/alt_bn128.hpp:147-149).
#pragma once
#include "../ff/params.hpp"
#include "../ff/mont_dev.hpp"
#include "../ff/fp2_dev.hpp"
#include "../ec/xyzz_dev.hpp"
#include "../ec/xyzzx_dev.hpp"
#include "../ec/xyzzx2_dev.hpp"

namespace sppark_amd {
#if defined(FEATURE_BLS12_381)
typedef bls12_381_g1_p curve_p;
#elif defined(FEATURE_BN254)
typedef alt_bn128_g1_p curve_p;
#elif defined(FEATURE_BLS12_377)         // poc/msm-cuda/cuda/pippenger_inf.cu:9-10, ff/bls12-377.hpp
typedef bls12_377_g1_p curve_p;
#elif defined(FEATURE_PALLAS)            // ff/pasta.hpp:93-98 (fp_t = pallas_t, fr_t = vesta_t); no pairing, no G2
typedef pallas_g1_p curve_p;
# define SPPARK_NO_G2 1
#elif defined(FEATURE_VESTA)             // ff/pasta.hpp:99-104
typedef vesta_g1_p curve_p;
# define SPPARK_NO_G2 1
#else
# error "no FEATURE"
#endif
template<class P> using fp_class = mont_dev<P>;
typedef fp_class<curve_p::fp> fp_d;          // wire-format field: generators, test hooks, conversions
typedef mont_dev<curve_p::fr> fr_d;
// coordinate field of the G1 bucket pipeline.  BLS12-381: the loosely-reduced 28-bit-limb
// class (ff/montx_dev.hpp): its mixed addition runs 1.30x faster than the 32-bit-limb one on
// MI355X (profiles/r01_montx_vs_mont32.log).  The 254/255-bit fields (alt_bn128, Pasta) take ten
// 28-bit limbs against eight 32-bit ones and still win: alt_bn128's accumulation 71.8 -> 66.7 ms
// for 2^26 points (tools/gpu_r2_job11.sh), because what the 32-bit form spends on carries exceeds
// the 56 % more multiply-adds.
// Round 5: the 254 / 255-bit base fields (alt_bn128, Pasta) on NINE 29-bit limbs: 81 instead of 100 multiply-adds per
// product and per reduction (ff/montx_dev.hpp: six or seven bits of head-room, rho = 169 / 128; the point formulas keep X
// normalised, ec/xyzzx_dev.hpp TIGHT).  alt_bn128 G1 2^26: k_accumulate 64.4 -> 57.4 ms on one box
// (profiles/r05_bn254_lb29_ab.log).  -DSPPARK_BN254_LB=28 builds the former alt_bn128 library for an A/B.
#if defined(FEATURE_BN254) && !defined(SPPARK_BN254_LB)
# define SPPARK_BN254_LB 29
#endif
#if !defined(SPPARK_FP32LIMB)    // 381 / 377 bits: 14 limbs of 28; 254 / 255 bits: 9 limbs of 29
# if defined(FEATURE_BN254)
typedef montx_dev<curve_p::fp, SPPARK_BN254_LB> msm_fp_d;
# elif defined(FEATURE_PALLAS) || defined(FEATURE_VESTA)
typedef montx_dev<curve_p::fp, 29> msm_fp_d;
# else
typedef montx_dev<curve_p::fp, 28> msm_fp_d;
# endif
#else
typedef fp_d msm_fp_d;
#endif
typedef xyzz_dev<fp_d> wire_bucket_d;       // XYZZ in the reference's wire form (ec/xyzz_t.hpp:17)
typedef wire_bucket_d::mem_t wire_bucket_m;
typedef xyzz_dev<msm_fp_d> bucket_d;       // register type
typedef bucket_d::mem_t bucket_m;           // memory image between the kernels
// G2: same pipeline over the quadratic extension.  The kernel translation units are compiled a second time with
// -DSPPARK_G2 to instantiate it.  Round 3: over the loosely-reduced 28-bit-limb base field (ff/fp2x_dev.hpp,
// ec/xyzzx2_dev.hpp) like G1; fp2_dev<> over the canonical 32-bit-limb class remains the wire-format type
// (-DSPPARK_FP2_32LIMB: the old pipeline type, for A/B).
typedef fp2_dev<curve_p::fp> fp2_wire_d;
#if !defined(SPPARK_FP2_32LIMB)
typedef fp2x_dev<curve_p::fp, 28> fp2_d;
#else
typedef fp2_wire_d fp2_d;
#endif
typedef xyzz_dev<fp2_d> bucket2_d;
typedef bucket2_d::mem_t bucket2_m;
#ifdef SPPARK_G2
typedef fp2_d inst_fp;
#else
typedef msm_fp_d inst_fp;
#endif
typedef xyzz_dev<inst_fp>::mem_t inst_m;    // what the k_*.hip units instantiate
}
