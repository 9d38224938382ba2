// The bucket-sum levels for SMALL grids (at most one resident round of waves): one wave per SIMD, no register cap,
// products in interleaved pairs -- the chain of dependent additions is the time there (msm_kernels.hpp).
// G1 bucket fields with their own records only (the driver never asks for them over Fp2).
#include "curve_select.hpp"
#include "msm_kernels.hpp"
#include "msm_coop_kernels.hpp"
namespace sppark_amd {
template __global__ void k_bucket_level1_lat<msm_fp_d>(bucket_m*, bucket_m*, const bucket_m*, unsigned, unsigned, unsigned, const u32*);
template __global__ void k_bucket_levelN_lat<msm_fp_d>(bucket_m*, bucket_m*, const bucket_m*, const bucket_m*,
                                                       unsigned, unsigned, unsigned, unsigned);
// ... and the subset-sum top in cooperative form (msm_coop_kernels.hpp)
template __global__ void k_bucket_top_bits_coop<msm_fp_d>(bucket_m*, const bucket_m*, const bucket_m*, unsigned, unsigned, unsigned, unsigned, unsigned);
template __global__ void k_bucket_top_sum_coop<msm_fp_d>(bucket_m*, const bucket_m*, unsigned, xyzz_mem<msm_fp_d::NW>*, u32*, u32*);
template __global__ void k_reduce_runs_coop<msm_fp_d>(bucket_m*, u32*, bucket_m*, const u32*, const bucket_m*,
                                                      unsigned, unsigned, unsigned, int, const u32*);
template __global__ void k_reduce_tail_coop<msm_fp_d>(bucket_m*, u32*, bucket_m*, u32*, bucket_m*, unsigned, unsigned, const u32*);
template __global__ void k_bucket_level1_coop<msm_fp_d>(bucket_m*, bucket_m*, const bucket_m*, unsigned, unsigned, unsigned, const u32*);
template __global__ void k_bucket_levelN_coop<msm_fp_d>(bucket_m*, bucket_m*, const bucket_m*, const bucket_m*,
                                                        unsigned, unsigned, unsigned, unsigned);
template __global__ void k_bucket_levelN_pipe<msm_fp_d>(bucket_m*, bucket_m*, const bucket_m*, const bucket_m*, unsigned, unsigned, unsigned, unsigned);
template __global__ void k_bucket_level1_pipe<msm_fp_d>(bucket_m*, bucket_m*, const bucket_m*, unsigned, unsigned, unsigned, const u32*);
template __global__ void k_bucket_small_bits_coop<msm_fp_d>(bucket_m*, const bucket_m*, const u32*, unsigned, unsigned);
template __global__ void k_piece_level_coop<msm_fp_d>(bucket_m*, u32*, bucket_m*, const u32*, unsigned, unsigned, unsigned, unsigned,
                                                      unsigned, unsigned, unsigned, u32*);
template __global__ void k_piece_tail_coop<msm_fp_d>(bucket_m*, u32*, bucket_m*, const u32*, unsigned, unsigned, unsigned, unsigned,
                                                     unsigned, unsigned, unsigned, u32*);
}
