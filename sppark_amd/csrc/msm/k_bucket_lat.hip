// The bucket-sum levels for SMALL grids (at most one resident round of waves): one wave per SIMD, no register cap,
// products in interleaved pairs -- the chain of dependent additions is the time there (msm_kernels.hpp).
// G1 bucket fields with their own records only (the driver never asks for them over Fp2).
#include "curve_select.hpp"
#include "msm_kernels.hpp"
namespace sppark_amd {
template __global__ void k_bucket_level1_lat<msm_fp_d>(bucket_m*, bucket_m*, const bucket_m*, unsigned, unsigned, unsigned, const u32*);
template __global__ void k_bucket_levelN_lat<msm_fp_d>(bucket_m*, bucket_m*, const bucket_m*, const bucket_m*,
                                                       unsigned, unsigned, unsigned, unsigned);
}
