#include "curve_select.hpp"
#include "msm_kernels.hpp"
namespace sppark_amd {
template __global__ void k_bucket_levelN<inst_fp>(inst_m*, inst_m*, const inst_m*, const inst_m*,
                                               unsigned, unsigned, unsigned, unsigned);
template __global__ void k_bucket_top_bits<inst_fp>(inst_m*, const inst_m*, const inst_m*, unsigned, unsigned, unsigned);
template __global__ void k_bucket_top_sum<inst_fp>(inst_m*, const inst_m*, unsigned);
}
