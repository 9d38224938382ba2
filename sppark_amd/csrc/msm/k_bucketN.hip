#include "curve_select.hpp"
#include "msm_kernels.hpp"
namespace sppark_amd {
template __global__ void k_bucket_levelN<fp_d>(bucket_m*, bucket_m*, const bucket_m*, const bucket_m*,
                                               unsigned, unsigned, unsigned, unsigned);
}
