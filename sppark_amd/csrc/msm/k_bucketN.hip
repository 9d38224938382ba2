#include "curve_select.hpp"
#include "msm_kernels.hpp"
namespace sppark_amd {
template __global__ void k_bucket_levelN<fp_d>(bucket_d*, bucket_d*, const bucket_d*, const bucket_d*,
                                               unsigned, unsigned, unsigned, unsigned);
}
