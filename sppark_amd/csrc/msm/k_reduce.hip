#include "curve_select.hpp"
#include "msm_kernels.hpp"
namespace sppark_amd {
template __global__ void k_reduce_runs<fp_d>(bucket_m*, u32*, bucket_m*, const u32*, const bucket_m*,
                                             unsigned, unsigned, unsigned, int);
}
