#include "curve_select.hpp"
#include "msm_kernels.hpp"
namespace sppark_amd {
template __global__ void k_reduce_runs<fp_d>(bucket_d*, u32*, bucket_d*, const u32*, const bucket_d*,
                                             unsigned, unsigned, unsigned, int);
}
