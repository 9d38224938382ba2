#include "curve_select.hpp"
#include "msm_kernels.hpp"
namespace sppark_amd {
template __global__ void k_bucket_level1<inst_fp>(inst_m*, inst_m*, const inst_m*, unsigned, unsigned, unsigned, const u32*);
}
