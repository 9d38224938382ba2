#include "curve_select.hpp"
#include "msm_kernels.hpp"
#include "msm_piece_kernels.hpp"
namespace sppark_amd {
template __global__ void k_reduce_runs<inst_fp>(inst_m*, u32*, inst_m*, const u32*, const inst_m*,
                                             unsigned, unsigned, unsigned, int, const u32*);
template __global__ void k_join_runs<inst_fp>(inst_m*, u32*, const u32*, const inst_m*, unsigned, u32*);
template __global__ void k_reduce_tail<inst_fp>(inst_m*, u32*, inst_m*, u32*, inst_m*, unsigned, unsigned, const u32*);
template __global__ void k_piece_level<inst_fp>(inst_m*, u32*, inst_m*, const u32*, unsigned, unsigned, unsigned, unsigned,
                                             unsigned, unsigned, unsigned, u32*);
}
