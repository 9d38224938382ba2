// Explicit instantiation of the hot MSM kernel (own translation unit so that
// the library builds in parallel; see msm_kernels.hpp for the kernel itself).
#include "curve_select.hpp"
#include "msm_kernels.hpp"
#if defined(SPPARK_G2) && !defined(SPPARK_FP2_32LIMB)
#include "msm_g2c_kernels.hpp"
#endif
namespace sppark_amd {
template __global__ void k_accumulate<inst_fp, false>(inst_m*, u32*, inst_m*, const unsigned char*, unsigned,
                                                   const u32*, const u32*, unsigned, unsigned, unsigned, unsigned, unsigned);
template __global__ void k_accumulate<inst_fp, true>(inst_m*, u32*, inst_m*, const unsigned char*, unsigned,
                                                  const u32*, const u32*, unsigned, unsigned, unsigned, unsigned, unsigned);
#if defined(SPPARK_G2) && !defined(SPPARK_FP2_32LIMB)     // G2: the accumulation with one Fp2 component per wave
template __global__ void k_accumulate_g2c<inst_fp>(inst_m*, u32*, inst_m*, const unsigned char*, unsigned,
                                                          const u32*, const u32*, unsigned, unsigned, unsigned, unsigned, unsigned);
#endif
#ifndef SPPARK_G2
template __global__ void k_bitmap_accumulate<inst_fp, false>(u32*, inst_m*, const unsigned char*, unsigned, unsigned, const u32*, const u32*, unsigned);
template __global__ void k_bitmap_accumulate<inst_fp, true>(u32*, inst_m*, const unsigned char*, unsigned, unsigned, const u32*, const u32*, unsigned);
#endif
}
