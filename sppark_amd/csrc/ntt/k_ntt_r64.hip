// Explicit instantiation of the radix-64 plan's kernels (ntt_r64_kernels.hpp) for one direction of
// the butterfly network (-DSPPARK_NTT_DIF=1: GS/DIF, =0: CT/DIT); single-word fields only.
#include "field_select.hpp"
#include "ntt_r64_kernels.hpp"
#ifndef SPPARK_NTT_DIF
# error "compile with -DSPPARK_NTT_DIF=0 or 1"
#endif
namespace sppark_amd {
#define SPPARK_R64_DEFINE(K, INV) \
    template __global__ void K<ntt_fr_t, (SPPARK_NTT_DIF != 0), INV>(ntt_fr_t*, ntt_r64_args<ntt_fr_t>);
SPPARK_R64_DEFINE(k_ntt6, false) SPPARK_R64_DEFINE(k_ntt6, true)
SPPARK_R64_DEFINE(k_ntt12, false) SPPARK_R64_DEFINE(k_ntt12, true)
#if SPPARK_NTT_DIF == 0
// the first step of sppark_lde's forward RN transform reading the compact coefficients (ntt_r64_args::lde_src)
template __global__ void k_ntt12<ntt_fr_t, false, false, true>(ntt_fr_t*, ntt_r64_args<ntt_fr_t>);
#endif
}
