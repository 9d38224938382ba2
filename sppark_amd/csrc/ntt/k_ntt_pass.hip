// Explicit instantiation of the NTT pass kernels for one direction of the butterfly
// network (-DSPPARK_NTT_DIF=1: GS/DIF passes, =0: CT/DIT passes); see field_select.hpp.
#include "field_select.hpp"
#ifndef SPPARK_NTT_DIF
# error "compile with -DSPPARK_NTT_DIF=0 or 1"
#endif
namespace sppark_amd {
#define SPPARK_NTT_DEFINE(DIF, INV, R1, R2) \
    template __global__ void k_ntt_pass<ntt_fr_t, DIF, INV, R1, R2>(ntt_fr_t*, ntt_tables<ntt_fr_t>, ntt_pass);
SPPARK_NTT_PASS_ALL(SPPARK_NTT_DEFINE, (SPPARK_NTT_DIF != 0))
#if SPPARK_NTT_DIF                                                 // (one of the two units carries the small-transform kernel)
#define SPPARK_NTT_SMALL_DEFINE(INV, LGC) \
    template __global__ void k_ntt_small<ntt_fr_t, INV, LGC>(ntt_fr_t*, ntt_tables<ntt_fr_t>, ntt_tables<ntt_fr_t>, unsigned);
#if defined(FEATURE_GOLDILOCKS) || defined(FEATURE_BABY_BEAR)
SPPARK_NTT_SMALL_ALL_NARROW(SPPARK_NTT_SMALL_DEFINE)
#else
SPPARK_NTT_SMALL_ALL_WIDE(SPPARK_NTT_SMALL_DEFINE)
#endif
#endif
#if defined(FEATURE_GOLDILOCKS) || defined(FEATURE_BABY_BEAR)      // wide fields stop at 4 stages per pass in registers ...
SPPARK_NTT_PASS_ALL_BIG(SPPARK_NTT_DEFINE, (SPPARK_NTT_DIF != 0))
#else                                                              // ... and run up to 8 with one stage per round
#define SPPARK_NTT_LAT_DEFINE(INV) \
    template __global__ void k_ntt_pass_lat<ntt_fr_t, (SPPARK_NTT_DIF != 0), INV>(ntt_fr_t*, ntt_tables<ntt_fr_t>, ntt_pass);
SPPARK_NTT_LAT_DEFINE(false) SPPARK_NTT_LAT_DEFINE(true)
#endif
}
