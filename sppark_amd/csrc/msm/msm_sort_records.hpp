// The level-A record of the MSM's two-level sort (msm_sort_kernels.hpp), in a header of its own so that the host emulation
// of the pipeline (tests/emu/emu_msm.cpp) packs and unpacks with the product's code.
#pragma once
#include "../ff/mont_dev.hpp"

namespace sppark_amd {

// ---- the level-A record ----------------------------------------------------------------------------------------------
// WIDE (8 bytes): { point index | sign << 31, k_lo }.
// PACKED (4 bytes, msm_plan::IB != 0): sign << 31 | (index mod 2^IB) << LBL | k_lo with IB = 31 - LBL.  The index bits above
// IB are not stored: level A writes the records of a partition in SLAB order (the cursor of slab s starts at the
// slab-exclusive prefix H[w][s][partition]), the slabs are 2^(IB - SH) points, so the records of index group g = index >> IB
// -- the slabs g 2^SH ... (g + 1) 2^SH - 1 -- are exactly the positions [H[w][g 2^SH][p], H[w][(g + 1) 2^SH][p]) of the
// partition: level B finds g from the POSITION of a record, by a binary search over the <= 128 group boundaries it keeps in LDS
// (recA<true>::entry).  Halves what level A writes and level B reads.
struct partA_fmt {
    const u32* H;                       // the slab-exclusive prefixes (level A's cursor bases)
    unsigned IB, SH, NGP, nslabs;       // NGP = index groups rounded up to a power of two (1: the index fits the record)
};
static constexpr unsigned PARTA_MAX_GROUPS = 128;
template<bool PK> struct recA;
template<> struct recA<false> {
    typedef uint2 type;
    SPPARK_DEVFN static type make(u32 idx_sign, u32 k, unsigned, unsigned) { return make_uint2(idx_sign, k); }
    SPPARK_DEVFN static u32 key(const type& r, u32) { return r.y; }
    SPPARK_DEVFN static u32 entry(const type& r, unsigned, unsigned, const partA_fmt&, const u32*) { return r.x; }
};
template<> struct recA<true> {
    typedef u32 type;
    SPPARK_DEVFN static type make(u32 idx_sign, u32 k, unsigned LBL, unsigned IB)
    {   return (idx_sign & 0x80000000u) | ((idx_sign & ((1u << IB) - 1)) << LBL) | k;   }
    SPPARK_DEVFN static u32 key(const type& r, u32 kmask) { return r & kmask; }
    // |q|: position of the record in its partition; |bnd|: bnd[0] = 0, bnd[g] = first position of index group g (0xffffffff
    // beyond the last group)
    SPPARK_DEVFN static u32 entry(const type& r, unsigned q, unsigned LBL, const partA_fmt& f, const u32* bnd)
    {
        unsigned g = 0;
        for (unsigned s = f.NGP >> 1; s; s >>= 1) if (bnd[g + s] <= q) g += s;
        return (r & 0x80000000u) | (((r & 0x7fffffffu) >> LBL) + (g << f.IB));
    }
};
// the group boundaries of partition (w, khi) into LDS (the caller's barrier follows)
SPPARK_DEVFN void partA_bounds(u32* bnd, const partA_fmt& f, unsigned w, unsigned khi, unsigned NA, unsigned tid)
{
    if (tid < f.NGP) {
        const unsigned slab = tid << f.SH;
        bnd[tid] = tid == 0 ? 0u : slab < f.nslabs ? f.H[((size_t)w * f.nslabs + slab) * NA + khi] : 0xffffffffu;
    }
}

} // namespace sppark_amd
