// HOST TEST HARNESS (tests only): the plan of an MSM (sppark_amd/csrc/msm/msm_plan.hpp) for the host, so that its
// invariants -- what the kernels' launch shapes and LDS sizes rely on -- can be checked over every size in the GPU-less
// container (tests/test_plan.py).
#include "../../sppark_amd/csrc/msm/msm_plan.hpp"
using namespace sppark_amd;

static void put(const msm_plan& p, unsigned out[21])
{
    const unsigned v[21] = {p.n, p.wbits, p.nwins, p.NB, p.nbits, p.HB, p.LB, p.NA, p.L, p.chunks_per_win, p.nslabs, p.slab_sz,
                            p.F, p.K, p.K1, p.G, p.wpg, p.big, p.IB, p.SH, p.NG};
    for (int i = 0; i < 21; i++) out[i] = v[i];
}
// out: {n, wbits, nwins, NB, nbits, HB, LB, NA, L, chunks_per_win, nslabs, slab_sz, F, K, K1, G, wpg, big, IB, SH, NG}
extern "C" void emu_make_plan(size_t npoints, unsigned scalar_bits, unsigned wbits, unsigned L, unsigned F, unsigned K,
                              unsigned nslabs, unsigned LB, unsigned groups, unsigned K1, unsigned out[21])
{
    msm_tunables t;
    t.wbits = wbits; t.L = L; t.F = F; t.K = K; t.nslabs = nslabs; t.LB = LB; t.groups = groups; t.K1 = K1;
    put(make_plan(npoints, scalar_bits, t), out);
}
// the automatic plan as the driver asks for it: with the device's resident k_accumulate lanes known
extern "C" void emu_make_plan_resident(size_t npoints, unsigned scalar_bits, size_t resident_lanes, unsigned out[21])
{
    msm_tunables t;
    t.resident_lanes = resident_lanes;
    put(make_plan(npoints, scalar_bits, t), out);
}
extern "C" void emu_make_fixed_plan(size_t npoints, unsigned fb_wbits, unsigned fb_nwins, unsigned register_stage, unsigned out[21])
{
    msm_tunables t;
    put(make_fixed_plan(npoints, fb_wbits, fb_nwins, register_stage, t), out);
}
