// Straight-line instruction account of ONE mixed addition of the alt_bn128 bucket pipeline -- NINE 29-bit limbs, the TIGHT
// formulas of ec/xyzzx_dev.hpp (X kept normalised) -- and of its parts, for tools/isa_stats.py (no GPU needed):
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/exp/madd_account_bn254.hip -o /tmp/madd_account_bn254.o
//     python tools/isa_stats.py /tmp/madd_account_bn254.o --classes
// As madd_account.hip (the 14-limb account): operands from / to memory at a lane stride so that nothing folds away.
#define FEATURE_BN254 1
#include "../../sppark_amd/csrc/msm/curve_select.hpp"
using namespace sppark_amd;
typedef msm_fp_d F;
static_assert(F::TIGHT && F::NL == 9, "the nine-limb class");
__device__ F ldF(const u32* p) { F r; for (int j = 0; j < F::NL; j++) r.l[j] = p[j * 64]; return r; }
__device__ void stF(u32* p, const F& r) { for (int j = 0; j < F::NL; j++) p[j * 64] = r.l[j]; }
extern "C" __global__ void k_ldst(u32* d) { d += threadIdx.x; F a = ldF(d), b = ldF(d + 1000); stF(d, a); stF(d + 1000, b); }
extern "C" __global__ void k_product(u32* d) { d += threadIdx.x; F a = ldF(d), b = ldF(d + 1000); stF(d, a * b); stF(d + 1000, b); }
extern "C" __global__ void k_product_pair_fat(u32* d) { d += threadIdx.x; F a = ldF(d), b = ldF(d + 1000), c = ldF(d + 2000), e = ldF(d + 3000), r0, r1; F::mul2<true, false>(r0, r1, a, b, c, e); stF(d, r0); stF(d + 1000, r1); }
extern "C" __global__ void k_product_pair_norm(u32* d) { d += threadIdx.x; F a = ldF(d), b = ldF(d + 1000), c = ldF(d + 2000), e = ldF(d + 3000), r0, r1; F::mul2<true, true>(r0, r1, a, b, c, e); stF(d, r0); stF(d + 1000, r1); }
extern "C" __global__ void k_square_pair(u32* d) { d += threadIdx.x; F a = ldF(d), b = ldF(d + 1000), r0, r1; F::sqr2(r0, r1, a, b); stF(d, r0); stF(d + 1000, r1); }
extern "C" __global__ void k_sum_of_two_products(u32* d) { d += threadIdx.x; F a = ldF(d), b = ldF(d + 1000), c = ldF(d + 2000), e = ldF(d + 3000); stF(d, F::mul_add(a, b, c, e)); stF(d + 1000, b); }
extern "C" __global__ void k_norm(u32* d) { d += threadIdx.x; F a = ldF(d), b = ldF(d + 1000); stF(d, a.norm()); stF(d + 1000, b); }
extern "C" __global__ void k_lazy_sub(u32* d) { d += threadIdx.x; F a = ldF(d), b = ldF(d + 1000); stF(d, F::sub<11, 1>(a, b)); stF(d + 1000, b); }
extern "C" __global__ void k_zero_test(u32* d) { d += threadIdx.x; F a = ldF(d), b = ldF(d + 1000); if (a.is_zero_mod<13>()) stF(d, b); }
extern "C" __global__ void k_mixed_addition_fast_path(u32* d, int negate)
{
    d += threadIdx.x;
    F X = ldF(d), Y = ldF(d + 1000), ZZ = ldF(d + 2000), ZZZ = ldF(d + 3000), pX = ldF(d + 4000), pY = ldF(d + 5000);
    F U2, S2;
    F::mul2<true, true>(U2, S2, pX, ZZ, pY, ZZZ);
    if (negate) S2 = F::template neg<3>(S2);
    F Pd = F::template sub<11, 1>(U2, X).norm();
    F Rd = F::template sub<4, 2>(S2, Y).norm();
    F PP, RR, PPP, Q;
    F::sqr2(PP, RR, Pd, Rd);
    F::mul2<true, false>(PPP, Q, Pd, PP, X, PP);
    F T   = PPP + Q + Q;
    F X3  = F::template sub<8, 3>(RR, T).norm();
    F D   = F::template sub<11, 1>(Q, X3);
    F nY  = F::template neg<4, 2>(Y);
    Y   = F::mul_add(D, Rd, nY, PPP);
    F::mul2<true, true>(ZZ, ZZZ, ZZ, PP, ZZZ, PPP);
    X = X3;
    stF(d, X); stF(d + 1000, Y); stF(d + 2000, ZZ); stF(d + 3000, ZZZ);
}
