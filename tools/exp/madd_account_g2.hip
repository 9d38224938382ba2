// Straight-line instruction account of ONE mixed addition of the BLS12-381 G2 bucket pipeline (the fast path of
// xyzz_dev<fp2x_dev>::madd, ec/xyzzx2_dev.hpp) and of its parts, for tools/isa_stats.py:
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/exp/madd_account_g2.hip -o /tmp/madd_account_g2.o
//     python tools/isa_stats.py /tmp/madd_account_g2.o --classes
// No GPU needed.  Operands are loaded from / stored to memory at a lane stride so that nothing folds away.
#define FEATURE_BLS12_381 1
#include "../../sppark_amd/csrc/msm/curve_select.hpp"
using namespace sppark_amd;
typedef fp2_d F;
typedef F::fp B;
__device__ F ldF(const u32* p) { F r; for (int j = 0; j < B::NL; j++) { r.c0.l[j] = p[j * 64]; r.c1.l[j] = p[(j + B::NL) * 64]; } return r; }
__device__ void stF(u32* p, const F& r) { for (int j = 0; j < B::NL; j++) { p[j * 64] = r.c0.l[j]; p[(j + B::NL) * 64] = r.c1.l[j]; } }
extern "C" __global__ void k_fp2_product(u32* d) { d += threadIdx.x; F a = ldF(d), b = ldF(d + 2000); stF(d, F::mul<3>(a, b)); stF(d + 2000, b); }
extern "C" __global__ void k_fp2_square(u32* d) { d += threadIdx.x; F a = ldF(d), b = ldF(d + 2000); stF(d, a.sqr<13>()); stF(d + 2000, b); }
extern "C" __global__ void k_fp2_norm(u32* d) { d += threadIdx.x; F a = ldF(d), b = ldF(d + 2000); stF(d, a.norm()); stF(d + 2000, b); }
extern "C" __global__ void k_fp2_lazy_sub_norm(u32* d) { d += threadIdx.x; F a = ldF(d), b = ldF(d + 2000); stF(d, F::sub<10>(a, b).norm()); stF(d + 2000, b); }
extern "C" __global__ void k_fp2_zero_test(u32* d) { d += threadIdx.x; F a = ldF(d), b = ldF(d + 2000); if (a.is_zero_mod<12>()) stF(d, b); }
extern "C" __global__ __launch_bounds__(256) void k_g2_mixed_addition_fast_path(u32* d, int negate)
{
    d += threadIdx.x;
    constexpr int KX = 10, KY = 6;
    F X = ldF(d), Y = ldF(d + 2000), ZZ = ldF(d + 4000), ZZZ = ldF(d + 6000), pX = ldF(d + 8000), pY = ldF(d + 10000);
    F U2 = F::mul<3>(pX, ZZ);
    F S2 = F::mul<3>(pY, ZZZ);
    if (negate) S2 = F::neg<3>(S2).norm();
    const F Pd = F::sub<KX>(U2, X).norm();
    const F Rd = F::sub<KY>(S2, Y).norm();
    const F PP  = Pd.sqr<13>();
    const F RR  = Rd.sqr<10>();
    const F PPP = F::mul<13>(Pd, PP);
    const F Q   = F::mul<KX>(X, PP);
    const F T   = PPP + Q + Q;
    const F X3  = F::sub<7, 3>(RR, T).norm();
    const F D   = F::sub<10>(Q, X3).norm();
    Y   = F::sub<3>(F::mul<13>(D, Rd), F::mul<KY>(Y, PPP)).norm();
    ZZ  = F::mul<3>(ZZ, PP);
    ZZZ = F::mul<3>(ZZZ, PPP);
    X = X3;
    stF(d, X); stF(d + 2000, Y); stF(d + 4000, ZZ); stF(d + 6000, ZZZ);
}
