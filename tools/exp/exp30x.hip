// Experiment harness for ff/montx_dev.hpp (not part of the product libraries; results: profiles/r01_montx_vs_mont32.log).
#include "ff/params.hpp"
#include "ff/mont_dev.hpp"
#include "ff/montx_dev.hpp"
#include <hip/hip_runtime.h>
#include <cstdio>
using namespace sppark_amd;
typedef montx_dev<bls12_381_fp_p, 28> f30;
typedef mont_dev<bls12_381_fp_p> f32;

// op 0: a*b  1: a^2  2: norm(sub<3>(a,b))  3: norm(a+b)  4: is_zero_mod<12>(a) -> l[0]
__global__ void k_op(u32* out, const u32* a, const u32* b, unsigned n, int op)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f30 x = f30::from_wire(a + i * 14), y = f30::from_wire(b + i * 14), r;
    if (op == 0) r = x * y;
    else if (op == 1) r = x.sqr();
    else if (op == 2) r = f30::sub<3>(x, y).norm();
    else if (op == 3) r = (x + y).norm();
    else { r = f30::zero(); r.l[0] = x.is_zero_mod<14>(); }
    r.to_wire(out + i * 14);
}

// throughput: chains of dependent operations on registers
template<int WHICH>
__global__ __launch_bounds__(256) void k_bench(u32* out, const u32* a, unsigned iters)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (WHICH < 2) {
        f30 x = f30::from_wire(a + (i & 1023) * 14), y = f30::from_wire(a + ((i + 7) & 1023) * 14);
        for (unsigned k = 0; k < iters; k++) { if (WHICH == 0) x = x * y; else x = x.sqr(); }
        x.to_wire(out + i * 14);
    } else if (WHICH < 4) {
        f32 x = f32::from_wire(a + (i & 1023) * 14), y = f32::from_wire(a + ((i + 7) & 1023) * 14);
        for (unsigned k = 0; k < iters; k++) { if (WHICH == 2) x = x * y; else x = x.sqr(); }
        x.to_wire(out + i * 14);
    } else if (WHICH == 4) {
        // the fast path of a mixed addition in lazy form (bounds: ec/xyzzx_dev.hpp)
        f30 X = f30::from_wire(a + (i & 1023) * 14), Y = f30::from_wire(a + ((i + 1) & 1023) * 14);
        f30 ZZ = f30::from_wire(a + ((i + 2) & 1023) * 14), ZZZ = f30::from_wire(a + ((i + 3) & 1023) * 14);
        f30 px = f30::from_wire(a + ((i + 4) & 1023) * 14), py = f30::from_wire(a + ((i + 5) & 1023) * 14);
        for (unsigned k = 0; k < iters; k++) {
            f30 U2, S2;
            f30::mul2(U2, S2, px, ZZ, py, ZZZ);
            if (k & 1) S2 = f30::neg<3>(S2);
            f30 Pd = f30::sub<11, 6>(U2, X).norm();
            f30 Rd = f30::sub<6, 4>(S2, Y).norm();
            if (Pd.is_zero_mod<13>()) { X = Y; continue; }
            f30 PP, RR, PPP, Q, M1, M2;
            f30::sqr2(PP, RR, Pd, Rd);
            f30::mul2(PPP, Q, Pd, PP, X, PP);
            f30 T = PPP + Q + Q;
            f30 X3 = f30::sub<8, 3>(RR, T);
            f30 D = f30::sub<11, 6>(Q, X3);
            f30::mul2(M1, M2, D, Rd, Y, PPP);
            Y = f30::sub<3>(M1, M2);
            f30::mul2(ZZ, ZZZ, ZZ, PP, ZZZ, PPP);
            X = X3;
        }
        (X + Y + ZZ + ZZZ).to_wire(out + i * 14);
    } else {
        // the same in the 32-bit-limb canonical form (what ec/xyzz_dev.hpp::madd does)
        f32 X = f32::from_wire(a + (i & 1023) * 14), Y = f32::from_wire(a + ((i + 1) & 1023) * 14);
        f32 ZZ = f32::from_wire(a + ((i + 2) & 1023) * 14), ZZZ = f32::from_wire(a + ((i + 3) & 1023) * 14);
        f32 px = f32::from_wire(a + ((i + 4) & 1023) * 14), py = f32::from_wire(a + ((i + 5) & 1023) * 14);
        for (unsigned k = 0; k < iters; k++) {
            f32 y2 = py.cneg(k & 1);
            f32 Pd = px * ZZ - X, Rd = y2 * ZZZ - Y;
            if (Pd.is_zero()) { X = Y; continue; }
            f32 PP = Pd.sqr(), PPP = Pd * PP, Q = X * PP;
            f32 X3 = Rd.sqr() - PPP - Q - Q;
            f32 Y3 = Rd * (Q - X3) - Y * PPP;
            ZZ = ZZ * PP; ZZZ = ZZZ * PPP; X = X3; Y = Y3;
        }
        (X + Y + ZZ + ZZZ).to_wire(out + i * 14);
    }
}

extern "C" int exp_op(int op, void* out, const void* a, const void* b, unsigned n)
{
    u32 *da, *db, *dout; size_t bytes = (size_t)n * 14 * 4;
    hipMalloc((void**)&da, bytes); hipMalloc((void**)&db, bytes); hipMalloc((void**)&dout, bytes);
    hipMemcpy(da, a, bytes, hipMemcpyHostToDevice); hipMemcpy(db, b, bytes, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_op, dim3((n + 255) / 256), dim3(256), 0, 0, dout, da, db, n, op);
    hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dout);
    return (int)hipGetLastError();
}
extern "C" float exp_bench(int which, const void* a, unsigned blocks, unsigned iters)
{
    u32 *da, *dout; size_t n = (size_t)blocks * 256;
    hipMalloc((void**)&da, 1024 * 14 * 4); hipMalloc((void**)&dout, n * 14 * 4);
    hipMemcpy(da, a, 1024 * 14 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        switch (which) {
            case 0: hipLaunchKernelGGL(k_bench<0>, dim3(blocks), dim3(256), 0, 0, dout, da, iters); break;
            case 1: hipLaunchKernelGGL(k_bench<1>, dim3(blocks), dim3(256), 0, 0, dout, da, iters); break;
            case 2: hipLaunchKernelGGL(k_bench<2>, dim3(blocks), dim3(256), 0, 0, dout, da, iters); break;
            case 3: hipLaunchKernelGGL(k_bench<3>, dim3(blocks), dim3(256), 0, 0, dout, da, iters); break;
            case 4: hipLaunchKernelGGL(k_bench<4>, dim3(blocks), dim3(256), 0, 0, dout, da, iters); break;
            default: hipLaunchKernelGGL(k_bench<5>, dim3(blocks), dim3(256), 0, 0, dout, da, iters); break;
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipFree(da); hipFree(dout);
    return best;
}
