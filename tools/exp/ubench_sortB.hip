// Where do k_sortB's 6 ms go?  A copy of the kernel's register path with phases switched off by a
// template mask, on synthetic level-A partitions (NA = 4096 partitions of ~16384 entries, 12 windows).
//   bit 0: no loads (synthetic keys)   bit 1: no counting atomics   bit 2: no placement (atomics + LDS image)
//   bit 3: no write-out                bit 4: no scan
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32;
#define SPPARK_DEVFN __device__ __forceinline__
static constexpr unsigned NT = 1024;
static constexpr int PER = 18;

SPPARK_DEVFN u32 block_scan_excl(u32 s, u32* wsum, u32* total)
{
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 incl = s;
    #pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) { u32 v = __shfl_up(incl, dlt); if (lane >= (unsigned)dlt) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 wv = lane < 16 ? wsum[lane] : 0, before = lane < wave ? wv : 0, tot = wv;
    #pragma unroll
    for (int m = 1; m < 16; m <<= 1) { before += __shfl_xor(before, m); tot += __shfl_xor(tot, m); }
    before = __shfl(before, 0); *total = __shfl(tot, 0);
    __syncthreads();
    return before + incl - s;
}

template<int OFF, int NTT>
__global__ __launch_bounds__(NTT) void k_sortB_x(u32* __restrict__ sorted, u32* __restrict__ off, const uint2* __restrict__ partA,
                                                const u32* __restrict__ offA, unsigned n, unsigned NA, unsigned LB)
{
    extern __shared__ u32 lds[];
    const unsigned NL = 1u << LB;
    u32* cnt = lds; u32* part = lds + NL; u32* stage = part + NTT;
    const unsigned khi = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
    const u32* oA = offA + (size_t)w * (NA + 1);
    const unsigned begin = oA[khi], end = oA[khi + 1];
    const uint2* src = partA + (size_t)w * n;
    constexpr int P = PER * 1024 / NTT;
    u32 rx[P], rk[P];
    for (unsigned b = tid; b < NL; b += NTT) cnt[b] = 0;
    __syncthreads();
    #pragma unroll
    for (int u = 0; u < P; u++) {
        unsigned j = begin + tid + u * NTT;
        uint2 v;
        if (OFF & 1) v = j < end ? make_uint2(j, (j * 2654435761u) >> (32 - 9)) : make_uint2(0, 0xffffffffu);
        else v = j < end ? src[j] : make_uint2(0, 0xffffffffu);
        rx[u] = v.x; rk[u] = v.y;
    }
    if (!(OFF & 2)) {
        #pragma unroll
        for (int u = 0; u < P; u++) if (rk[u] != 0xffffffffu) atomicAdd(&cnt[rk[u]], 1u);
    }
    __syncthreads();
    if (!(OFF & 16)) {
        const unsigned per = (NL + NTT - 1) / NTT;
        const unsigned lo = min(NL, tid * per), hi = min(NL, lo + per);
        u32 sum = 0;
        for (unsigned b = lo; b < hi; b++) sum += cnt[b];
        u32 all, run;
        if (NTT == 1024) run = begin + block_scan_excl(sum, part, &all);
        else {
            part[tid] = sum; __syncthreads();
            for (unsigned d = 1; d < NTT; d <<= 1) { u32 v = tid >= d ? part[tid - d] : 0; __syncthreads(); part[tid] += v; __syncthreads(); }
            run = begin + part[tid] - sum;
        }
        u32* o = off + (size_t)w * (((size_t)NA << LB) + 1) + ((size_t)khi << LB);
        for (unsigned b = lo; b < hi; b++) { u32 c = cnt[b]; cnt[b] = run; o[b] = run; run += c; }
        __syncthreads();
    }
    u32* dst = sorted + (size_t)w * n;
    if (!(OFF & 4)) {
        #pragma unroll
        for (int u = 0; u < P; u++)
            if (rk[u] != 0xffffffffu) stage[(atomicAdd(&cnt[rk[u]], 1u) - begin) % (PER * 1024)] = rx[u];
    } else {
        #pragma unroll
        for (int u = 0; u < P; u++) if (rk[u] != 0xffffffffu) stage[tid + u * NTT] = rx[u];
    }
    __syncthreads();
    if (!(OFF & 8)) for (unsigned i = tid; i < end - begin; i += NTT) dst[begin + i] = stage[i];
    else if (stage[tid] == 0x12345678u) dst[0] = 1;
}

template<int OFF, int NTT> static float run(u32* sorted, u32* off, uint2* partA, u32* offA, unsigned n, unsigned NA, unsigned LB, unsigned W)
{
    size_t lds = ((size_t)1 << LB) * 4 + NTT * 4 + (size_t)PER * 1024 * 4;
    hipFuncSetAttribute((const void*)k_sortB_x<OFF, NTT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_sortB_x<OFF, NTT>), dim3(NA, W), dim3(NTT), lds, 0, sorted, off, partA, offA, n, NA, LB);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

__global__ void k_fill(uint2* partA, unsigned n, unsigned W)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)n * W) { u32 h = (u32)i * 2654435761u; partA[i] = make_uint2((u32)(i % n), (h ^ (h >> 15)) & 511u); }
}

int main()
{
    const unsigned n = 1u << 26, W = 12, NA = 4096, LB = 9;
    uint2* partA; u32 *sorted, *off, *offA;
    hipMalloc(&partA, (size_t)n * W * 8); hipMalloc(&sorted, (size_t)n * W * 4);
    hipMalloc(&off, ((size_t)(NA << LB) + 1) * W * 4); hipMalloc(&offA, (size_t)(NA + 1) * W * 4);
    hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)n * W + 255) / 256)), dim3(256), 0, 0, partA, n, W);
    std::vector<u32> h((size_t)(NA + 1) * W);
    for (unsigned w = 0; w < W; w++) for (unsigned k = 0; k <= NA; k++) h[(size_t)w * (NA + 1) + k] = (u32)((size_t)k * n / NA);
    hipMemcpy(offA, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
#define RUN(OFF, NTT, what) printf("%-64s %7.3f ms\n", what, run<OFF, NTT>(sorted, off, partA, offA, n, NA, LB, W));
    RUN(0, 1024, "full kernel, 1024 lanes x 18 entries")
    RUN(1, 1024, "no global loads (synthetic keys)")
    RUN(2, 1024, "no counting atomics")
    RUN(4, 1024, "no placement atomics (linear LDS image)")
    RUN(8, 1024, "no write-out")
    RUN(16, 1024, "no scan / offsets")
    RUN(2 | 4 | 16, 1024, "loads + LDS image + write-out only (copy)")
    RUN(1 | 8, 1024, "LDS work only (no loads, no write-out)")
    RUN(1 | 2 | 4 | 8 | 16, 1024, "empty shell (launch + barriers)")
    RUN(0, 512, "full kernel, 512 lanes x 36 entries")
    RUN(0, 256, "full kernel, 256 lanes x 72 entries")
    return 0;
}
