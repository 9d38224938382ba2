// Where do k_scatterA_staged's 5.7 ms go?  Copy of the kernel with phases switched off by a template
// mask, on synthetic digits (2^26 points, 12 windows, 4096 partitions, 64 slabs).
//   bit 0: no digit loads   bit 1: no histogram atomics   bit 2: no placement (linear LDS image instead)
//   bit 3: no global write-out   bit 4: no scan   bit 5: __syncthreads instead of the LDS-only barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32;
#define SPPARK_DEVFN __device__ __forceinline__
static constexpr unsigned NT = 1024;

template<int OFF> SPPARK_DEVFN void bar()
{
    if (OFF & 32) __syncthreads();
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template<int OFF, int PER>
__global__ __launch_bounds__(NT)
void k_scat(uint2* __restrict__ partA, const u32* __restrict__ digits, const u32* __restrict__ H, const u32* __restrict__ offA,
            unsigned n, unsigned nslabs, unsigned slab_sz, unsigned NA, unsigned LB)
{
    extern __shared__ u32 lds_sa[];
    constexpr unsigned TILE = PER * NT;
    u32* cnt = lds_sa; u32* G = cnt + NA; u32* wsum = G + NA;
    uint2* stage = reinterpret_cast<uint2*>(wsum + 16);
    const unsigned slab = blockIdx.x, w = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32* h = H + ((size_t)w * nslabs + slab) * NA;
    const u32* o = offA + (size_t)w * (NA + 1);
    u32 cur[4];
    #pragma unroll
    for (int i = 0; i < 4; i++) { unsigned b = 4 * tid + i; cur[i] = b < NA ? h[b] + o[b] : 0; if (b < NA) cnt[b] = 0; }
    const unsigned lo = slab * slab_sz, hi = min(n, lo + slab_sz);
    const u32* dig = digits + (size_t)w * n;
    uint2* dst = partA + (size_t)w * n;
    const u32 lomask = (1u << LB) - 1;
    u32 d[PER], dn[PER];
    #pragma unroll
    for (int u = 0; u < PER; u++) { unsigned j = lo + u * NT + tid; dn[u] = (OFF & 1) ? (j * 2654435761u >> 11) + 1 : (j < hi ? dig[j] : 0); }
    __syncthreads();
    for (unsigned t0 = lo; t0 < hi; t0 += TILE) {
        #pragma unroll
        for (int u = 0; u < PER; u++) d[u] = dn[u];
        #pragma unroll
        for (int u = 0; u < PER; u++) { unsigned j = t0 + TILE + u * NT + tid; dn[u] = (OFF & 1) ? (j * 2654435761u >> 11) + 1 : (j < hi ? dig[j] : 0); }
        if (!(OFF & 2)) {
            #pragma unroll
            for (int u = 0; u < PER; u++) if (d[u]) atomicAdd(&cnt[((d[u] & 0x7fffffffu) - 1) >> LB], 1u);
        }
        bar<OFF>();
        u32 total = TILE;
        if (!(OFF & 16)) {
            u32 c[4], s = 0;
            #pragma unroll
            for (int i = 0; i < 4; i++) { unsigned b = 4 * tid + i; c[i] = b < NA ? cnt[b] : 0; s += c[i]; }
            u32 incl = s;
            #pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) { u32 v = __shfl_up(incl, dlt); if (lane >= (unsigned)dlt) incl += v; }
            if (lane == 63) wsum[wave] = incl;
            bar<OFF>();
            u32 wv = lane < 16 ? wsum[lane] : 0, before = lane < wave ? wv : 0; total = wv;
            #pragma unroll
            for (int m = 1; m < 16; m <<= 1) { before += __shfl_xor(before, m); total += __shfl_xor(total, m); }
            before = __shfl(before, 0); total = __shfl(total, 0);
            u32 run = before + incl - s;
            #pragma unroll
            for (int i = 0; i < 4; i++) {
                unsigned b = 4 * tid + i;
                if (b < NA) { cnt[b] = run; G[b] = cur[i] - run; }
                cur[i] += c[i]; run += c[i];
            }
            bar<OFF>();
        }
        if (!(OFF & 4)) {
            #pragma unroll
            for (int u = 0; u < PER; u++) {
                if (d[u]) {
                    u32 k = (d[u] & 0x7fffffffu) - 1, p = k >> LB;
                    u32 pos = atomicAdd(&cnt[p], 1u) % TILE;
                    stage[pos] = make_uint2((t0 + u * NT + tid) | (d[u] & 0x80000000u), (p << 16) | (k & lomask));
                }
            }
        } else {
            #pragma unroll
            for (int u = 0; u < PER; u++) { u32 k = (d[u] & 0x7fffffffu) - 1; stage[u * NT + tid] = make_uint2(t0 + u * NT + tid, ((k >> LB) << 16) | (k & lomask)); }
        }
        bar<OFF>();
        #pragma unroll
        for (int i = 0; i < 4; i++) { unsigned b = 4 * tid + i; if (b < NA) cnt[b] = 0; }
        #pragma unroll
        for (int u = 0; u < PER; u++) {
            unsigned e = u * NT + tid;
            if (e < total) {
                uint2 v = stage[e];
                if (!(OFF & 8)) dst[(G[v.y >> 16] + e) % n] = make_uint2(v.x, v.y & 0xffffu);
                else if (v.x == 0xdeadbeefu) dst[0] = v;
            }
        }
        bar<OFF>();
    }
}

template<int OFF, int PER> static float run(uint2* partA, u32* digits, u32* H, u32* offA, unsigned n, unsigned nslabs, unsigned NA, unsigned LB, unsigned W)
{
    size_t lds = (size_t)NA * 8 + 64 + (size_t)PER * NT * 8;
    hipFuncSetAttribute((const void*)k_scat<OFF, PER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_scat<OFF, PER>), dim3(nslabs, W), dim3(NT), lds, 0, partA, digits, H, offA, n, nslabs, n / nslabs, NA, LB);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

__global__ void k_fill(u32* digits, size_t total)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) { u32 h = (u32)i * 2654435761u; h ^= h >> 13; h *= 0x85ebca6bu; digits[i] = ((h >> 11) & 0x1fffffu) + 1; }
}

int main()
{
    const unsigned n = 1u << 26, W = 12, NA = 4096, LB = 9, nslabs = 64;
    uint2* partA; u32 *digits, *H, *offA;
    hipMalloc(&partA, (size_t)n * W * 8); hipMalloc(&digits, (size_t)n * W * 4);
    hipMalloc(&H, (size_t)W * nslabs * NA * 4); hipMalloc(&offA, (size_t)(NA + 1) * W * 4);
    hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)n * W + 255) / 256)), dim3(256), 0, 0, digits, (size_t)n * W);
    // destination offsets: every (slab, partition) gets its nominal share, the writes wrap modulo n (bench only)
    std::vector<u32> h((size_t)W * nslabs * NA), oa((size_t)(NA + 1) * W);
    for (unsigned w = 0; w < W; w++) for (unsigned s = 0; s < nslabs; s++) for (unsigned k = 0; k < NA; k++) h[((size_t)w * nslabs + s) * NA + k] = s * (n / NA / nslabs);
    for (unsigned w = 0; w < W; w++) for (unsigned k = 0; k <= NA; k++) oa[(size_t)w * (NA + 1) + k] = (u32)((size_t)k * n / NA);
    hipMemcpy(H, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(offA, oa.data(), oa.size() * 4, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
#define RUN(OFF, PER, what) printf("%-64s %7.3f ms\n", what, run<OFF, PER>(partA, digits, H, offA, n, nslabs, NA, LB, W));
    RUN(0, 14, "full kernel, 14 entries per lane and tile")
    RUN(32, 14, "with __syncthreads barriers")
    RUN(1, 14, "no digit loads")
    RUN(2, 14, "no histogram atomics")
    RUN(4, 14, "no placement atomics (linear LDS image)")
    RUN(8, 14, "no global write-out")
    RUN(16 | 2, 14, "no histogram, no scan")
    RUN(2 | 4 | 16, 14, "loads + LDS image + write-out only")
    RUN(1 | 8, 14, "LDS work only")
    RUN(1 | 2 | 4 | 8 | 16, 14, "shell (linear LDS image, barriers)")
    RUN(0, 8, "full kernel, 8 entries per lane and tile")
    RUN(0, 12, "full kernel, 12 entries per lane and tile")
    return 0;
}
