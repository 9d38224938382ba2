// Counting-sort kernels of the MSM pipeline (non-template; included by exactly
// one translation unit, api/msm_api.hip).  See msm_kernels.hpp for the overview.
#pragma once
#include "../ff/mont_dev.hpp"

namespace sppark_amd {

// ---------------------------------------------------------------------------
// hist: block (slab, window) counts its slab's digits of that window in LDS.
// H[(w*nslabs + slab)*NB + b] = count.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void k_hist(u32* __restrict__ H, const u32* __restrict__ digits, unsigned n,
            unsigned nslabs, unsigned slab_sz, unsigned NB)
{
    extern __shared__ u32 lds_cnt[];
    const unsigned slab = blockIdx.x, w = blockIdx.y;
    for (unsigned b = threadIdx.x; b < NB; b += blockDim.x) lds_cnt[b] = 0;
    __syncthreads();

    const unsigned lo = slab * slab_sz, hi = min(n, lo + slab_sz);
    const u32* dig = digits + (size_t)w * n;
    for (unsigned i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        u32 d = dig[i];
        if (d) atomicAdd(&lds_cnt[(d & 0x7fffffffu) - 1], 1u);
    }
    __syncthreads();

    u32* out = H + ((size_t)w * nslabs + slab) * NB;
    for (unsigned b = threadIdx.x; b < NB; b += blockDim.x) out[b] = lds_cnt[b];
}

// slab-exclusive prefix per (window, bucket); tot[w*NB+b] = bucket total
__global__ __launch_bounds__(256)
void k_scan_slabs(u32* __restrict__ H, u32* __restrict__ tot, unsigned nslabs, unsigned NB, unsigned nwins)
{
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)nwins * NB) return;
    const unsigned w = id / NB, b = id % NB;
    u32 run = 0;
    for (unsigned s = 0; s < nslabs; s++) {
        u32* p = H + ((size_t)w * nslabs + s) * NB + b;
        u32 t = *p; *p = run; run += t;
    }
    tot[id] = run;
}

// bucket-exclusive prefix per window: off[w*(NB+1) + b], off[..+NB] = #entries
__global__ __launch_bounds__(1024)
void k_scan_buckets(u32* __restrict__ off, const u32* __restrict__ tot, unsigned NB)
{
    __shared__ u32 part[1024];
    const unsigned w = blockIdx.x, tid = threadIdx.x;
    const unsigned per = (NB + 1023) / 1024;
    const unsigned lo = min(NB, tid * per), hi = min(NB, lo + per);
    const u32* t = tot + (size_t)w * NB;
    u32 sum = 0;
    for (unsigned b = lo; b < hi; b++) sum += t[b];
    part[tid] = sum;
    __syncthreads();
    for (unsigned d = 1; d < 1024; d <<= 1) {
        u32 v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    u32 run = part[tid] - sum;
    u32* o = off + (size_t)w * (NB + 1);
    for (unsigned b = lo; b < hi; b++) { o[b] = run; run += t[b]; }
    if (tid == 1023) o[NB] = part[1023];
}

// ---------------------------------------------------------------------------
// scatter: LDS cursors = slab-exclusive + bucket-exclusive offsets.
// sorted[w*n + pos] = point index | sign<<31
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void k_scatter(u32* __restrict__ sorted, const u32* __restrict__ digits,
               const u32* __restrict__ H, const u32* __restrict__ off,
               unsigned n, unsigned nslabs, unsigned slab_sz, unsigned NB)
{
    extern __shared__ u32 lds_cur[];
    const unsigned slab = blockIdx.x, w = blockIdx.y;
    const u32* h = H + ((size_t)w * nslabs + slab) * NB;
    const u32* o = off + (size_t)w * (NB + 1);
    for (unsigned b = threadIdx.x; b < NB; b += blockDim.x) lds_cur[b] = h[b] + o[b];
    __syncthreads();

    const unsigned lo = slab * slab_sz, hi = min(n, lo + slab_sz);
    const u32* dig = digits + (size_t)w * n;
    u32* dst = sorted + (size_t)w * n;
    for (unsigned i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        u32 d = dig[i];
        if (d) {
            u32 pos = atomicAdd(&lds_cur[(d & 0x7fffffffu) - 1], 1u);
            dst[pos] = i | (d & 0x80000000u);
        }
    }
}

} // namespace sppark_amd
