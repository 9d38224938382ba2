// C-ABI of the NTT library (one .so per field: -DFEATURE_GOLDILOCKS /
// -DFEATURE_BABY_BEAR, as poc/ntt-cuda/build.rs selects them).  Declarations +
// reference citations: include/sppark_amd.h.
#include "../ff/params.hpp"
#include "../ntt/ntt_driver.hpp"
#ifdef SPPARK_NTT_WITH_MSM            // same .so as msm_api.hip, which already defines the common symbols
# define SPPARK_FFI extern "C" __attribute__((visibility("default")))
#else
# include "common_api.hpp"
#endif

using namespace sppark_amd;

#if defined(FEATURE_GOLDILOCKS)
typedef gl64_dev fr_t;
#elif defined(FEATURE_BABY_BEAR)
typedef bb31_dev fr_t;
#elif defined(FEATURE_BLS12_381)       // poc/ntt-cuda/cuda/ntt_api.cu:7-8 -> ff/bls12-381.hpp fr_t
typedef fr256_dev<bls12_381_fr_p> fr_t;
#elif defined(FEATURE_BN254)           // ntt_api.cu:15-16 -> ff/alt_bn128.hpp fr_t
typedef fr256_dev<alt_bn128_fr_p> fr_t;
#else
# error "no FEATURE"
#endif

template<class Fn> static RustError guarded(Fn&& fn)
{
    try { fn(); return rust_ok(); }
    catch (const hip_error& e) { (void)hipGetLastError(); return rust_err(e.code(), e.what()); }
    catch (const std::exception& e) { return rust_err(-1, e.what()); }
    catch (...) { return rust_err(-1, "unknown exception"); }
}

static void ntt_any(size_t device_id, void* inout, uint32_t lg, int order, int direction, int type, hipStream_t stream)
{
    if (lg == 0) return;
    const gpu_info& gpu = select_gpu((int)device_id);
    const size_t bytes = sizeof(fr_t) << lg;
    if (is_device_pointer(inout)) {
        ntt_engine<fr_t>::instance().run(gpu, (fr_t*)inout, lg, order, direction, type, stream);
        if (stream == nullptr) HIP_OK(hipStreamSynchronize(stream));
        return;
    }
    // host buffer: H2D, transform, D2H (NTT::Base, ntt/ntt.cuh:216-244)
    fr_t* d = nullptr;
    HIP_OK(hipMalloc((void**)&d, bytes));
    try {
        HIP_OK(hipMemcpyAsync(d, inout, bytes, hipMemcpyHostToDevice, stream));
        ntt_engine<fr_t>::instance().run(gpu, d, lg, order, direction, type, stream);
        HIP_OK(hipMemcpyAsync(inout, d, bytes, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
    } catch (...) { (void)hipFree(d); throw; }
    HIP_OK(hipFree(d));
}

SPPARK_FFI RustError compute_ntt(size_t device_id, void* inout, uint32_t lg_domain_size,
                                 int ntt_order, int ntt_direction, int ntt_type)
{   return guarded([&] { ntt_any(device_id, inout, lg_domain_size, ntt_order, ntt_direction, ntt_type, nullptr); });   }

SPPARK_FFI RustError sppark_ntt(size_t device_id, void* inout, uint32_t lg_domain_size,
                                int ntt_order, int ntt_direction, int ntt_type, void* stream)
{   return guarded([&] { ntt_any(device_id, inout, lg_domain_size, ntt_order, ntt_direction, ntt_type, (hipStream_t)stream); });   }

// ---- device test hook: element-wise field ops (tests/test_ntt_gpu.py) -----------
// op 0: a+b  1: a-b  2: a*b  3: a*2^k (gl64 only; k = b's low byte mod 192)  7/8: fused butterfly sum/difference
__global__ void k_small_field_op(fr_t* out, const fr_t* a, const fr_t* b, unsigned n, int op)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_t x = a[i], y = b[i], r = x;
    if (op == 0) r = x + y;
    else if (op == 1) r = x - y;
    else if (op == 2) r = x * y;
    else if (op == 4) { r = x; for (int k = 0; k < 12; k++) r = r * r; }
    else if (op == 5) r = field_pow(x, (u64)(*reinterpret_cast<const u32*>(&y) & 0xffff));
    else if (op == 6) { r = x; if (i & 1) { for (unsigned k = 0; k < (i & 15); k++) r = r * r + y; } }
#if defined(FEATURE_GOLDILOCKS)
    else if (op == 7 || op == 8) { fr_t sm, df; gl64_dev::bfly(x, y, sm, df); r = op == 7 ? sm : df; }
    else {
        const unsigned e = (unsigned)(y.v & 0xff) % 192;
        r = gl64_dev::mul_pow2(x, e % 96);
        if (e >= 96) r = gl64_dev::from_raw(0) - r;
    }
#else
    else if (op == 7 || op == 8) { fr_t sm, df; fr_t::bfly(x, y, sm, df); r = op == 7 ? sm : df; }
#endif
    out[i] = r;
}

SPPARK_FFI RustError sppark_devtest_small_field_op(int op, void* out, const void* a, const void* b, size_t n)
{
    return guarded([&] {
        (void)select_gpu(-1);
        size_t bytes = n * sizeof(fr_t);
        fr_t *d_a, *d_b, *d_o;
        HIP_OK(hipMalloc((void**)&d_a, bytes)); HIP_OK(hipMalloc((void**)&d_b, bytes)); HIP_OK(hipMalloc((void**)&d_o, bytes));
        HIP_OK(hipMemcpy(d_a, a, bytes, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_b, b, bytes, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_small_field_op, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_o, d_a, d_b, (unsigned)n, op);
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpy(out, d_o, bytes, hipMemcpyDeviceToHost));
        (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_o);
    });
}
