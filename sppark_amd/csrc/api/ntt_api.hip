// C-ABI of the NTT library (one .so per field: -DFEATURE_GOLDILOCKS /
// -DFEATURE_BABY_BEAR, as poc/ntt-cuda/build.rs selects them).  Declarations +
// reference citations: include/sppark_amd.h.
#include "../ntt/ntt_driver.hpp"
#include "common_api.hpp"

using namespace sppark_amd;

#if defined(FEATURE_GOLDILOCKS)
typedef gl64_dev fr_t;
#elif defined(FEATURE_BABY_BEAR)
typedef bb31_dev fr_t;
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
