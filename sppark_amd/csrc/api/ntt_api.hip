// C-ABI of the NTT library (one .so per field: -DFEATURE_GOLDILOCKS /
// -DFEATURE_BABY_BEAR, as poc/ntt-cuda/build.rs selects them).  Declarations +
// reference citations: include/sppark_amd.h.
#include "../ntt/field_select.hpp"
namespace sppark_amd {
#define SPPARK_NTT_EXTERN(DIF, INV, R1, R2) \
    extern template __global__ void k_ntt_pass<ntt_fr_t, DIF, INV, R1, R2>(ntt_fr_t*, ntt_tables<ntt_fr_t>, ntt_pass);
SPPARK_NTT_PASS_ALL(SPPARK_NTT_EXTERN, true) SPPARK_NTT_PASS_ALL(SPPARK_NTT_EXTERN, false)
#define SPPARK_NTT_SMALL_EXTERN(INV, LGC) \
    extern template __global__ void k_ntt_small<ntt_fr_t, INV, LGC>(ntt_fr_t*, ntt_tables<ntt_fr_t>, ntt_tables<ntt_fr_t>, unsigned);
#if defined(FEATURE_GOLDILOCKS) || defined(FEATURE_BABY_BEAR)
SPPARK_NTT_SMALL_ALL_NARROW(SPPARK_NTT_SMALL_EXTERN)
#else
SPPARK_NTT_SMALL_ALL_WIDE(SPPARK_NTT_SMALL_EXTERN)
#endif
#if defined(FEATURE_GOLDILOCKS) || defined(FEATURE_BABY_BEAR)
SPPARK_NTT_PASS_ALL_BIG(SPPARK_NTT_EXTERN, true) SPPARK_NTT_PASS_ALL_BIG(SPPARK_NTT_EXTERN, false)
#else
#define SPPARK_NTT_LAT_EXTERN(DIF, INV) \
    extern template __global__ void k_ntt_pass_lat<ntt_fr_t, DIF, INV>(ntt_fr_t*, ntt_tables<ntt_fr_t>, ntt_pass);
SPPARK_NTT_LAT_EXTERN(true, false) SPPARK_NTT_LAT_EXTERN(true, true) SPPARK_NTT_LAT_EXTERN(false, false) SPPARK_NTT_LAT_EXTERN(false, true)
#endif
}
#if defined(FEATURE_GOLDILOCKS) || defined(FEATURE_BABY_BEAR)      // the radix-64 plan: ntt/k_ntt_r64.hip
#include "../ntt/ntt_r64_kernels.hpp"
namespace sppark_amd {
#define SPPARK_R64_EXTERN(K, DIF, INV) \
    extern template __global__ void K<ntt_fr_t, DIF, INV>(ntt_fr_t*, ntt_r64_args<ntt_fr_t>);
SPPARK_R64_EXTERN(k_ntt6, true, false) SPPARK_R64_EXTERN(k_ntt6, true, true) SPPARK_R64_EXTERN(k_ntt6, false, false) SPPARK_R64_EXTERN(k_ntt6, false, true)
SPPARK_R64_EXTERN(k_ntt12, true, false) SPPARK_R64_EXTERN(k_ntt12, true, true) SPPARK_R64_EXTERN(k_ntt12, false, false) SPPARK_R64_EXTERN(k_ntt12, false, true)
extern template __global__ void k_ntt12<ntt_fr_t, false, false, true>(ntt_fr_t*, ntt_r64_args<ntt_fr_t>);
}
#endif
#include "../ntt/ntt_driver.hpp"
#ifdef SPPARK_NTT_WITH_MSM            // same .so as msm_api.hip, which already defines the common symbols
# define SPPARK_FFI extern "C" __attribute__((visibility("default")))
#else
# include "common_api.hpp"
#endif

using namespace sppark_amd;
typedef ntt_fr_t fr_t;

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
    pooled_scratch buf(bytes);                  // (util/runtime.hpp: kept between calls)
    fr_t* d = (fr_t*)buf.p;
    HIP_OK(hipMemcpyAsync(d, inout, bytes, hipMemcpyHostToDevice, stream));
    ntt_engine<fr_t>::instance().run(gpu, d, lg, order, direction, type, stream);
    HIP_OK(hipMemcpyAsync(inout, d, bytes, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    buf.done();
}

SPPARK_FFI RustError compute_ntt(size_t device_id, void* inout, uint32_t lg_domain_size,
                                 int ntt_order, int ntt_direction, int ntt_type)
{   return guarded([&] { ntt_any(device_id, inout, lg_domain_size, ntt_order, ntt_direction, ntt_type, nullptr); });   }

SPPARK_FFI RustError sppark_ntt(size_t device_id, void* inout, uint32_t lg_domain_size,
                                int ntt_order, int ntt_direction, int ntt_type, void* stream)
{   return guarded([&] { ntt_any(device_id, inout, lg_domain_size, ntt_order, ntt_direction, ntt_type, (hipStream_t)stream); });   }

// ---- low-degree extension (C++-only in the reference: NTT::LDE / LDE_aux / LDE_powers / LDE_expand) ----
static void lde_any(size_t device_id, void* inout, uint32_t lg_domain, uint32_t lg_blowup, void* aux_out, hipStream_t stream)
{
    const gpu_info& gpu = select_gpu((int)device_id);
    const size_t dom = (size_t)1 << lg_domain, ext = dom << lg_blowup;
    const bool dev = is_device_pointer(inout), aux_dev = aux_out && is_device_pointer(aux_out);
    // scratch: [tmp: dom][aux: dom, when it has to be staged][ext, when inout is a host buffer]
    const size_t need = dom + (aux_out && !aux_dev ? dom : 0) + (dev ? 0 : ext);
    pooled_scratch buf(need * sizeof(fr_t));
    fr_t* scratch = (fr_t*)buf.p;
    {
        fr_t* d_tmp = scratch;
        fr_t* d_aux = aux_out ? (aux_dev ? (fr_t*)aux_out : scratch + dom) : nullptr;
        fr_t* d_ext = dev ? (fr_t*)inout : scratch + need - ext;
        if (!dev) HIP_OK(hipMemcpyAsync(d_ext, inout, dom * sizeof(fr_t), hipMemcpyHostToDevice, stream));
        ntt_engine<fr_t>::instance().lde(gpu, d_ext, d_tmp, d_aux, lg_domain, lg_blowup, stream);
        if (aux_out && !aux_dev) HIP_OK(hipMemcpyAsync(aux_out, d_aux, dom * sizeof(fr_t), hipMemcpyDeviceToHost, stream));
        if (!dev) HIP_OK(hipMemcpyAsync(inout, d_ext, ext * sizeof(fr_t), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
    }
    buf.done();
}
SPPARK_FFI void sppark_ntt_release_cached(void)
{
    dev_scratch_pool::instance().release();
    ntt_engine<fr_t>::instance().release_tables();
}
// diagnostics for the tests: idle scratch bytes of this library's pool, cached twiddle tables
SPPARK_FFI size_t sppark_ntt_cached_scratch_bytes(void) { return dev_scratch_pool::instance().idle_bytes(); }
SPPARK_FFI size_t sppark_ntt_cached_tables(void) { return ntt_engine<fr_t>::instance().cached_table_count(); }

SPPARK_FFI RustError sppark_lde(size_t device_id, void* inout, uint32_t lg_domain_size, uint32_t lg_blowup,
                                void* aux_out, void* stream)
{   return guarded([&] { lde_any(device_id, inout, lg_domain_size, lg_blowup, aux_out, (hipStream_t)stream); });   }

SPPARK_FFI RustError sppark_lde_powers(size_t device_id, void* d_inout, uint32_t lg_domain_size, void* stream)
{
    return guarded([&] {
        if (!is_device_pointer(d_inout)) HIP_OK(hipErrorInvalidValue);
        ntt_engine<fr_t>::instance().lde_powers(select_gpu((int)device_id), (fr_t*)d_inout, lg_domain_size, (hipStream_t)stream);
        if (stream == nullptr) HIP_OK(hipStreamSynchronize(nullptr));
    });
}

SPPARK_FFI RustError sppark_lde_expand(size_t device_id, void* d_out, const void* d_in, uint32_t lg_domain_size,
                                       uint32_t lg_blowup, void* stream)
{
    return guarded([&] {
        if (!is_device_pointer(d_out) || !is_device_pointer(d_in)) HIP_OK(hipErrorInvalidValue);
        ntt_engine<fr_t>::instance().lde_spread(select_gpu((int)device_id), (fr_t*)d_out, (const fr_t*)d_in,
                                                lg_domain_size, lg_blowup, false, (hipStream_t)stream);
        if (stream == nullptr) HIP_OK(hipStreamSynchronize(nullptr));
    });
}

#include "poly_api.hpp"
