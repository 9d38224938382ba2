// ORACLE / TEST INFRASTRUCTURE ONLY -- never linked into, loaded by or shipped with the product.
//
// The reference's OWN NTT, built for gfx950 by its own HIP path (rust/src/build.rs:69-112: hipcc, -include
// util/cuda2hip.hpp) from the sources where they lie under /root/reference, into oracle/_ref/libref_ntt_<field>.so
// (oracle/Makefile: ref_ntt).  The library holds the reference's compute_ntt (poc/ntt-cuda/cuda/ntt_api.cu, compiled as
// it is) and util/all_gpus.cpp; THIS file adds what ntt_api.cu does not export -- the device-pointer and LDE forms of the
// same class (ntt/ntt.cuh:215-365) and a timed loop -- so that the GPU tests can hold the HIP kernels of sppark_amd
// against the reference's kernels RUNNING ON THE SAME MI355X, bit for bit, and bench.py can time them beside each other.
// Nothing here restates the algorithm: every entry point forwards to NTT::*.
#if defined(FEATURE_BLS12_381)
# include <ff/bls12-381.hpp>
#elif defined(FEATURE_BLS12_377)
# include <ff/bls12-377.hpp>
#elif defined(FEATURE_PALLAS) || defined(FEATURE_VESTA)
# include <ff/pasta.hpp>
#elif defined(FEATURE_BN254)
# include <ff/alt_bn128.hpp>
#elif defined(FEATURE_GOLDILOCKS)
# include <ff/goldilocks.hpp>
#elif defined(FEATURE_BABY_BEAR)
# include <ff/baby_bear.hpp>
#else
# error "no FEATURE"
#endif
#include <ntt/ntt.cuh>

#define REF_FFI extern "C" __attribute__((visibility("default")))

REF_FFI size_t ref_ntt_elem_bytes() { return sizeof(fr_t); }

// ntt/ntt.cuh:344-350 on memory the CALLER owns on the device; synchronous
REF_FFI int ref_ntt_dev(fr_t* d_inout, uint32_t lg, int order, int direction, int type)
{
    try {
        auto& gpu = select_gpu(0);
        gpu.select();
        stream_t& s = gpu;
        NTT::Base_dev_ptr(s, d_inout, lg, (NTT::InputOutputOrder)order, (NTT::Direction)direction, (NTT::Type)type);
        s.sync();
        return 0;
    } catch (const cuda_error& e) { return e.code() ? e.code() : -1; }
}

// |iters| back-to-back transforms of the same device buffer between two events on the reference's stream:
// *ms = the average of one.  (The data is transformed |iters| times: the caller times, it does not check, here.)
REF_FFI int ref_ntt_dev_timed(fr_t* d_inout, uint32_t lg, int order, int direction, int type, int iters, float* ms)
{
    try {
        auto& gpu = select_gpu(0);
        gpu.select();
        stream_t& s = gpu;
        NTT::Base_dev_ptr(s, d_inout, lg, (NTT::InputOutputOrder)order, (NTT::Direction)direction, (NTT::Type)type);   // warm-up
        s.sync();
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
        (void)hipEventRecord(a, (hipStream_t)s);
        for (int i = 0; i < iters; i++)
            NTT::Base_dev_ptr(s, d_inout, lg, (NTT::InputOutputOrder)order, (NTT::Direction)direction, (NTT::Type)type);
        (void)hipEventRecord(b, (hipStream_t)s);
        (void)hipEventSynchronize(b);
        float t = 0;
        (void)hipEventElapsedTime(&t, a, b);
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        *ms = t / (iters > 0 ? iters : 1);
        return 0;
    } catch (const cuda_error& e) { return e.code() ? e.code() : -1; }
}

// ntt/ntt.cuh:338-342 / :280-336: host buffer of 2^(lg + lg_blowup) elements, the first 2^lg hold the input
REF_FFI int ref_lde(fr_t* inout, uint32_t lg, uint32_t lg_blowup)
{
    try {
        RustError e = NTT::LDE(select_gpu(0), inout, lg, lg_blowup);
        free(e.message);
        return e.code;
    } catch (const cuda_error& e) { return e.code() ? e.code() : -1; }
}
REF_FFI int ref_lde_aux(fr_t* inout, uint32_t lg, uint32_t lg_blowup, fr_t* aux_out)
{
    try {
        RustError e = NTT::LDE_aux(select_gpu(0), inout, lg, lg_blowup, aux_out);
        free(e.message);
        return e.code;
    } catch (const cuda_error& e) { return e.code() ? e.code() : -1; }
}
