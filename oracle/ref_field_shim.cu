// ORACLE / TEST INFRASTRUCTURE ONLY -- never linked into, loaded by or shipped with the product.
//
// The reference's OWN device field classes, built for gfx950 by its own HIP path (rust/src/build.rs:69-112: hipcc,
// -include util/cuda2hip.hpp) from the headers where they lie under /root/reference, into
// oracle/_ref/libref_field_<curve>.so (oracle/Makefile: ref_field).  Under __HIPCC__ the reference defines
// fp_t and fr_t over ff/mont_t.hip (ff/bls12-381.hpp:63-83, ff/alt_bn128.hpp:60-82, ff/bls12-377.hpp:61-85,
// ff/pasta.hpp:55-79); that class has + - * sqr() to() from() (ff/mont_t.hip:96-218) and needs nothing the image lacks.
//
// This is the one layer of the MSM side that the reference itself can pin here (its host field comes from blst, which is
// not vendored; its point classes and msm/pippenger.cuh are not part of its HIP path).  tests/test_field_vs_reference_gpu.py
// holds the product's device fields -- ff/mont_dev.hpp and, through from_std / to_std, ff/montx_dev.hpp -- against it
// bit for bit on the same MI355X.  Nothing here restates arithmetic: the kernel only calls the reference's operators.
#if defined(FEATURE_BLS12_381)
# include <ff/bls12-381.hpp>
#elif defined(FEATURE_BLS12_377)
# include <ff/bls12-377.hpp>
#elif defined(FEATURE_PALLAS) || defined(FEATURE_VESTA)
# include <ff/pasta.hpp>
#elif defined(FEATURE_BN254)
# include <ff/alt_bn128.hpp>
#else
# error "no FEATURE"
#endif
#include <hip/hip_runtime.h>

#define REF_FFI extern "C" __attribute__((visibility("default")))

// op 0: a + b   1: a - b   2: a * b   3: a.sqr()   4: a.to()   5: a.from()      (ff/mont_t.hip:96-218)
template<class F>
__global__ void k_ref_field_op(F* out, const F* a, const F* b, unsigned n, int op)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F x = a[i], y = b[i];
    switch (op) {
        case 0: x += y; break;
        case 1: x -= y; break;
        case 2: x = x * y; break;
        case 3: x.sqr(); break;
        case 4: x.to(); break;
        default: x.from(); break;
    }
    out[i] = x;
}

template<class F>
static int run(int op, void* out, const void* a, const void* b, size_t n)
{
    size_t bytes = n * sizeof(F);
    F *d_a = nullptr, *d_b = nullptr, *d_o = nullptr;
    int rc = 0;
    if (hipMalloc((void**)&d_a, bytes) != hipSuccess || hipMalloc((void**)&d_b, bytes) != hipSuccess ||
        hipMalloc((void**)&d_o, bytes) != hipSuccess) rc = -1;
    if (!rc && hipMemcpy(d_a, a, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = -2;
    if (!rc && hipMemcpy(d_b, b ? b : a, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = -2;
    if (!rc) {
        hipLaunchKernelGGL(k_ref_field_op<F>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, d_o, d_a, d_b, (unsigned)n, op);
        if (hipGetLastError() != hipSuccess) rc = -3;
    }
    if (!rc && hipMemcpy(out, d_o, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = -4;
    (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_o);
    return rc;
}

REF_FFI size_t ref_field_bytes(int field) { return field == 0 ? sizeof(fp_t) : sizeof(fr_t); }

// field 0 = the curve's base field fp_t, 1 = its scalar field fr_t; host buffers of n elements in the reference's own
// memory image (little-endian 32-bit words of the Montgomery form)
REF_FFI int ref_field_op(int field, int op, void* out, const void* a, const void* b, size_t n)
{
    return field == 0 ? run<fp_t>(op, out, a, b, n) : run<fr_t>(op, out, a, b, n);
}
