// Device test hook for the single-word NTT fields (exported as sppark_devtest_small_field_op from
// libsppark_<field>_devtest.so -- a TEST library, not part of the product libraries): element-wise
// field operations on the GPU, compared with Python big-ints by tests/test_ntt_gpu.py.
#include "../ntt/field_select.hpp"
#include "../util/runtime.hpp"

using namespace sppark_amd;
typedef ntt_fr_t fr_t;
#define SPPARK_FFI extern "C" __attribute__((visibility("default")))

template<class Fn> static RustError guarded(Fn&& fn)
{
    try { fn(); return rust_ok(); }
    catch (const hip_error& e) { (void)hipGetLastError(); return rust_err(e.code(), e.what()); }
    catch (const std::exception& e) { return rust_err(-1, e.what()); }
    catch (...) { return rust_err(-1, "unknown exception"); }
}
SPPARK_FFI void drop_error_message(char* ptr) { free(ptr); }

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
