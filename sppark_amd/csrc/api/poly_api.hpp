// C entry points of the polynomial primitives over the library's field fr_t (included by
// api/ntt_api.hip for the NTT-capable libraries and by api/poly_only_api.hip for fields that have
// no NTT parameters in the reference: Mersenne31, the BabyBear quartic extension).
// Expects: fr_t, guarded(), SPPARK_FFI, select_gpu, is_device_pointer in scope.
#pragma once
#include "../poly/poly_driver.hpp"

// ---- polynomial primitives over this library's field (C++ templates only in the reference:
// polynomial/prefix_op.cuh, evaluate.cuh, div_by_x_minus_z.cuh) ---------------------------------
namespace {
// device view of a host-or-device buffer: copied in when it lives on the host, copied back on flush()
struct dev_view {
    fr_t* d = nullptr; void* host = nullptr; size_t bytes = 0, got = 0; bool owned = false, synced = false; int dev = 0; hipStream_t stream;
    dev_view(const void* p, size_t count, bool copy_in, hipStream_t s) : bytes(count * sizeof(fr_t)), stream(s)
    {
        if (count == 0) return;
        if (p == nullptr) HIP_OK(hipErrorInvalidValue);
        if (is_device_pointer(p)) { d = (fr_t*)p; return; }
        host = (void*)p; owned = true;
        HIP_OK(hipGetDevice(&dev));
        d = (fr_t*)dev_scratch_pool::instance().take(dev, bytes, got);      // (util/runtime.hpp: kept between calls)
        if (copy_in) HIP_OK(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, stream));
    }
    void flush() { if (owned && bytes) { HIP_OK(hipMemcpyAsync(host, d, bytes, hipMemcpyDeviceToHost, stream)); HIP_OK(hipStreamSynchronize(stream)); synced = true; } }
    void done() { synced = true; }             // the caller has synchronised the stream (an input-only view)
    ~dev_view() { if (!(owned && d)) return; if (synced) dev_scratch_pool::instance().give(dev, d, got); else (void)hipFree(d); }
    dev_view(const dev_view&) = delete;
};
}

SPPARK_FFI RustError sppark_prefix_op(size_t device_id, void* out, const void* inp, size_t len, int op, void* stream)
{
    return guarded([&] {
        (void)select_gpu((int)device_id);
        hipStream_t s = (hipStream_t)stream;
        if (len == 0) return;
        dev_view vin(inp, len, true, s);
        if (out == inp) {
            poly_engine<fr_t>::prefix_op(vin.d, vin.d, len, op, s);
            vin.flush();
        } else {
            dev_view vout(out, len, false, s);
            poly_engine<fr_t>::prefix_op(vout.d, vin.d, len, op, s);     // (synchronises the stream)
            vin.done();
            vout.flush();
        }
    });
}

SPPARK_FFI RustError sppark_poly_evaluate(size_t device_id, void* ret, const void* x, size_t n,
                                          const void* coeffs, size_t len, void* stream)
{
    return guarded([&] {
        (void)select_gpu((int)device_id);
        hipStream_t s = (hipStream_t)stream;
        if (n == 0) return;
        dev_view vx(x, n, true, s), vc(coeffs, len, true, s), vr(ret, n, false, s);
        poly_engine<fr_t>::evaluate(vr.d, vx.d, n, vc.d, len, s);       // (synchronises the stream)
        vx.done(); vc.done();
        vr.flush();
    });
}

SPPARK_FFI RustError sppark_div_by_x_minus_z(size_t device_id, void* inout, size_t len, const void* z, int rotate, void* stream)
{
    return guarded([&] {
        (void)select_gpu((int)device_id);
        hipStream_t s = (hipStream_t)stream;
        if (len == 0) return;
        if (z == nullptr) HIP_OK(hipErrorInvalidValue);
        fr_t zv;
        if (is_device_pointer(z)) HIP_OK(hipMemcpy(&zv, z, sizeof(zv), hipMemcpyDeviceToHost));
        else memcpy(&zv, z, sizeof(zv));
        dev_view v(inout, len, true, s);
        poly_engine<fr_t>::div_by_x_minus_z(v.d, len, zv, rotate != 0, s);
        v.flush();
    });
}
