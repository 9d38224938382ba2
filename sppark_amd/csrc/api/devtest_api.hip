// Device test hooks and micro-benchmarks (exported as sppark_devtest_* from
// libsppark_<curve>_devtest.so -- a TEST library, not part of the product libraries).
// They exist so that tests/ can check the DEVICE field / point arithmetic
// element-by-element against the oracle, and so that DESIGN.md's instruction
// costs are measured on the box rather than assumed.  Not part of the
// reference's surface.
#include "../msm/curve_select.hpp"
#include "../ec/xyzz_coop.hpp"
#include "../util/runtime.hpp"
#include <vector>

using namespace sppark_amd;

#define SPPARK_FFI extern "C" __attribute__((visibility("default")))
SPPARK_FFI void drop_error_message(char* ptr) { free(ptr); }

template<class F>
__global__ void k_field_op(u32* out, const u32* a, const u32* b, unsigned n, int op)
{
    constexpr int N = F::N;
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F x = F::from_wire(a + (size_t)i * N), y = F::from_wire(b + (size_t)i * N), r;
    switch (op) {
        case 0: r = x + y; break;
        case 1: r = x - y; break;
        case 2: r = x * y; break;
        case 3: r = x.sqr(); break;
        case 4: r = x.neg(); break;
        case 5: r = x.from(); break;
        case 6: r = x.to(); break;
        default: r = x.dbl(); break;
    }
    r.to_wire(out + (size_t)i * N);
}

// op 0: a += b (xyzz)   1: a += affine(b)   2: a -= affine(b)   3: a = 2a
__global__ void k_xyzz_op(wire_bucket_m* out, const wire_bucket_m* a, const unsigned char* b, unsigned n, int op)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    wire_bucket_d p = wire_bucket_d::load(&a[i]);
    if (op == 0) {
        p.add(wire_bucket_d::load(reinterpret_cast<const wire_bucket_m*>(b) + i));
    } else if (op == 3) {
        p.dbl();
    } else {
        affine_dev<fp_d> q = load_affine<fp_d, false>(b, i, 8 * fp_d::N);
        p.madd(q, op == 2);
    }
    p.store(&out[i]);
}

template<class Fn> static RustError guarded(Fn&& fn)
{
    try { fn(); return rust_ok(); }
    catch (const hip_error& e) { (void)hipGetLastError(); return rust_err(e.code(), e.what()); }
    catch (const std::exception& e) { return rust_err(-1, e.what()); }
}

template<class F>
static void run_field_op(int op, void* out, const void* a, const void* b, size_t n)
{
    (void)select_gpu(-1);
    size_t bytes = n * sizeof(F);
    u32 *d_a, *d_b, *d_o;
    HIP_OK(hipMalloc((void**)&d_a, bytes)); HIP_OK(hipMalloc((void**)&d_b, bytes)); HIP_OK(hipMalloc((void**)&d_o, bytes));
    HIP_OK(hipMemcpy(d_a, a, bytes, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_b, b ? b : a, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_field_op<F>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, d_o, d_a, d_b, (unsigned)n, op);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpy(out, d_o, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_o);
}

// field 0 = base field fp, 1 = scalar field fr
SPPARK_FFI RustError sppark_devtest_field_op(int field, int op, void* out, const void* a, const void* b, size_t n)
{
    return guarded([&] {
        if (field == 0) run_field_op<fp_d>(op, out, a, b, n);
        else            run_field_op<fr_d>(op, out, a, b, n);
    });
}

SPPARK_FFI RustError sppark_devtest_xyzz_op(int op, void* out, const void* a, const void* b, size_t n)
{
    return guarded([&] {
        (void)select_gpu(-1);
        size_t ab = n * sizeof(wire_bucket_m), bb = n * (op == 0 || op == 4 ? sizeof(wire_bucket_m) : 8 * fp_d::N);
        wire_bucket_m *d_a, *d_o; unsigned char* d_b;
        HIP_OK(hipMalloc((void**)&d_a, ab)); HIP_OK(hipMalloc((void**)&d_o, ab)); HIP_OK(hipMalloc((void**)&d_b, bb ? bb : 16));
        HIP_OK(hipMemcpy(d_a, a, ab, hipMemcpyHostToDevice));
        if (op != 3 && op != 5) HIP_OK(hipMemcpy(d_b, b, bb, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_xyzz_op, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, d_o, d_a, d_b, (unsigned)n, op);
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpy(out, d_o, ab, hipMemcpyDeviceToHost));
        (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_o);
    });
}

// ---------------------------------------------------------------------------
// The bucket pipeline's own field (ff/montx_dev.hpp on BLS12-381): element-wise operations on
// internal limbs, and point operations computed in the loosely-reduced form but fed and read
// back in the reference's wire form, so that they can be compared bit for bit with the
// 32-bit-limb class and with the oracle.
// ---------------------------------------------------------------------------
template<class F>
__global__ void k_bucket_field_op(u32* out, const u32* a, const u32* b, unsigned n, int op)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if constexpr (field_is_internal<F>::value) {
        constexpr int NL = F::NL;
        F x = F::from_wire(a + (size_t)i * NL), y = F::from_wire(b + (size_t)i * NL), r = F::zero();
        switch (op) {
            case 0: r = x * y; break;                               // x may be fat (limbs < 2^31), y normalised
            case 1: r = x.sqr(); break;                             // x normalised
            case 2: r = F::template sub<3>(x, y).norm(); break;     // y normalised, < 2p
            case 3: r = (x + y).norm(); break;
            case 4: r.l[0] = x.template is_zero_mod<13>(); break;   // x normalised, < 13p
            case 5: r = F::from_std(a + (size_t)i * NL); break;     // first NW words: a wire-form element
            default: { u32 w[NL] = {}; x.to_std(w); for (int j = 0; j < NL; j++) r.l[j] = w[j]; } break;
        }
        r.to_wire(out + (size_t)i * NL);
    }
}

// op 0: a += b (xyzz)   1: a += affine(b)   2: a -= affine(b)   3: a = 2a ; all in wire form
template<class F>
__global__ void k_bucket_xyzz_op(wire_bucket_m* out, const wire_bucket_m* a, const unsigned char* b, unsigned n, int op)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if constexpr (field_is_internal<F>::value) {
        constexpr int NW = fp_d::N;
        auto to_internal = [](const wire_bucket_m* src) {
            xyzz_dev<F> r;
            bool inf = true;
            for (int k = 2 * NW; k < 4 * NW; k++) inf &= src->w[k] == 0;
            if (inf) { r.set_inf(); return r; }
            r.X = F::from_std(src->w); r.Y = F::from_std(src->w + NW);
            r.ZZZ = F::from_std(src->w + 2 * NW); r.ZZ = F::from_std(src->w + 3 * NW);
            return r;
        };
        xyzz_dev<F> p = to_internal(&a[i]);
        if (op == 0) p.add(to_internal(reinterpret_cast<const wire_bucket_m*>(b) + i));
        else if (op == 3) p.dbl();
        else {
            affine_dev<fp_d> qs = load_affine<fp_d, false>(b, i, 8 * NW);
            u32 wx[NW], wy[NW];
            qs.X.to_wire(wx); qs.Y.to_wire(wy);
            affine_dev<F> q; q.X = F::from_std(wx); q.Y = F::from_std(wy); q.inf = qs.inf;
            p.madd(q, op == 2);
        }
        p.store_std(&out[i]);
    }
}

// op 4: a += b (xyzz), op 5: a = 2a -- through the COOPERATIVE forms (ec/xyzz_coop.hpp): four waves per 64 operations
template<class F>
__global__ __launch_bounds__(256) void k_bucket_xyzz_coop(wire_bucket_m* out, const wire_bucket_m* a, const wire_bucket_m* b, unsigned n, int op)
{
    if constexpr (field_is_montx<F>::value) {
        __shared__ coop_lds<F> ex;
        constexpr int NW = fp_d::N;
        const unsigned lane = threadIdx.x & 63, role = threadIdx.x >> 6, i = blockIdx.x * 64 + lane;
        auto to_internal = [](const wire_bucket_m* src) {
            xyzz_dev<F> r;
            bool inf = true;
            for (int k = 2 * NW; k < 4 * NW; k++) inf &= src->w[k] == 0;
            if (inf) { r.set_inf(); return r; }
            r.X = F::from_std(src->w); r.Y = F::from_std(src->w + NW);
            r.ZZZ = F::from_std(src->w + 2 * NW); r.ZZ = F::from_std(src->w + 3 * NW);
            return r;
        };
        xyzz_dev<F> p, q;
        p.set_inf(); q.set_inf();
        if (i < n) { p = to_internal(&a[i]); if (op == 4) q = to_internal(&b[i]); }
        coop_ctx<F> c{&ex, role, lane, 0};
        // two operations in a row, so that the hand-over of the exchange sets between operations is exercised too:
        // op 4: (a + b) + b - checked against two serial additions; op 5: 2(2a)
        if (op == 4) { coop_add<F>(p, q, c); coop_add<F>(p, q, c); }
        else         { coop_dbl<F>(p, c); coop_dbl<F>(p, c); }
        if (i < n && role == (i & 3)) p.store_std(&out[i]);     // every copy is the same: let the four waves take turns
    }
}

// Latency micro-benchmark of a CHAIN of point operations, one work-group per CU (the regime of the MSM's tail):
// mode 0: add_pairs by one wave, 1: coop_add by four waves, 2: dbl_pairs, 3: coop_dbl, 4: serial add (single chains)
template<class F>
__global__ __launch_bounds__(256) void k_chain_bench(wire_bucket_m* out, const wire_bucket_m* a, const wire_bucket_m* b, int mode, int reps)
{
    if constexpr (field_is_montx<F>::value) {
        __shared__ coop_lds<F> ex;
        constexpr int NW = fp_d::N;
        const unsigned lane = threadIdx.x & 63, role = threadIdx.x >> 6;
        auto to_internal = [](const wire_bucket_m* src) {
            xyzz_dev<F> r;
            r.X = F::from_std(src->w); r.Y = F::from_std(src->w + NW);
            r.ZZZ = F::from_std(src->w + 2 * NW); r.ZZ = F::from_std(src->w + 3 * NW);
            return r;
        };
        xyzz_dev<F> p = to_internal(&a[lane]), q = to_internal(&b[lane]);
        coop_ctx<F> c{&ex, role, lane, 0};
        const bool coop = mode == 1 || mode == 3;
        if (!coop && role != 0) return;
        #pragma unroll 1
        for (int r = 0; r < reps; r++) {
            if (mode == 0) p.add_pairs(q);
            else if (mode == 1) coop_add<F>(p, q, c);
            else if (mode == 2) p.dbl_pairs();
            else if (mode == 3) coop_dbl<F>(p, c);
            else p.add(q);
        }
        if (role == 0 && blockIdx.x == 0) p.store_std(&out[lane]);
    }
}
SPPARK_FFI RustError sppark_devtest_chain_bench(int mode, int reps, int nblocks, void* out, const void* a, const void* b, float* ms)
{
    return guarded([&] {
        if (!field_is_montx<msm_fp_d>::value) HIP_OK(hipErrorNotSupported);
        (void)select_gpu(-1);
        const size_t ab = 64 * sizeof(wire_bucket_m);
        wire_bucket_m *d_a, *d_b, *d_o;
        HIP_OK(hipMalloc((void**)&d_a, ab)); HIP_OK(hipMalloc((void**)&d_b, ab)); HIP_OK(hipMalloc((void**)&d_o, ab));
        HIP_OK(hipMemcpy(d_a, a, ab, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_b, b, ab, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
        for (int it = 0; it < 2; it++) {
            HIP_OK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_chain_bench<msm_fp_d>, dim3(nblocks), dim3(256), 0, 0, d_o, d_a, d_b, mode, reps);
            HIP_OK(hipEventRecord(e1, 0));
            HIP_OK(hipEventSynchronize(e1));
        }
        HIP_OK(hipEventElapsedTime(ms, e0, e1));
        HIP_OK(hipMemcpy(out, d_o, ab, hipMemcpyDeviceToHost));
        (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_o); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    });
}

// element size: NL words for ops on internal limbs (see k_bucket_field_op)
SPPARK_FFI RustError sppark_devtest_bucket_field_op(int op, void* out, const void* a, const void* b, size_t n)
{
    return guarded([&] {
        if (!field_is_internal<msm_fp_d>::value) HIP_OK(hipErrorNotSupported);
        (void)select_gpu(-1);
        size_t bytes = n * msm_fp_d::N * 4;
        u32 *d_a, *d_b, *d_o;
        HIP_OK(hipMalloc((void**)&d_a, bytes)); HIP_OK(hipMalloc((void**)&d_b, bytes)); HIP_OK(hipMalloc((void**)&d_o, bytes));
        HIP_OK(hipMemcpy(d_a, a, bytes, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_b, b ? b : a, bytes, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_bucket_field_op<msm_fp_d>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, d_o, d_a, d_b, (unsigned)n, op);
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpy(out, d_o, bytes, hipMemcpyDeviceToHost));
        (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_o);
    });
}
SPPARK_FFI int sppark_devtest_bucket_field_limbs(void)
{   return field_is_internal<msm_fp_d>::value ? (int)msm_fp_d::N : 0;   }

SPPARK_FFI RustError sppark_devtest_bucket_xyzz_op(int op, void* out, const void* a, const void* b, size_t n)
{
    return guarded([&] {
        if (!field_is_internal<msm_fp_d>::value) HIP_OK(hipErrorNotSupported);
        (void)select_gpu(-1);
        size_t ab = n * sizeof(wire_bucket_m), bb = n * (op == 0 || op == 4 ? sizeof(wire_bucket_m) : 8 * fp_d::N);
        wire_bucket_m *d_a, *d_o; unsigned char* d_b;
        HIP_OK(hipMalloc((void**)&d_a, ab)); HIP_OK(hipMalloc((void**)&d_o, ab)); HIP_OK(hipMalloc((void**)&d_b, bb ? bb : 16));
        HIP_OK(hipMemcpy(d_a, a, ab, hipMemcpyHostToDevice));
        if (op != 3 && op != 5) HIP_OK(hipMemcpy(d_b, b, bb, hipMemcpyHostToDevice));
        if (op >= 4) {
            if (!field_is_montx<msm_fp_d>::value) HIP_OK(hipErrorNotSupported);
            hipLaunchKernelGGL(k_bucket_xyzz_coop<msm_fp_d>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, 0, d_o, d_a, (const wire_bucket_m*)d_b, (unsigned)n, op);
        } else
            hipLaunchKernelGGL(k_bucket_xyzz_op<msm_fp_d>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, d_o, d_a, d_b, (unsigned)n, op);
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpy(out, d_o, ab, hipMemcpyDeviceToHost));
        (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_o);
    });
}
