// Symbols every sppark_amd library exports, mirroring util/all_gpus.cpp:65-86
// and util/gpu_t.cuh:365-369 (SPPARK_FFI = extern "C" + default visibility).
#pragma once
#include "../util/runtime.hpp"

#define SPPARK_FFI extern "C" __attribute__((visibility("default")))

// true iff at least one usable device (util/all_gpus.cpp:65-66)
SPPARK_FFI bool cuda_available()
{   return !sppark_amd::gpus_t::all().empty();   }

// gpu_ptr_t<void> is ONE pointer to {T* ptr; atomic<size_t> ref_cnt; int real_id}
// (util/gpu_t.cuh:269-318); the last drop frees the device memory on the
// owning device.
SPPARK_FFI void drop_gpu_ptr_t(void** ref)
{
    auto* in = reinterpret_cast<sppark_amd::gpu_ptr_inner*>(*ref);
    if (in && in->ref_cnt.fetch_sub(1, std::memory_order_seq_cst) == 1) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != in->real_id) (void)hipSetDevice(in->real_id);
        (void)hipFree(in->ptr);
        if (cur != in->real_id) (void)hipSetDevice(cur);
        delete in;
    }
    *ref = nullptr;
}

SPPARK_FFI void* clone_gpu_ptr_t(void* const* ref)
{
    auto* in = reinterpret_cast<sppark_amd::gpu_ptr_inner*>(*ref);
    if (in) in->ref_cnt.fetch_add(1, std::memory_order_relaxed);
    return in;
}

// extension: wrap a fresh device allocation in a gpu_ptr_t (the reference
// creates these from C++ only, e.g. gpu_ptr_t<T>{(T*)gpu.Dmalloc(..)}).
SPPARK_FFI void* sppark_gpu_ptr_alloc(size_t bytes)
{
    void* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    auto* in = new sppark_amd::gpu_ptr_inner{d, {1}, 0};
    (void)hipGetDevice(&in->real_id);
    return in;
}
SPPARK_FFI void* sppark_gpu_ptr_get(void* const* ref)
{   auto* in = reinterpret_cast<sppark_amd::gpu_ptr_inner*>(*ref); return in ? in->ptr : nullptr;   }

// TAKE_RESPONSIBILITY_FOR_ERROR_MESSAGE is always on here, as in both of the
// reference's build drivers (rust/src/build.rs:9, go/sppark.go:296).
SPPARK_FFI void drop_error_message(char* ptr)
{   free(ptr);   }

// Go bridge smoke test symbol (poc/go/poc.cu:17-32, called by poc/go/poc.go:3-22):
// launches a trivial kernel and returns {code, strdup(message)} -- the message
// is always set, as in the reference.
__global__ void sppark_hello_kernel(int* flag) { if (flag) *flag = 1; }
SPPARK_FFI sppark_amd::RustError cuda_func(void* ptr)
{
    (void)ptr;
    hipLaunchKernelGGL(sppark_hello_kernel, dim3(1), dim3(1), 0, 0, (int*)nullptr);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipDeviceSynchronize();
    return sppark_amd::RustError{(int)err, strdup(hipGetErrorString(err))};
}
