// Host-side cost of the pieces of a small sppark_ntt call (microseconds per call, wall clock, 2000 calls each):
// what the 17-19 us gaps between the kernels of back-to-back small transforms are made of
// (profiles/r04_ntt_small_vs_reference.log).   hipcc -O2 -o host_overhead host_overhead.hip -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <mutex>
#include <map>
#include <tuple>

__global__ void k_empty(int* p) { if (p) *p = 1; }
struct err_t { int code; char* message; };
typedef err_t (*ntt_fn)(size_t, void*, uint32_t, int, int, int, void*);

template<class Fn> static double per_call_us(int n, Fn&& fn)
{
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) fn();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main(int argc, char** argv)
{
    const int N = 2000;
    void* d = nullptr;
    hipMalloc(&d, 1 << 24);
    hipStream_t s; hipStreamCreate(&s);
    hipDeviceSynchronize();
    hipPointerAttribute_t attr;
    printf("hipPointerGetAttributes      %6.2f us\n", per_call_us(N, [&] { hipPointerGetAttributes(&attr, d); }));
    int mt = 0;
    printf("hipPointerGetAttribute(type) %6.2f us\n", per_call_us(N, [&] { hipPointerGetAttribute(&mt, HIP_POINTER_ATTRIBUTE_MEMORY_TYPE, d); }));
    printf("hipSetDevice                 %6.2f us\n", per_call_us(N, [&] { hipSetDevice(0); }));
    int cur;
    printf("hipGetDevice                 %6.2f us\n", per_call_us(N, [&] { hipGetDevice(&cur); }));
    printf("hipGetLastError              %6.2f us\n", per_call_us(N, [&] { (void)hipGetLastError(); }));
    std::mutex m; std::map<std::tuple<int, unsigned, int>, int> mp; mp[{0, 8u, 0}] = 1;
    printf("mutex + map lookup           %6.2f us\n", per_call_us(N, [&] { std::lock_guard<std::mutex> lk(m); (void)mp.find({0, 8u, 0}); }));
    printf("launch empty kernel (stream) %6.2f us\n", per_call_us(N, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (int*)nullptr); }));
    hipStreamSynchronize(s);
    printf("launch empty kernel (null)   %6.2f us\n", per_call_us(N, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, (int*)nullptr); }));
    hipDeviceSynchronize();
    printf("launch + stream sync         %6.2f us\n", per_call_us(N, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (int*)nullptr); hipStreamSynchronize(s); }));
    for (int a = 1; a < argc; a++) {
        void* h = dlopen(argv[a], RTLD_NOW | RTLD_LOCAL);
        if (!h) { printf("%s: %s\n", argv[a], dlerror()); continue; }
        ntt_fn ntt = (ntt_fn)dlsym(h, "sppark_ntt");
        for (unsigned lg : {8u, 12u, 16u}) {
            for (int i = 0; i < 10; i++) ntt(0, d, lg, 1, 0, 0, s);
            hipStreamSynchronize(s);
            double issue = per_call_us(N, [&] { ntt(0, d, lg, 1, 0, 0, s); });
            auto t0 = std::chrono::steady_clock::now();
            hipStreamSynchronize(s);
            double drain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("%s sppark_ntt 2^%-2u issue %6.2f us per call (queue drained %.0f us after the last)\n", argv[a], lg, issue, drain);
        }
    }
    return 0;
}
