// Tool (not product, not test): back-to-back sppark_ntt calls issued from C between two events on the caller's stream, so
// that the small-size comparison with the reference's build (oracle/ref_ntt_shim.cu: ref_ntt_dev_timed, a C loop too)
// does not charge our side a Python call per transform.  hipcc -O2 -fPIC -shared tools/ntt_loop.cpp -o tools/libntt_loop.so
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdlib>

struct RustError { int code; char* message; };
typedef RustError (*sppark_ntt_fn)(size_t, void*, uint32_t, int, int, int, void*);

// *ms = average milliseconds per call between the events; *issue_us = host time per call to enqueue
extern "C" __attribute__((visibility("default")))
int ntt_loop(void* fn, void* d, uint32_t lg, int order, int direction, int type, void* stream, int iters, float* ms, float* issue_us)
{
    sppark_ntt_fn f = (sppark_ntt_fn)fn;
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < 3; i++) { RustError e = f(0, d, lg, order, direction, type, stream); if (e.code) { free(e.message); return e.code; } }
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
    (void)hipEventRecord(a, s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; i++) { RustError e = f(0, d, lg, order, direction, type, stream); if (e.code) { free(e.message); return e.code; } }
    auto t1 = std::chrono::steady_clock::now();
    (void)hipEventRecord(b, s);
    (void)hipEventSynchronize(b);
    float t = 0;
    (void)hipEventElapsedTime(&t, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *ms = t / (iters > 0 ? iters : 1);
    *issue_us = (float)(std::chrono::duration<double, std::micro>(t1 - t0).count() / (iters > 0 ? iters : 1));
    return 0;
}
