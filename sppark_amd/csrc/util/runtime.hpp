// HIP runtime glue for the C-ABI libraries: error type, device registry, and
// the by-value error struct of the reference's FFI.
//
//   RustError            <- util/rusterror.h:18-36   {int code; char* message}
//                           returned BY VALUE; code 0 = success, otherwise the
//                           negated runtime error code (util/exception.cuh:19);
//                           message strdup'ed (freed by the caller: Rust `free`,
//                           Go `drop_error_message`).
//   hip_error / HIP_OK   <- util/exception.cuh:11-21  "expr@file:line failed: msg"
//   gpus_t / select_gpu  <- util/all_gpus.cpp:11-63   filtered device list,
//                           id = -1 means "current device"
//   gpu_ptr_inner        <- util/gpu_t.cuh:269-318    ref-counted device pointer
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <atomic>
#include <exception>
#include <cstring>
#include <cstdlib>
#include <mutex>

namespace sppark_amd {

struct RustError {
    int code;
    char* message;
};

class hip_error : public std::exception {
    int _code;
    std::string _what;
public:
    hip_error(int err, const std::string& w) : _code(err), _what(w) {}
    const char* what() const noexcept override { return _what.c_str(); }
    int code() const { return _code; }
};

#define SPPARK_STR2(x) #x
#define SPPARK_STR(x) SPPARK_STR2(x)
#define HIP_OK(expr) do {                                                    \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess)                                                    \
        throw sppark_amd::hip_error(-(int)_e, std::string(#expr "@" __FILE__ ":" SPPARK_STR(__LINE__) " failed: ") \
                                    + hipGetErrorString(_e));                \
} while (0)

static inline RustError rust_ok() { return RustError{0, nullptr}; }
static inline RustError rust_err(int code, const char* what)
{   return RustError{code, what && *what ? strdup(what) : nullptr};   }

struct gpu_info {
    int gid;            // index in the filtered list
    int hip_id;         // HIP device ordinal
    hipDeviceProp_t prop;
};

class gpus_t {
    std::vector<gpu_info> gpus;
    gpus_t()
    {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) return;
        for (int id = 0; id < n; id++) {
            hipDeviceProp_t prop;
            // CDNA parts report major == 9 (gfx9xx); same filter as
            // util/all_gpus.cpp:3-8,19-24 minus cooperativeLaunch, which this
            // implementation never uses.
            if (hipGetDeviceProperties(&prop, id) == hipSuccess && prop.major >= 9)
                gpus.push_back(gpu_info{(int)gpus.size(), id, prop});
        }
    }
public:
    static const std::vector<gpu_info>& all()
    {   static gpus_t g; return g.gpus;   }
};

static inline const gpu_info& select_gpu(int id)
{
    auto& gpus = gpus_t::all();
    if (gpus.empty()) HIP_OK(hipErrorNoDevice);
    if (id == -1) {                     // the calling thread's current device; never switched silently:
        int cur;                        // the caller's device pointers live there
        HIP_OK(hipGetDevice(&cur));
        for (auto& g : gpus) if (g.hip_id == cur) return g;
        HIP_OK(hipErrorInvalidDevice);  // current device is not in the filtered list
    }
    if (id < 0 || (size_t)id >= gpus.size()) HIP_OK(hipErrorInvalidDevice);
    HIP_OK(hipSetDevice(gpus[id].hip_id));
    return gpus[id];
}

struct gpu_ptr_inner {
    void* ptr;
    std::atomic<size_t> ref_cnt;
    int real_id;
};

static inline bool is_device_pointer(const void* p)
{
    if (p == nullptr) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// Device scratch of the entry points that stage or need temporaries (compute_ntt / sppark_lde on host buffers, the
// LDE's coefficient copy, the polynomial scans), kept between calls: a hipMalloc + hipFree pair per call -- the free
// synchronises the whole device -- costs more than a 2^22 -> 2^24 Goldilocks extension itself (0.50 -> 0.33 ms with the
// pool, profiles/r03_ntt_lde.log).  A buffer is taken for ONE call and goes back only after that call has synchronised
// its stream (done()), so concurrent calls never share one; a call that ends in an exception frees its buffer instead.
// Bounded: at most four idle buffers AND at most cache_limit() idle bytes per library (SPPARK_SCRATCH_CACHE_BYTES,
// default 1 GiB; a buffer above the limit is freed when it comes back), and a failed allocation -- here or in the MSM
// contexts of the same library -- frees every idle buffer and tries once more, so the cache can never be the reason a
// call that fits the device runs out of memory.  Every .so has its own pool (one per field / curve).
struct dev_scratch_pool {
    struct item { int dev; void* p; size_t bytes; };
    std::mutex m;
    std::vector<item> idle;
    static dev_scratch_pool& instance() { static dev_scratch_pool pool; return pool; }
    static size_t cache_limit()
    {
        static const size_t lim = [] {
            const char* e = getenv("SPPARK_SCRATCH_CACHE_BYTES");
            return e ? (size_t)strtoull(e, nullptr, 0) : (size_t)1 << 30;
        }();
        return lim;
    }
    // hipMalloc that gives the idle buffers back to the device before it reports out-of-memory
    static hipError_t malloc_or_drain(void** p, size_t bytes)
    {
        hipError_t e = hipMalloc(p, bytes);
        if (e != hipErrorOutOfMemory) return e;
        (void)hipGetLastError();
        instance().release();
        return hipMalloc(p, bytes);
    }
    void* take(int dev, size_t bytes, size_t& got)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            int best = -1;
            for (int i = 0; i < (int)idle.size(); i++)
                if (idle[i].dev == dev && idle[i].bytes >= bytes && (best < 0 || idle[i].bytes < idle[best].bytes)) best = i;
            if (best >= 0) { item it = idle[best]; idle.erase(idle.begin() + best); got = it.bytes; return it.p; }
        }
        void* p = nullptr;
        HIP_OK(malloc_or_drain(&p, bytes ? bytes : 16));
        got = bytes ? bytes : 16;
        return p;
    }
    void give(int dev, void* p, size_t bytes)
    {
        std::vector<item> drop;
        {
            std::lock_guard<std::mutex> lk(m);
            idle.push_back(item{dev, p, bytes});
            // over the count (more than four idle buffers): the smallest goes; over the byte limit: the largest
            for (;;) {
                size_t total = 0;
                for (auto& it : idle) total += it.bytes;
                if (idle.size() <= 4 && total <= cache_limit()) break;
                int victim = 0;
                if (idle.size() <= 4) { for (int i = 1; i < (int)idle.size(); i++) if (idle[i].bytes > idle[victim].bytes) victim = i; }   // over the byte limit: the largest
                else                  { for (int i = 1; i < (int)idle.size(); i++) if (idle[i].bytes < idle[victim].bytes) victim = i; }   // over the count: the smallest
                drop.push_back(idle[victim]); idle.erase(idle.begin() + victim);
            }
        }
        if (drop.empty()) return;
        int cur = 0; (void)hipGetDevice(&cur);
        for (auto& it : drop) { (void)hipSetDevice(it.dev); (void)hipFree(it.p); }
        (void)hipSetDevice(cur);
    }
    size_t idle_bytes()
    {
        std::lock_guard<std::mutex> lk(m);
        size_t total = 0;
        for (auto& it : idle) total += it.bytes;
        return total;
    }
    void release()
    {
        std::vector<item> all;
        { std::lock_guard<std::mutex> lk(m); all.swap(idle); }
        if (all.empty()) return;
        int cur = 0; (void)hipGetDevice(&cur);
        for (auto& it : all) { (void)hipSetDevice(it.dev); (void)hipFree(it.p); }
        (void)hipSetDevice(cur);
    }
};
// one buffer of the pool on the CURRENT device for the duration of a call
struct pooled_scratch {
    void* p = nullptr; size_t bytes = 0; int dev = 0; bool ok = false;
    explicit pooled_scratch(size_t want) { HIP_OK(hipGetDevice(&dev)); p = dev_scratch_pool::instance().take(dev, want, bytes); }
    void done() { ok = true; }                  // the work that used the buffer has been synchronised
    ~pooled_scratch() { if (!p) return; if (ok) dev_scratch_pool::instance().give(dev, p, bytes); else (void)hipFree(p); }
    pooled_scratch(const pooled_scratch&) = delete;
};

} // namespace sppark_amd
