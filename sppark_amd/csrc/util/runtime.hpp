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

} // namespace sppark_amd
