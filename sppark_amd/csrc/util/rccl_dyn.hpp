// RCCL for the one-process-per-GPU exchange step, bound at run time.
//
// The product libraries do not LINK librccl: a prover that drives the node from one process (sppark_msm_multi*) never
// needs it, and a process that already holds a copy -- PyTorch ships its own beside the one under /opt/rocm -- must go
// on using that one, because a communicator is only valid inside the library instance that made it.  So the caller
// creates the communicator (ncclCommInitRank in ITS copy), and the three entry points used here are looked up in the
// library SPPARK_RCCL_LIB names when set, else in the copy that is already mapped (RTLD_NOLOAD), else in librccl.so.1.
// The few types and values of the NCCL/RCCL ABI that these three calls need are declared HERE (they have been stable
// since NCCL 2.0: ncclSuccess = 0, ncclUint8 = 1, an opaque communicator handle), so the libraries build on a ROCm
// install without the rccl development headers; tests/test_abi.py checks them against <rccl/rccl.h> where that exists.
#pragma once
#include "runtime.hpp"
#include <dlfcn.h>
#include <cerrno>

namespace sppark_amd {

typedef struct ncclComm* ncclComm_t;                // opaque (rccl.h: typedef struct ncclComm* ncclComm_t)
typedef int ncclResult_t;                           // enum; ncclSuccess = 0
typedef int ncclDataType_t;                         // enum; ncclUint8 = 1 (ncclInt8 = 0)
static constexpr ncclResult_t ncclSuccess = 0;
static constexpr ncclDataType_t ncclUint8 = 1;

struct rccl_dyn {
    ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_count)(const ncclComm_t, int*) = nullptr;
    const char*  (*error_string)(ncclResult_t) = nullptr;
    std::string origin;

    static const rccl_dyn& get()
    {
        static const rccl_dyn api;
        if (!api.all_gather)
            throw hip_error(ENOSYS, "RCCL is not available to this process: " + api.origin);
        return api;
    }
    static void ok(ncclResult_t r, const char* what)
    {
        if (r != ncclSuccess)
            throw hip_error(EIO, std::string(what) + " failed: " + get().error_string(r));
    }
private:
    rccl_dyn()
    {
        // an explicit choice wins; else a copy that is already in the process; else the system's
        const char* env = getenv("SPPARK_RCCL_LIB");
        const char* names[2] = {"librccl.so.1", "librccl.so"};
        void* h = nullptr;
        if (env && *env) { h = dlopen(env, RTLD_NOW | RTLD_LOCAL); origin = env; }
        else
            for (int pass = 0; pass < 2 && !h; pass++)
                for (const char* n : names) {
                    h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                    if (h) { origin = n; break; }
                }
        if (!h) { const char* e = dlerror(); origin = e ? e : "librccl.so.1 not found"; return; }
        all_gather   = reinterpret_cast<decltype(all_gather)>(dlsym(h, "ncclAllGather"));
        comm_count   = reinterpret_cast<decltype(comm_count)>(dlsym(h, "ncclCommCount"));
        error_string = reinterpret_cast<decltype(error_string)>(dlsym(h, "ncclGetErrorString"));
        if (!all_gather || !comm_count || !error_string) { all_gather = nullptr; origin += ": ncclAllGather / ncclCommCount / ncclGetErrorString missing"; }
    }
};

// all ranks of |comm| contribute |bytes| of host memory; returns the nranks * bytes of all of them in rank order.
// ONE collective on |stream|; the device staging comes from the scratch pool.
inline std::vector<unsigned char> rccl_all_gather_host(const void* mine, size_t bytes, ncclComm_t comm, hipStream_t stream)
{
    const rccl_dyn& api = rccl_dyn::get();
    int nranks = 0;
    rccl_dyn::ok(api.comm_count(comm, &nranks), "ncclCommCount");
    if (nranks < 1) throw hip_error(EINVAL, "communicator without ranks");
    const size_t slot = (bytes + 15) & ~(size_t)15;
    pooled_scratch buf(slot * (1 + (size_t)nranks));
    unsigned char* d_mine = (unsigned char*)buf.p, *d_all = d_mine + slot;
    std::vector<unsigned char> padded(slot, 0), all(slot * nranks);
    memcpy(padded.data(), mine, bytes);
    HIP_OK(hipMemcpyAsync(d_mine, padded.data(), slot, hipMemcpyHostToDevice, stream));
    rccl_dyn::ok(api.all_gather(d_mine, d_all, slot, ncclUint8, comm, stream), "ncclAllGather");
    HIP_OK(hipMemcpyAsync(all.data(), d_all, slot * nranks, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    buf.done();
    std::vector<unsigned char> out(bytes * nranks);
    for (int r = 0; r < nranks; r++) memcpy(out.data() + (size_t)r * bytes, all.data() + (size_t)r * slot, bytes);
    return out;
}

} // namespace sppark_amd
