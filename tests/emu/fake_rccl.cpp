// TEST DOUBLE (tests only): the three RCCL entry points csrc/util/rccl_dyn.hpp binds, for N "ranks" that are THREADS of one
// process sharing one device -- RCCL itself refuses two ranks on one GPU, and the pool has single-GPU boxes only, so this
// is how the N > 1 shape of the native exchange (sppark_msm_rccl: slot layout, rank order, a rank that failed locally)
// is exercised on hardware (tests/test_msm_gpu.py::test_msm_rccl_exchange_many_ranks_with_a_test_double).  Selected with
// SPPARK_RCCL_LIB.  An all-gather here is: wait until every rank has arrived with its send buffer, copy the N pieces
// device-to-device into this rank's receive buffer, wait until every rank has copied.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <condition_variable>
#include <mutex>
#include <vector>

namespace {
struct group {
    int nranks;
    std::mutex m;
    std::condition_variable cv;
    std::vector<const void*> send;
    int arrived = 0, generation = 0;
    explicit group(int n) : nranks(n), send(n, nullptr) {}
    void barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        const int gen = generation;
        if (++arrived == nranks) { arrived = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};
struct fake_comm { group* g; int rank; };
}

extern "C" {
// nranks communicators of one group: out[i] = the handle of rank i (opaque; passed where an ncclComm_t is expected)
__attribute__((visibility("default"))) void fake_rccl_make(int nranks, void** out)
{
    group* g = new group(nranks);
    for (int i = 0; i < nranks; i++) out[i] = new fake_comm{g, i};
}
__attribute__((visibility("default"))) ncclResult_t ncclCommCount(const ncclComm_t comm, int* count)
{
    *count = reinterpret_cast<const fake_comm*>(comm)->g->nranks;
    return ncclSuccess;
}
__attribute__((visibility("default"))) const char* ncclGetErrorString(ncclResult_t) { return "test double"; }
__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount,
                                                                  ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
    if (datatype != ncclUint8) return ncclInvalidArgument;
    fake_comm* c = reinterpret_cast<fake_comm*>(comm);
    group& g = *c->g;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;      // this rank's send buffer is written
    { std::lock_guard<std::mutex> lk(g.m); g.send[c->rank] = sendbuff; }
    g.barrier();
    for (int r = 0; r < g.nranks; r++)
        if (hipMemcpyAsync((char*)recvbuff + (size_t)r * sendcount, g.send[r], sendcount, hipMemcpyDeviceToDevice, stream) != hipSuccess)
            return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    g.barrier();                                                                        // nobody's send buffer is reused before all have read it
    return ncclSuccess;
}
}
