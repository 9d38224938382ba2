"""Multi-GPU MSM: one process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI), contiguous point/scalar shards, ONE exchange step.

The reference has no multi-GPU MSM at all (one gpu_t per msm_t,
msm/pippenger.cuh:329,351-353); this is new design for BASELINE.json config 4.
Sum_i s_i*P_i splits over any partition of the index set, so every rank runs
the single-GPU pipeline on its shard and the partial results (one 144-byte
Jacobian point each) are exchanged with an all-gather of raw bytes -- elliptic
curve addition is not an RCCL reduction operator, so all-reduce does not apply
(SURVEY 8(e)).  The payload is latency-bound (world_size * 144 B); every rank
then adds the world_size points on the host, so all ranks hold the result.
"""
import numpy as np

from . import msm as _msm


def shard_bounds(npoints, world_size, rank):
    """contiguous index range [lo, hi) of |rank| (SURVEY 8(e) partitioning)"""
    base, rem = divmod(npoints, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def combine_partials(partials, curve="bls12_381"):
    """sum of per-rank Jacobian partial results (host arithmetic in the C-ABI library)"""
    return _msm.jacobian_sum(np.ascontiguousarray(partials, dtype=np.uint8), curve)


def all_gather_bytes(local, group=None):
    """all-gather equal-length byte strings; uint8 tensor on the current CUDA
    device when the backend is nccl (RCCL), on the CPU for gloo."""
    import torch
    import torch.distributed as dist
    ws = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.uint8)).to(dev)
    out = torch.empty((ws, t.numel()), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out.view(-1), t, group=group)
    return out.cpu().numpy()


def msm_sharded(local_msm, points, scalars, curve="bls12_381", group=None):
    """local_msm(points, scalars) -> Jacobian bytes of this rank's shard (the HIP
    path: MsmContext.invoke).  Returns the Jacobian result of the whole MSM on
    every rank."""
    import torch.distributed as dist
    part = local_msm(points, scalars)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return part
    return combine_partials(all_gather_bytes(part, group), curve)


# ---- the native exchange: sppark_msm_rccl / sppark_msm_rccl_sum over an RCCL communicator made WITHOUT torch ----
# What a Rust / Go / C prover that runs one process per GPU does (INTEGRATION.md "Multi-GPU"): rank 0 makes an
# ncclUniqueId, hands its 128 bytes to the other ranks by whatever it has (here: the caller's choice -- a file, MPI,
# torch.distributed's store), every rank calls ncclCommInitRank and passes the communicator to the library.
import ctypes
import os


def _rccl():
    # $SPPARK_RCCL_LIB when set, else the copy the process already holds (torch brings one; dlopen of a soname that is
    # mapped returns that mapping), else the ROCm one: the resolution order of csrc/util/rccl_dyn.hpp, so that the
    # communicator is made by the copy that will use it
    lib = ctypes.CDLL(os.environ.get("SPPARK_RCCL_LIB") or "librccl.so.1")
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    return lib


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]          # ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), by value


class RcclComm:
    """An ncclComm_t of this process (one rank; the calling thread's current HIP device)."""

    @staticmethod
    def unique_id():
        """rank 0: the 128 bytes every rank needs"""
        lib, uid = _rccl(), _UniqueId()
        RcclComm._ok(lib, lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        return bytes(bytearray(uid.internal))

    @staticmethod
    def _ok(lib, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, lib.ncclGetErrorString(rc).decode()))

    def __init__(self, nranks, rank, unique_id):
        self.lib = _rccl()
        uid = _UniqueId()
        ctypes.memmove(uid.internal, unique_id, 128)
        self.handle = ctypes.c_void_p()
        self.lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        self._ok(self.lib, self.lib.ncclCommInitRank(ctypes.byref(self.handle), nranks, uid, rank), "ncclCommInitRank")
        self.nranks, self.rank = nranks, rank

    def destroy(self):
        if self.handle:
            self.lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            self.lib.ncclCommDestroy(self.handle)
            self.handle = ctypes.c_void_p()


def rccl_sum(partial, comm, curve="bls12_381", g2=False, stream=None):
    """sppark_msm_rccl_sum: every rank passes its Jacobian partial sum, every rank gets the sum."""
    from . import ffi
    L = ffi.load(curve)
    part = np.ascontiguousarray(partial, dtype=np.uint8)
    out = np.zeros_like(part)
    ffi.check(L, L.sppark_msm_rccl_sum(out.ctypes.data, part.ctypes.data, int(g2), comm.handle, stream))
    return out


def msm_rccl(points, scalars, comm, curve="bls12_381", mont=False, ffi_affine_sz=None, stream=None):
    """sppark_msm_rccl: this rank's shard (host arrays or tensors on the current device) -> the whole MSM on every rank."""
    from . import ffi
    L = ffi.load(curve)
    fb = _msm.FP_BYTES[curve]
    stride = ffi_affine_sz or 2 * fb
    n = _msm._npoints(points, stride) if _msm._nbytes(points) else 0
    if _msm._nbytes(scalars) != 32 * n:
        raise ValueError("length mismatch")
    pp, _k1 = ffi.as_pointer(points)
    sp, _k2 = ffi.as_pointer(scalars)
    out = np.zeros(3 * fb, dtype=np.uint8)
    ffi.check(L, L.sppark_msm_rccl(out.ctypes.data, pp, n, sp, int(mont), stride, comm.handle, stream))
    return out
