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
