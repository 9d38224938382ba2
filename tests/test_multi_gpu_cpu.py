"""CPU, world_size 2 and 3, gloo: the multi-GPU MSM host logic (sharding, byte
all-gather, combine) with the oracle standing in for the per-rank HIP MSM.  Cases: even and
uneven shards, a rank whose partial sum is the point at infinity, a rank with an empty shard."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(n, zero_rank, world):
    import oracle as O
    import recipe
    from sppark_amd import multi_gpu
    pts, sc = recipe.msm_inputs(O.BLS12_381, n, 4242)
    if zero_rank is not None:                       # this rank's partial result is the point at infinity
        lo, hi = multi_gpu.shard_bounds(n, world, zero_rank)
        sc[lo:hi] = 0
    return pts, sc


def _worker(rank, world, port, n, q, zero_rank=None):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import oracle as O
    import recipe
    from sppark_amd import multi_gpu
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pts, sc = _inputs(n, zero_rank, world)
    lo, hi = multi_gpu.shard_bounds(n, world, rank)
    local = lambda p, s: O.msm(O.BLS12_381, p, s, algo=0, param=1)      # stand-in for MsmContext.invoke
    out = multi_gpu.msm_sharded(local, pts[lo:hi], sc[lo:hi])
    q.put((rank, out.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds():
    from sppark_amd import multi_gpu
    for n in (0, 1, 7, 8, 1000):
        for ws in (1, 2, 3, 8):
            spans = [multi_gpu.shard_bounds(n, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


@pytest.mark.parametrize("n,world,zero_rank", [(301, 2, None), (100, 3, 1), (2, 3, None), (64, 3, 0)])
def test_msm_sharded_gloo(oracle, libs, n, world, zero_rank):
    O = oracle
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q, zero_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pts, sc = _inputs(n, zero_rank, world)
    exp = O.msm_affine(O.BLS12_381, pts, sc)
    for r in range(world):
        got = np.frombuffer(res[r], dtype=np.uint8)
        assert (O.jac_to_affine(O.BLS12_381, got) == exp).all()
