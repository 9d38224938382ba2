"""BASELINE configs[3] (BLS12-381 G1 MSM, 2^28 points over 8 GPUs) rehearsed at FULL SIZE on ONE MI355X.
No multi-GPU box is in this pool, so nothing here is a scaling measurement; what it shows is that every
piece of the 8-rank job executes at its real size and gives the oracle's result:

  a  `torchrun --nproc-per-node 8 bench.py --gpus 8 --backend gloo --total-lg 28`: eight ranks of 2^25
     points share the device (DRY RUN: exchange through host memory), combined result vs the oracle
  b  the same workload through the C ABI in ONE process: sppark_msm_multi_shards_ms with eight
     device-resident shards of 2^25 points, all on device 0, result vs the oracle, the eight out_ms recorded
  c  one plain-path MSM of 2^28 + 3 * 2048 points (not a power of two) on one GPU vs the oracle: the first
     run above 2^26 of every 32-bit index in msm_driver.hpp / msm_kernels.hpp / msm_sort_kernels.hpp

    python tools/gpu_config3_rehearsal.py [a] [b] [c] [--total-lg 28]      -> gpurun_out/r04_config3_rehearsal.json
"""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import numpy as np
import torch

import sppark_amd
from sppark_amd import synth

PERIOD = 2048
legs = [a for a in sys.argv[1:] if a in ("a", "b", "c")] or ["a", "b", "c"]
total_lg = int(sys.argv[sys.argv.index("--total-lg") + 1]) if "--total-lg" in sys.argv else 28
WORLD = 8
OUT = {"what": "BASELINE configs[3] rehearsal on one GPU (DRY RUN: not a scaling measurement)", "total_lg": total_lg,
       "device": torch.cuda.get_device_name(0)}


def big_scalars(n, seed):
    """n uniform scalars in pieces of 2^25 (bounded temporaries), each piece with its own seed"""
    out = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    step = 1 << 25
    for k, lo in enumerate(range(0, n, step)):
        m = min(step, n - lo)
        out[lo:lo + m] = synth.uniform_scalars(m, "bls12_381", seed + 1000 * k)
    return out


def expect_for(base, sc):
    import oracle as O
    from oracle import fold
    return O.msm_affine(O.BLS12_381, base.cpu().numpy(), fold.fold_scalars(sc, PERIOD, O.FR_MODULUS[O.BLS12_381]), algo=0, param=8)


if "a" in legs:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(WORLD),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(WORLD), "--backend", "gloo", "--total-lg", str(total_lg), "--steps", "2", "--warmup", "1",
           "--no-ntt", "--no-extras", "--no-cpu-baseline"]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print("[a] rc", r.returncode, "wall %.1f s" % (time.time() - t0), flush=True)
    if r.returncode != 0 or len(lines) != 1:
        print(r.stderr[-4000:]); OUT["a"] = {"error": r.stderr[-1500:]}
    else:
        d = json.loads(lines[0])
        print("[a]", lines[0], flush=True)
        assert d["config"]["points_per_gpu"] == (1 << total_lg) // WORLD and d["parity"]["timed_msm_equals_oracle"] is True
        assert "DRY RUN" in d["config"]["workload"] and ("configs[3]" in d["config"]["workload"]) == (total_lg == 28)
        OUT["a"] = {"command": " ".join(cmd[1:]).replace(ROOT + "/", ""), "line": d}

if "b" in legs or "c" in legs:
    import oracle as O
    O.build()

if "b" in legs:
    per = (1 << total_lg) // WORLD
    pts, base = synth.replicated_points(per, "bls12_381", PERIOD, 0x5eed5eed0001)        # every shard: whole periods of the same list
    shards, allsc = [], []
    for k in range(WORLD):
        sc = synth.uniform_scalars(per, "bls12_381", 0x5eed5eed0001 + k)
        shards.append((pts, sc)); allsc.append(sc)
    torch.cuda.synchronize()
    res = None
    for it in range(2):                                          # the second call reuses the pooled contexts' scratch
        t0 = time.perf_counter()
        res, ms = sppark_amd.msm_multi_shards(shards, device_ids=[0] * WORLD, timings=True)
        wall = time.perf_counter() - t0
        print("[b] call %d: wall %.1f ms, out_ms %s" % (it, wall * 1e3, ["%.1f" % v for v in ms]), flush=True)
    # oracle: class sums over ALL shards (every shard starts at a period boundary)
    from oracle import fold
    r_mod = O.FR_MODULUS[O.BLS12_381]
    tot = [0] * PERIOD
    for sc in allsc:
        f = fold.fold_scalars(sc, PERIOD, r_mod)
        for j in range(PERIOD):
            tot[j] = (tot[j] + int.from_bytes(f[j].tobytes(), "little")) % r_mod
    folded = np.stack([np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8) for v in tot])
    exp = O.msm_affine(O.BLS12_381, base.cpu().numpy(), folded, algo=0, param=8)
    ok = bool((sppark_amd.to_affine(res) == exp).all())
    print("[b] equals oracle:", ok, flush=True)
    assert ok
    OUT["b"] = {"entry_point": "sppark_msm_multi_shards_ms", "shards": WORLD, "points_per_shard": per, "device_ids": [0] * WORLD,
                "out_ms": ms, "wall_ms": wall * 1e3, "equals_oracle": ok,
                "note": "eight host threads + pooled contexts on ONE device: the out_ms overlap and are not per-GPU times"}
    sppark_amd.ffi.load("bls12_381").sppark_msm_release_cached()
    del shards, allsc, pts
    torch.cuda.empty_cache()

if "c" in legs:
    n = (1 << total_lg) + 3 * PERIOD
    pts, base = synth.replicated_points(n, "bls12_381", PERIOD, 0x5eed5eed0001)
    sc = big_scalars(n, 77)
    ctx = sppark_amd.MsmContext("bls12_381", stream=torch.cuda.current_stream().cuda_stream)
    ctx.enable_timing(True)
    out = ctx.invoke(pts, sc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = ctx.invoke(pts, sc)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    plan = ctx.plan(n)
    ok = bool((sppark_amd.to_affine(out) == expect_for(base, sc)).all())
    print("[c] n = 2^%d + 3*2048 = %d: %.1f ms (accumulate %.1f), chunks %d, scratch %.1f GB, plan %s, equals oracle: %s"
          % (total_lg, n, dt * 1e3, ctx.kernel_ms(1), ctx.last_chunks(), ctx.scratch_bytes() / 1e9, plan, ok), flush=True)
    assert ok
    OUT["c"] = {"points": n, "ms": dt * 1e3, "accumulate_ms": ctx.kernel_ms(1), "points_per_s": n / dt, "chunks": ctx.last_chunks(),
                "scratch_gb": ctx.scratch_bytes() / 1e9, "plan": plan, "equals_oracle": ok}
    ctx.close()

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r04_config3_rehearsal.json"), "w") as f:
    json.dump(OUT, f, indent=1)
print("wrote gpurun_out/r04_config3_rehearsal.json")
