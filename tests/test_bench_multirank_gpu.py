"""GPU (one device is enough): bench.py's world_size > 1 control flow end to end -- contiguous shards in whole
point periods, per-rank scalar seeds, the all-gather of the partial sums, the gather of the folded scalars
for the checker, the MAX all-reduce of the elapsed time and the config strings of the three modes -- as a
DRY RUN: `--backend gloo` lets the ranks share the visible GPU and exchange through host memory.  Every run
asserts its combined result against the oracle inside bench.py; here the JSON line is checked."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, extra):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--backend", "gloo", "--steps", "2", "--warmup", "1",
           "--no-ntt", "--no-extras", "--no-cpu-baseline"] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                    # ONE line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("world,extra,total,per_gpu,scaling", [
    (2, ["--lg", "20"], 1 << 20, 1 << 19, "strong"),                       # default mode: the 2^lg MSM cut into N shards
    (2, ["--total-lg", "21"], 1 << 21, 1 << 20, "strong"),                 # configs[3]'s mode
    (2, ["--scaling", "weak", "--lg", "19"], 1 << 20, 1 << 19, "weak"),    # 2^lg per rank
    (3, ["--lg", "18"], 1 << 18, 43 * 2048, "strong"),                     # uneven shards: 43 + 43 + 42 periods
    (8, ["--total-lg", "24"], 1 << 24, 1 << 21, "strong"),                 # configs[3]'s rank count (its full size: tools/gpu_config3_rehearsal.py)
])
def test_bench_multirank_dry_run(libs, world, extra, total, per_gpu, scaling):
    d = _run(world, extra)
    assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == scaling
    assert d["config"]["points_total"] == total and d["config"]["points_per_gpu"] == per_gpu
    assert d["config"]["backend"] == "gloo" and "DRY RUN" in d["config"]["workload"]
    assert d["parity"]["timed_msm_equals_oracle"] is True
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["vs_baseline"] is None and d["unit"] == "points/s"
    assert d["ms_per_step_min"] <= d["ms_per_step_median"] <= d["ms_per_step"] * 2
