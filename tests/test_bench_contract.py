"""CPU: the bench line recorded on the GPU box (profiles/r03_bench_line.json, the raw last line of
`python bench.py`) has every field of the driver's contract, with consistent values."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_recorded_bench_line_schema():
    line = open(os.path.join(ROOT, "profiles", "r03_bench_line.json")).read().strip()
    assert "\n" not in line                                     # ONE line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "points/s" and d["higher_is_better"] is True and d["scaling"] in ("strong", "weak")
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert "model" not in d["config"]
    # value = points of the whole job / wall time of the timed steps
    n = d["config"]["points_total"]
    assert n == d["config"]["points_per_gpu"] * d["n_gpus"] == 1 << 26            # BASELINE.json: 2^26
    assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["parity"]["timed_msm_equals_oracle"] is True       # the timed result is asserted against the oracle
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes (128 B per point) / the kernel's measured duration per step
    assert abs(r["achieved"] - 128 * d["config"]["points_per_gpu"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6
    assert r["kernel_ms"] <= d["ms_per_step"] and r["launches_per_step"] >= 1
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_gb_per_step"]
    a = d["roofline_alu"]
    assert 0 < a["frac"] < 1 and abs(a["frac"] - a["achieved"] / a["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "cpu_model"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["parity_with_gpu_on_sample"] is True
    assert d["ntt"]["roofline"]["bound"] == "hbm" and 0 < d["ntt"]["roofline"]["frac"] < 1
    assert d["ntt"]["equals_oracle"] is True
    # round 3: the per-GPU shard sizes of the headline MSM on this one GPU, each asserted against the oracle
    sh = d["extras"]["shard_sizes"]
    for lg in (25, 24, 23, 20, 16):
        e = sh["msm_ms_at_2^%d" % lg]
        assert e["equals_oracle"] is True and e["ms"] > 0
    assert sh["msm_ms_at_2^25"]["ms"] < d["ms_per_step"] < 2.2 * sh["msm_ms_at_2^25"]["ms"]


def test_tools_and_job_scripts_parse():
    """tools/*.py byte-compile and tools/jobs/*.sh pass `bash -n`: the measurement helpers named in profiles/ and DESIGN.md
    are at least syntactically whole (they run on the GPU box only)."""
    import glob
    import py_compile
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pys = sorted(glob.glob(os.path.join(root, "tools", "*.py")))
    assert len(pys) >= 20
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        for p in pys:
            py_compile.compile(p, doraise=True, cfile=os.path.join(tmp, os.path.basename(p) + "c"))
    shs = sorted(glob.glob(os.path.join(root, "tools", "jobs", "*.sh")) + glob.glob(os.path.join(root, "tools", "*.sh")))
    assert shs
    for p in shs:
        r = subprocess.run(["bash", "-n", p], capture_output=True, text=True)
        assert r.returncode == 0, (p, r.stderr)
