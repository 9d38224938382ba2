"""profiles/rNN_pmc_traffic.json from the two rocprofv3 --pmc passes of `python bench.py --steps 2 --warmup 1
--no-cpu-baseline --no-ntt --no-extras`: FETCH_SIZE / WRITE_SIZE of k_accumulate per launch, with the in-run
calibration of the counters on kernels with known byte counts (MI355X_MICROARCH.md, HBM section).
    python tools/make_pmc_traffic.py <fetch.db> <write.db> <lg> > profiles/r04_pmc_traffic.json"""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    q = ("select kernel_name, count(*), avg(v), avg(d) from (select kernel_name, dispatch_id, sum(value) as v, max(duration) as d "
         "from counters_collection where counter_name = ? group by kernel_name, dispatch_id) group by kernel_name")
    return {name: (calls, val, dur / 1e3) for name, calls, val, dur in db.execute(q, (counter,))}


fetch, write, lg = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
n = 1 << lg


def pick(d, needle, biggest=True):
    c = [(v[1], k, v) for k, v in d.items() if needle in k]
    c.sort()
    return c[-1][1:] if c else (None, (0, 0.0, 0.0))


acc_name, (calls, f_kib, f_us) = pick(fetch, "k_accumulate")
_, (_, w_kib, w_us) = pick(write, "k_accumulate")
_, (_, bd_f, _) = pick(fetch, "k_breakdown"); _, (_, bd_w, _) = pick(write, "k_breakdown")
_, (_, cv_f, _) = pick(fetch, "k_convert_points"); _, (_, cv_w, _) = pick(write, "k_convert_points")
nwins = 12 if lg >= 26 else None
out = {
    "kernel": acc_name.split("(")[0].replace("void sppark_amd::", "").replace("sppark_amd::", ""),
    "workload": "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-extras (BLS12-381 G1 MSM 2^%d, one k_accumulate launch per step)" % lg,
    "lg": lg, "curve": "bls12_381",
    "FETCH_SIZE_KiB_per_launch": f_kib, "WRITE_SIZE_KiB_per_launch": w_kib,
    "fetch_correction": 2.0, "write_correction": 1.0,
    "fetch_bytes": f_kib * 1024 * 2.0, "write_bytes": w_kib * 1024, "raw_fetch_bytes": f_kib * 1024,
    "avg_kernel_us_under_pmc": f_us, "dispatches_averaged": calls,
    "collection": "rocprofv3 --pmc FETCH_SIZE and rocprofv3 --pmc WRITE_SIZE in separate passes, no trace domains (tools/jobs/r4_evidence.sh)",
    "calibration_in_the_same_passes": {
        "k_breakdown": {"reads_KiB": n * 32 / 1024, "FETCH_SIZE_KiB": bd_f, "ratio": bd_f / (n * 32 / 1024) if bd_f else None,
                        "writes_KiB": (nwins or 0) * n * 4 / 1024, "WRITE_SIZE_KiB": bd_w},
        "k_convert_points": {"reads_KiB": n * 96 / 1024, "FETCH_SIZE_KiB": cv_f, "ratio": cv_f / (n * 96 / 1024) if cv_f else None,
                             "writes_KiB": n * 128 / 1024, "WRITE_SIZE_KiB": cv_w, "write_ratio": cv_w / (n * 128 / 1024) if cv_w else None},
    },
    "correction": "MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads; "
                  "the calibration kernels above confirm x0.50 for reads and x1.0 for writes in these very passes, hence FETCH x2, WRITE x1. "
                  "For the GATHERS of k_accumulate the factor 2 is an upper bound (profiles/r02_pmc_traffic.json).",
}
print(json.dumps(out, indent=1))
