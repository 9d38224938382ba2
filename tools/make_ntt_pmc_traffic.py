"""profiles/rNN_ntt_gl64_pmc.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only) of
`python tools/gpu_ntt_one.py gl64 <lg> <reps>`: bytes per TRANSFORM = the sum over the kernels of one forward NR transform
(k_ntt6 x2 + k_ntt12 at 2^24), FETCH x2 as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950.
bench.py reads the file for ntt.roofline.traffic.
    python tools/make_ntt_pmc_traffic.py <fetch.db> <write.db> <lg> <reps> > profiles/r05_ntt_gl64_pmc.json"""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    q = ("select kernel_name, count(*), avg(v), avg(d) from (select kernel_name, dispatch_id, sum(value) as v, max(duration) as d "
         "from counters_collection where counter_name = ? group by kernel_name, dispatch_id) group by kernel_name")
    return {re.sub(r"\(.*", "", n).replace("void ", "").replace("sppark_amd::", ""): (c, v, d / 1e3) for n, c, v, d in db.execute(q, (counter,))}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
lg, reps = int(sys.argv[3]), int(sys.argv[4])
n = 1 << lg
kernels, f_bytes, w_bytes, us = {}, 0.0, 0.0, 0.0
for name, (calls, kib, dur) in sorted(fetch.items()):
    if not name.startswith("k_ntt") or calls % reps:
        continue
    per = calls // reps                                     # launches of this kernel per transform
    wk = write.get(name, (0, 0.0, 0.0))[1]
    kernels[name] = {"launches_per_transform": per, "FETCH_SIZE_KiB_per_launch": kib, "WRITE_SIZE_KiB_per_launch": wk, "avg_us_under_pmc": dur}
    f_bytes += per * kib * 1024 * 2.0; w_bytes += per * wk * 1024; us += per * dur
print(json.dumps({
    "workload": "python tools/gpu_ntt_one.py gl64 %d %d (Goldilocks forward NR, device-resident, non-null stream)" % (lg, reps),
    "field": "gl64", "lg": lg, "kernels": kernels,
    "fetch_correction": 2.0, "write_correction": 1.0,
    "fetch_bytes": f_bytes, "write_bytes": w_bytes, "algorithmic_bytes": 16 * n,
    "traffic_over_algorithmic": (f_bytes + w_bytes) / (16.0 * n),
    "sum_of_kernel_us_under_pmc": us,
    "collection": "rocprofv3 --pmc FETCH_SIZE and rocprofv3 --pmc WRITE_SIZE in separate passes, no trace domains",
    "correction": "MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads (x2); WRITE_SIZE x1",
}, indent=1))
