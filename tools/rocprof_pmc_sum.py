"""Per-kernel PMC values of a rocprofv3 --pmc database: counter summed over its instances per dispatch,
then averaged over the dispatches of the kernel (plus calls and the average duration).
    python tools/rocprof_pmc_sum.py <pmc.db> [min_avg_us]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1]); min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
q = ("select kernel_name, counter_name, count(*), avg(v), avg(d) from (select kernel_name, counter_name, dispatch_id, "
     "sum(value) as v, max(duration) as d from counters_collection group by kernel_name, counter_name, dispatch_id) "
     "group by kernel_name, counter_name order by sum(d) desc")
print("%-60s %-24s %6s %18s %12s" % ("kernel", "counter", "calls", "avg value/dispatch", "avg us"))
for name, ctr, calls, val, dur in db.execute(q):
    if dur / 1e3 < min_us:
        continue
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("sppark_amd::", "")[:60]
    print("%-60s %-24s %6d %18.1f %12.1f" % (name, ctr, calls, val, dur / 1e3))
