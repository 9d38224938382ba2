"""Summarise rocprofv3 rocpd databases (bench_results.db) into a small text file:
per-kernel calls / total / average duration, and per-kernel averaged PMC values.

    python tools/rocprof_summary.py <stats.db> [<pmc.db> ...] > profiles/<name>.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)                # drop the argument list
    name = re.sub(r"<.*", "", name) if name.startswith("void at::") else name
    return name.replace("void ", "").replace("sppark_amd::", "")[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    print("# rocprofv3 --kernel-trace --stats : %s" % sys.argv[1])
    print("%-72s %6s %14s %14s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for name, calls, total, avg, pct in db.execute("select * from top_kernels"):
        print("%-72s %6d %14.3f %14.3f %6.2f%%" % (short(name), calls, total / 1e3, avg / 1e3, pct))
    for path in sys.argv[2:]:
        pdb = sqlite3.connect(path)
        print("\n# rocprofv3 --pmc : %s   (value averaged over dispatches; FETCH_SIZE/WRITE_SIZE in KiB as reported)" % path)
        print("%-72s %-12s %6s %16s %14s" % ("kernel", "counter", "calls", "avg_value", "avg_us"))
        q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
             "group by kernel_name, counter_name order by sum(duration) desc")
        for name, ctr, calls, val, dur in pdb.execute(q):
            print("%-72s %-12s %6d %16.1f %14.1f" % (short(name), ctr, calls, val, dur / 1e3))


if __name__ == "__main__":
    main()
