"""Timeline of the LAST `count` kernel dispatches of a rocprofv3 database: start offset, duration and
the idle gap before each dispatch (device timestamps)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
rows = rows[-count:]
t0 = rows[0][1]; prev_end = t0
for name, s, e, gx, gy, wx in rows:
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("sppark_amd::", "")[:48]
    print("%-48s at %9.3f ms  dur %8.3f ms  gap %7.3f ms  grid %d x %d / %d" % (name, (s - t0) / 1e6, (e - s) / 1e6, (s - prev_end) / 1e6, gx, gy, wx))
    prev_end = max(prev_end, e)
print("span %.3f ms" % ((prev_end - t0) / 1e6))
