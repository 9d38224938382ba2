"""List kernel dispatches (in order) with durations from a rocprofv3 rocpd database."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, start, end, grid_size_x, grid_size_y, workgroup_size_x, lds_block_size from kernels order by start").fetchall() \
    if False else None
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print(cols)
q = "select * from kernels order by start"
for r in db.execute(q):
    d = dict(zip(cols, r))
    name = re.sub(r"\(.*", "", d.get("name", d.get("kernel_name", "?"))).replace("void ", "").replace("sppark_amd::", "")[:40]
    print("%-40s %10.3f ms grid %s wg %s lds %s" % (name, (d["end"] - d["start"]) / 1e6,
          (d.get("grid_x"), d.get("grid_y")), d.get("workgroup_x"), d.get("lds_size", d.get("lds_block_size"))))
