"""per-kernel time per training step from a rocprofv3 --kernel-trace database (rocpd sqlite) of `bench.py`

    python tools/prof_summary.py <results.db> <steps incl. warm-up> [out.txt]
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
nsteps = int(sys.argv[2])
rows = list(db.execute("select name, count(*), sum(end-start) from kernels group by name order by sum(end-start) desc"))
tot = sum(r[2] for r in rows)
lines = ["total kernel time %.2f ms/step over %d steps (%d kernels)" % (tot / 1e6 / nsteps, nsteps, len(rows)),
         "%-78s %9s %10s %9s" % ("kernel", "calls/step", "ms/step", "avg us")]
small = 0
for n, c, t in rows:
    nm = re.sub(r"\(.*", "", n).replace("void ", "")[:78]
    lines.append("%-78s %9.1f %10.3f %9.1f" % (nm, c / nsteps, t / 1e6 / nsteps, t / 1e3 / c))
    if t / 1e3 / c < 15.0:
        small += c / nsteps
lines.append("launches shorter than 15 us: %.0f per step" % small)
out = "\n".join(lines)
print(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(out + "\n")
