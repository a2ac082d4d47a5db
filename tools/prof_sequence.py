"""the launch sequence of ONE training step from a rocprofv3 --kernel-trace database of bench.py: kernel, duration and the
gap to the previous kernel's end -- to see what surrounds the small launches (copies, fills, finalize kernels)

    python tools/prof_sequence.py <results.db> [which step from the end, default 2] [out.txt]
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = list(db.execute("select name, start, end from kernels order by start"))
# a step starts at the NCHW -> NHWC conversion of the input batch
starts = [i for i, r in enumerate(rows) if "k_nchw_to_nhwc" in r[0]]
a, b = starts[-back - 1], starts[-back]
out = []
prev_end = rows[a - 1][2] if a > 0 else rows[a][1]
for n, s, e in rows[a:b]:
    nm = re.sub(r"\(.*", "", n).replace("void ", "").replace("iswm::", "")[:70]
    out.append("%-70s %8.1f us  gap %6.1f us" % (nm, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
txt = "\n".join(out)
print("%d launches in the step, %.2f ms from first start to last end" % (b - a, (rows[b - 1][2] - rows[a][1]) / 1e6))
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(txt + "\n")
else:
    print(txt)
