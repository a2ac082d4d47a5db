"""HBM traffic of the conv kernels PER GEOMETRY: pairs the dispatches of a rocprofv3 --pmc run of tools/pmc_step.py with the
launch sequence that run wrote (PMC_SEQ=file: kernel name, geometry tag and FLOPs of every conv launch of the second step, in order).
usage: pmc_by_geometry.py FETCH.csv WRITE.csv seq.json > table
Algorithmic bytes per launch (planes format, SURVEY 8d restated for the 6-byte activation layout): weight gradient x*6 + dy*6 + dw*4;
forward x*6 + w*6 + y*4; data gradient dy*6 + w*6 + dx*4 (an accumulating one also reads dx and, fused with a BatchNorm backward,
the producer's output and mask: not in the figure -- read its ratio with that in mind)."""
import collections, csv, json, re, sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("iswm::", "")


def load(path, counter):
    d = collections.defaultdict(list)            # kernel -> [(dispatch id, value)]
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                d[short(row["Kernel_Name"])].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    for k in d:
        d[k].sort()
    return d


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
seq = json.load(open(sys.argv[3]))
per_kernel = collections.defaultdict(list)       # kernel -> [(tag, flops)] in launch order
for name, tag, fl in seq:
    per_kernel[name.replace("+reduce", "")].append((tag, fl))


def alg_bytes(kernel, tag):
    m = re.match(r"n(\d+) (\d+)x(\d+) c(\d+)->(\d+) k(\d+) s(\d+) d(\d+)", tag)
    if not m:
        return None
    n, h, w, cin, cout, k, s, d = map(int, m.groups())
    ho, wo = (h - 1) // s + 1, (w - 1) // s + 1            # "same" padding everywhere on this path
    x, y, wt = n * h * w * cin, n * ho * wo * cout, cout * cin * k * k
    if "wgrad" in kernel:
        return 6 * x + 6 * y + 4 * wt
    dgrad = re.search(r"true(, false, 0)?>$", kernel) is not None
    return 6 * y + 6 * wt + 4 * x if dgrad else 6 * x + 6 * wt + 4 * y


rows = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for kernel, launches in per_kernel.items():
    f, w = fetch.get(kernel, []), write.get(kernel, [])
    n = len(launches)
    if n == 0 or len(f) < n or len(w) < n:
        continue
    f, w = f[-n:], w[-n:]                                  # the second step's dispatches
    for (tag, fl), (_, fv), (_, wv) in zip(launches, f, w):
        r = rows[(kernel, tag)]
        r[0] += 1
        r[1] += (2.0 * fv + wv) * 1024.0                   # gfx950: FETCH_SIZE x2 (MI355X guide), KiB
        r[2] += wv * 1024.0
        r[3] = alg_bytes(kernel, tag) or 0.0
print("%-40s %-36s %5s %10s %10s %10s %6s" % ("kernel", "geometry", "calls", "MB/launch", "written", "algorithmic", "ratio"))
tot_m = collections.defaultdict(float); tot_a = collections.defaultdict(float); tot_n = collections.defaultdict(int)
for (kernel, tag), (n, b, wb, alg) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print("%-40s %-36s %5d %10.1f %10.1f %10.1f %6.2f" % (kernel, tag, n, b / n / 1e6, wb / n / 1e6, alg / 1e6, b / n / alg if alg else 0.0))
    tot_m[kernel] += b; tot_a[kernel] += alg * n; tot_n[kernel] += n
print()
for k in sorted(tot_m, key=lambda k: -tot_m[k]):
    if tot_a[k]:
        print("%-40s %5d launches  measured %8.1f MB/launch  algorithmic %8.1f  ratio %.2f" % (k, tot_n[k], tot_m[k] / tot_n[k] / 1e6, tot_a[k] / tot_n[k] / 1e6, tot_m[k] / tot_a[k]))
