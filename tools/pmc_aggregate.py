"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel over the WHOLE run of tools/pmc_step.py
(2 identical training steps) -> bytes per launch.  usage: pmc_aggregate.py FETCH.csv WRITE.csv out.json"""
import csv, json, re, sys, collections

def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)           # drop the argument list
    return name.replace("iswm::", "")

def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(row["Counter_Value"])
    return agg

fetch = load(sys.argv[1], "FETCH_SIZE")     # KiB
write = load(sys.argv[2], "WRITE_SIZE")     # KiB
out = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    n = fetch.get(k, write.get(k))[0]
    fb = fetch.get(k, [0, 0.0])[1] * 1024.0
    wb = write.get(k, [0, 0.0])[1] * 1024.0
    out[k] = {"launches_2_steps": n, "fetch_size_bytes_per_launch_raw": fb / n,
              "write_size_bytes_per_launch": wb / n,
              # gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> x2 (MI355X guide, HBM)
              "hbm_bytes_per_launch": (2.0 * fb + wb) / n}
import hashlib, os
h = hashlib.sha256()
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "iswm_amd", "csrc")
for f in sorted(os.listdir(d)):
    if f.endswith((".hip", ".h")):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
out["_csrc_sha16"] = h.hexdigest()[:16]          # bench.py reports `traffic` only for a build of these sources
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(((k, v) for k, v in out.items() if isinstance(v, dict)), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_2_steps"])[:12]:
    print("%-44s %5d launches  %9.1f MB/launch" % (k, v["launches_2_steps"], v["hbm_bytes_per_launch"] / 1e6))
