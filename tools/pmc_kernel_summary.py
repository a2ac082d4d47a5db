"""mean counter values per launch of the kernels whose name contains a pattern, from a rocprofv3 --pmc counter_collection.csv
usage: pmc_kernel_summary.py file.csv pattern"""
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
csv.field_size_limit(1 << 30)
with open(sys.argv[1], newline="") as f:
    for row in csv.DictReader(f):
        if sys.argv[2] in row["Kernel_Name"]:
            k = row["Counter_Name"]
            agg[k][0] += 1
            agg[k][1] += float(row["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print("%-32s %6d launches  mean %16.1f" % (k, n, v / n))
