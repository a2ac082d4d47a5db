"""mean counter values per launch of the kernels matching a substring: pmc_summarize.py SUBSTR file.csv [file.csv ...]"""
import csv, sys, collections
sub = sys.argv[1]
for path in sys.argv[2:]:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path, newline="")):
        if sub in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print("%-34s n=%d mean %.4g" % (k, len(v), sum(v) / len(v)))
