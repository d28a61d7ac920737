"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch)."""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for pat in sys.argv[1:]:
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "shk::" not in k:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("   %-28s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
