"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel (mean per dispatch)."""
import collections
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k[k.find("::") + 2:][:48] if "::" in k else k[:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        if any(t in k for t in ("group_", "so3", "gram", "lift", "wino", "vnsmall", "fft48", "cgemm", "conv_s2")):
            n = max(len(disp[k]), 1)
            print(k, f"dispatches={n}", {a: round(b / n) for a, b in sorted(v.items())})
