"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: mean of every counter.  python tools/pmc_agg.py CSV [filter]"""
import collections
import csv
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"])[:100]
    if flt in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("    %-32s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
