"""Sums rocprofv3 --pmc counter CSVs per kernel name.  Usage: summarize_pmc.py <dir> [kernel substring]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(d + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub and sub not in k:
            continue
        k = k.replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
for k in tot:
    print(k)
    for c in sorted(tot[k]):
        print("  %-32s total %.6g  launches %d  per-launch %.6g" % (c, tot[k][c], calls[k][c], tot[k][c] / calls[k][c]))
