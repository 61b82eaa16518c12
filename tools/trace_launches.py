#!/usr/bin/env python3
"""Prints, per kernel name, the durations of its launches in launch order (one draw's worth
from the end of a rocprofv3 --kernel-trace CSV), to see how time is spread over bounces."""
import csv, glob, sys, collections
d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:8.1f} us  {r["Kernel_Name"][:60]}')
