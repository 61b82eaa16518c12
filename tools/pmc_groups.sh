#!/bin/bash
# GPU box: one rocprofv3 --pmc pass per counter group (a pass holds at most 8 SQ, 4 TCP, 2 TA, 2 TD, 2 GRBM counters),
# then per-kernel sums with the template arguments kept apart.
# usage: tools/pmc_groups.sh OUTDIR "GROUP 1 COUNTERS" "GROUP 2 COUNTERS" ... -- bench.py arguments
out=$1; shift
groups=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do groups+=("$1"); shift; done
shift
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for g in "${groups[@]}"; do
  timeout -s KILL 150 rocprofv3 --pmc $g --output-format csv -d "$out/pass$i" -- python3 bench.py "$@" --no-cpu-baseline --no-profile > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out/pass$i.log"; exit 1; }
  i=$((i+1))
done
python3 - "$out" <<'PY'
import csv, glob, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("rta::", "").replace("void ", "").split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", tot[k].get("GRBM_GUI_ACTIVE", 0))):
    if "rocclr" in k: continue
    print(k)
    for c in sorted(tot[k]):
        print("  %-36s total %.6g  launches %d  per-launch %.6g" % (c, tot[k][c], calls[k][c], tot[k][c] / calls[k][c]))
PY
