#!/bin/bash
# On the GPU box: casa_pt and synth_pt for several values of one tuning environment variable.
# Usage: tools/sweep_env.sh VAR value...
cd "$(dirname "$0")/.."
var=$1; shift
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["config"]["workload"][:12], round(d["value"],1), "Mrays/s", round(d["ms_per_step"],3), "ms/step")'
for v in "$@"; do
  export $var=$v
  timeout -k 10 120 python3 -u bench.py --steps 64 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "$fmt" "$var=$v" || exit 1
  timeout -k 10 300 python3 -u bench.py --workload synth_pt --steps 24 --warmup 4 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "$fmt" "$var=$v" || exit 1
done
