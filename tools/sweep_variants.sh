#!/bin/bash
# On the GPU box: runs the benches for every tuning variant of the library
# (raytracing_amd/libraytracing_amd.<variant>.so) and, first and last, the real one, all on the
# same box (boxes differ by a few percent).  Usage: tools/sweep_variants.sh [--synth] [variant...]
cd "$(dirname "$0")/.."
SYNTH=0; if [ "$1" = "--synth" ]; then SYNTH=1; shift; fi
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["config"]["workload"][:12], round(d["value"],1), "Mrays/s", round(d["ms_per_step"],3), "ms/step")'
for v in default "$@" default; do
  if [ "$v" = default ]; then unset RAYTRACING_AMD_LIB; else export RAYTRACING_AMD_LIB=$PWD/raytracing_amd/libraytracing_amd.$v.so; fi
  timeout -k 10 120 python3 -u bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "$fmt" $v || exit 1
  timeout -k 10 120 python3 -u bench.py --workload casa_isect --steps 6 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "$fmt" $v || exit 1
  if [ $SYNTH = 1 ]; then timeout -k 10 300 python3 -u bench.py --workload synth_pt --steps 24 --warmup 4 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "$fmt" $v || exit 1; fi
done
