#!/bin/bash
# On the GPU box: casa frame time with one draw in flight over thresholds of the tail launch (--tail) and library variants.
# Usage: tools/tail_sweep.sh "<thresholds>" [variant...]
cd "$(dirname "$0")/.."
ths=${1:-"0 65536 131072 200000 300000 500000"}; shift
for v in default "$@"; do
  if [ "$v" = default ]; then unset RAYTRACING_AMD_LIB; else export RAYTRACING_AMD_LIB=$PWD/raytracing_amd/libraytracing_amd.$v.so; fi
  for th in $ths; do
    timeout -k 10 120 python3 -u bench.py --steps 64 --warmup 8 --pipeline 1 --tail $th --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'th', $th, round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms/step')" || exit 1
  done
done
