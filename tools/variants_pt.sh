#!/bin/bash
# On the GPU box: casa path tracing with draws in flight and one draw at a time, for the real library and the named variant builds.
cd "$(dirname "$0")/.."
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], round(d["value"],1), "Mrays/s", round(d["ms_per_step"],3), "ms/step")'
for v in default "$@" default; do
  if [ "$v" = default ]; then unset RAYTRACING_AMD_LIB; else export RAYTRACING_AMD_LIB=$PWD/raytracing_amd/libraytracing_amd.$v.so; fi
  timeout -k 10 120 python3 -u bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "$fmt" $v pt || exit 1
  timeout -k 10 120 python3 -u bench.py --steps 32 --warmup 8 --pipeline 1 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "$fmt" $v pt_serial || exit 1
  timeout -k 10 120 python3 -u bench.py --steps 64 --warmup 16 --width 640 --height 480 --depth 2 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "$fmt" $v pt_640_d2 || exit 1
done
