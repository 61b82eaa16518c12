#!/bin/bash
# GPU box: the default casa benchmark (batches of 16, three draws in flight) over variant builds of the library and over
# the blocks per CU of a traversal launch that shares the device.  usage: tools/sweep_defaults.sh
# env: LIBS="default pr20 ..." (variant names), BLOCKS="2 3 4" (RTA_DRAW_BLOCKS_PER_CU values), REPS="1 2"
cd "$(dirname "$0")/.."
out=gpurun_out/sweep_defaults.log
: > $out
one() {  # label, bench args...
  l=$1; shift
  timeout -k 10 200 python3 -u bench.py --no-cpu-baseline --no-profile "$@" > gpurun_out/sweep_one.json 2> gpurun_out/sweep_one.err || { echo "$l failed" | tee -a $out; tail -3 gpurun_out/sweep_one.err; exit 1; }
  python3 -c 'import sys,json; d=json.loads(open("gpurun_out/sweep_one.json").read()); print(sys.argv[1], round(d["ms_per_step"],4), "ms/step", round(d["value"],1))' "$l" | tee -a $out
}
for rep in ${REPS:-1 2}; do
  for lib in ${LIBS:-default}; do
    if [ "$lib" = default ]; then unset RAYTRACING_AMD_LIB; else export RAYTRACING_AMD_LIB=$PWD/raytracing_amd/libraytracing_amd.$lib.so; fi
    one "lib=$lib rep=$rep" "$@" || exit 1
  done
  unset RAYTRACING_AMD_LIB
  for b in ${BLOCKS:-}; do
    RTA_DRAW_BLOCKS_PER_CU=$b one "blocks=$b rep=$rep" "$@" || exit 1
  done
done
