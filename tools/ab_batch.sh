#!/bin/bash
# GPU box: the library against the variant build named "old" (raytracing_amd/libraytracing_amd.old.so), alternating,
# on the batched casa benchmark.  usage: tools/ab_batch.sh [extra bench.py arguments]
out=gpurun_out/ab_batch
mkdir -p $out
run() { # lib batch pipeline args...
    local lib=$1 b=$2 pl=$3; shift 3
    if [ "$lib" = default ]; then unset RAYTRACING_AMD_LIB; else export RAYTRACING_AMD_LIB=$PWD/raytracing_amd/libraytracing_amd.$lib.so; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-profile --batch $b --pipeline $pl --steps 64 --warmup 16 "$@" > $out/${lib}_b${b}_p${pl}.json 2> $out/${lib}.err || { echo "$lib failed"; tail -5 $out/${lib}.err; return 1; }
    python -c "import json; j=json.load(open('$out/${lib}_b${b}_p${pl}.json')); print('%-8s batch %2d pipeline %d: %.3f ms/sample %.0f Mrays/s' % ('$lib', $b, $pl, j['ms_per_step'], j['value']), flush=True)"
}
for rep in 1 2; do
  for cfg in "16 1" "8 2" "1 1"; do
    for lib in default old; do
      run $lib $cfg "$@" || exit 1
    done
  done
done
