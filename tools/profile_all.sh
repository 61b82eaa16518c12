#!/bin/bash
# On the GPU box: rocprofv3 kernel statistics + the bench's own JSON line for every workload.
# Results under gpurun_out/prof_<tag>/ ; copy the *_kernel_stats.csv and *.json into profiles/.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
run() {  # tag, bench args...
  tag=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py "$@" > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err || { echo "$tag failed"; tail -5 gpurun_out/prof_$tag.err; exit 1; }
  echo "$tag ok"
}
run casa_pt || exit 1
run casa_pt_batch_serial --steps 64 --warmup 16 --pipeline 1 --batch 16 || exit 1
run casa_pt_serial --steps 16 --warmup 2 --pipeline 1 --batch 1 || exit 1
run casa_isect --workload casa_isect --steps 8 || exit 1
run synth_pt --workload synth_pt || exit 1
