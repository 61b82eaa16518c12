#!/bin/bash
# Here, after tools/refresh_profiles.sh and tools/bench_all.sh <tag> --big have run on the GPU box: copies what is
# kept under profiles/.  Usage: tools/collect_profiles.sh pmc | bench <tag> <round prefix, e.g. r02>
cd "$(dirname "$0")/.."
what=$1; tag=$2; r=${3:-r03}
if [ "$what" = pmc ]; then
  for w in casa_pt casa_pt_shipped casa_isect synth_pt; do
    cp gpurun_out/${r}_pmc_$w.json profiles/${r}_pmc_$w.json && cp gpurun_out/${r}_pmc_$w.txt profiles/${r}_pmc_$w.txt || exit 1
  done
  for w in casa_pt casa_pt_batch_serial casa_pt_serial casa_isect synth_pt; do
    f=$(ls -t gpurun_out/prof_$w/*/*kernel_stats.csv 2>/dev/null | head -1)   # the newest: earlier runs leave theirs behind
    [ -n "$f" ] && cp "$f" profiles/${r}_${w}_kernel_stats.csv
    cp gpurun_out/prof_$w.json profiles/${r}_bench_${w}_under_rocprof.json
  done
else
  for w in casa_pt casa_pt_rep2 casa_pt_rep3 casa_pt_batch_serial casa_pt_serial casa_isect synth_pt config5_10m config5_50m casa_pt_20_steps; do
    [ -f gpurun_out/${tag}_$w.json ] && cp gpurun_out/${tag}_$w.json profiles/${r}_bench_$w.json
  done
fi
ls -la profiles | tail -25
