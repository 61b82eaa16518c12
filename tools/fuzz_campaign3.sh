#!/bin/bash
# GPU box: both fuzzers and the soak after a change to the traversal kernels.
out=gpurun_out/fuzz_campaign
mkdir -p $out
for seed in 1 4; do
  timeout -k 10 900 python -u tools/flat_fuzz.py 135 20000 $seed > $out/flat3_$seed.log 2>&1 || { echo "flat_fuzz seed $seed FAILED"; tail -5 $out/flat3_$seed.log; exit 1; }
  tail -1 $out/flat3_$seed.log
done
timeout -k 10 900 python -u tools/render_fuzz.py 110 2 > $out/render3.log 2>&1 || { echo "render_fuzz FAILED"; tail -5 $out/render3.log; exit 1; }
tail -1 $out/render3.log
timeout -k 10 600 python -u tools/soak.py 90 > $out/soak3.log 2>&1 || { echo "soak FAILED"; tail -5 $out/soak3.log; exit 1; }
tail -1 $out/soak3.log
