#!/bin/bash
# GPU box: both fuzzers over several seeds, then the random-settings soak; stops at the first mismatch.
out=gpurun_out/fuzz_campaign
mkdir -p $out
for seed in 1 2 3; do
  timeout -k 10 900 python -u tools/flat_fuzz.py 135 20000 $seed > $out/flat_$seed.log 2>&1 || { echo "flat_fuzz seed $seed FAILED"; tail -5 $out/flat_$seed.log; exit 1; }
  tail -1 $out/flat_$seed.log
done
timeout -k 10 900 python -u tools/render_fuzz.py 110 > $out/render.log 2>&1 || { echo "render_fuzz FAILED"; tail -5 $out/render.log; exit 1; }
tail -1 $out/render.log
timeout -k 10 600 python -u tools/soak.py > $out/soak.log 2>&1 || { echo "soak FAILED"; tail -5 $out/soak.log; exit 1; }
tail -1 $out/soak.log
