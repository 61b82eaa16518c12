#!/bin/bash
# GPU box: the draw fuzzer over two seeds and the soak (after a change to shading).
out=gpurun_out/fuzz_campaign
mkdir -p $out
for seed in 1 5; do
  timeout -k 10 900 python -u tools/render_fuzz.py 110 $seed > $out/render_$seed.log 2>&1 || { echo "render_fuzz seed $seed FAILED"; tail -5 $out/render_$seed.log; exit 1; }
  tail -1 $out/render_$seed.log
done
timeout -k 10 600 python -u tools/soak.py 120 > $out/soak2.log 2>&1 || { echo "soak FAILED"; tail -5 $out/soak2.log; exit 1; }
tail -1 $out/soak2.log
