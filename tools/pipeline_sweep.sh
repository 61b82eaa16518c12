#!/bin/bash
# GPU box: draws in flight against time per sample, casa 1080p and the 10 M-triangle scene.
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], round(d["value"],1), "Mrays/s", round(d["ms_per_step"],3), "ms/step")'
for n in 4 6 8 10 12 14 16; do
  timeout -k 10 120 python3 -u bench.py --steps 64 --warmup 16 --pipeline $n --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "$fmt" casa_pt $n || exit 1
done
for n in 3 6 10 14; do
  timeout -k 10 300 python3 -u bench.py --workload synth_pt --steps 16 --warmup 4 --pipeline $n --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "$fmt" synth_pt $n || exit 1
done
