#!/bin/bash
# GPU box: how a FEW samples are best handed over (what a rank of a strong-scaling run is left with): samples per draw x
# draws in flight, ms per sample.  usage: tools/small_slices.sh
cd "$(dirname "$0")/.."
for n in 3 5 10; do
  for cfg in "1 3" "$n 1" "$(( (n + 1) / 2 )) 2" "$(( (n + 2) / 3 )) 3"; do
    set -- $cfg
    best=
    for rep in 1 2 3; do
      timeout -k 10 120 python3 bench.py --steps $n --warmup $n --batch $1 --pipeline $2 --no-cpu-baseline --no-profile > gpurun_out/ss.json 2> gpurun_out/ss.err || { tail -3 gpurun_out/ss.err; exit 1; }
      best="$best $(python3 -c "import json; print(round(json.load(open('gpurun_out/ss.json'))['ms_per_step'],3))")"
    done
    echo "samples $n batch $1 pipeline $2: $best ms/sample"
  done
done
