#!/bin/bash
# GPU box: grouping inside the shade launches (pt_pass_set_material_grouping 0 / -1 / 1), one draw at a time and ten in flight.
cd "$(dirname "$0")/.."
fmt='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernel_ms") or {}; print(sys.argv[1], sys.argv[2], sys.argv[3], round(d["value"],1), "Mrays/s", round(d["ms_per_step"],3), "ms/step", "shade ms/draw", round(k.get("shade_ms",0)/max(1,k.get("over_draws",1)),3), d.get("parity_vs_oracle"))'
for g in 0 -1 0 -1; do
  timeout -k 10 200 python3 -u bench.py --steps 32 --warmup 8 --pipeline 1 --material-grouping $g 2>/dev/null | python3 -c "$fmt" casa serial $g || exit 1
  timeout -k 10 200 python3 -u bench.py --steps 64 --warmup 16 --material-grouping $g --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "$fmt" casa pipelined $g || exit 1
done
for g in 0 -1 1; do
  timeout -k 10 300 python3 -u bench.py --workload synth_pt --steps 12 --warmup 4 --material-grouping $g --no-cpu-baseline 2>/dev/null | python3 -c "$fmt" synth pipelined $g || exit 1
  timeout -k 10 400 python3 -u bench.py --workload synth_pt --steps 8 --warmup 2 --depth 16 --dielectric 0.5 --textures 16 --texture-size 4096 --material-grouping $g --no-cpu-baseline 2>/dev/null | python3 -c "$fmt" config5 pipelined $g || exit 1
done
