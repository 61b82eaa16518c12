#!/bin/bash
# On the GPU box: the bench lines kept under profiles/ (plain runs, no profiler attached).
# Usage: tools/bench_all.sh <tag> [--big]     results: gpurun_out/<tag>_<workload>.json
cd "$(dirname "$0")/.."
tag=$1; shift
run() {  # name, seconds, bench args...
  n=$1; s=$2; shift 2
  timeout -k 10 $s python3 -u bench.py "$@" > gpurun_out/${tag}_$n.json 2> gpurun_out/${tag}_$n.err || { echo "$n failed"; tail -5 gpurun_out/${tag}_$n.err; exit 1; }
  python3 -c 'import sys,json; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], round(d["value"],1), d["unit"], round(d["ms_per_step"],3), "ms/step", "serial", (d.get("kernel_ms") or {}).get("one_draw_in_flight_ms_per_step"))' gpurun_out/${tag}_$n.json $n
}
run casa_pt 200 || exit 1                                            # the default line: 96 samples, 16 per draw, three draws in flight
run casa_pt_rep2 200 || exit 1                                       # (twice more: run-to-run scatter)
run casa_pt_rep3 200 || exit 1
run casa_pt_20_steps 200 --steps 20 --warmup 5 || exit 1             # the shape the driver runs
run casa_pt_batch_serial 200 --steps 64 --warmup 16 --pipeline 1 --batch 16 || exit 1   # one batched draw in flight
run casa_pt_serial 200 --steps 32 --warmup 4 --pipeline 1 --batch 1 || exit 1           # one single-sample draw in flight
run casa_isect 200 --workload casa_isect --steps 8 || exit 1
run synth_pt 400 --workload synth_pt || exit 1
if [ "$1" = "--big" ]; then
  run config5_10m 600 --workload synth_pt --steps 8 --warmup 2 --depth 16 --dielectric 0.5 --textures 16 --texture-size 4096 || exit 1
  run config5_50m 1000 --workload synth_pt --tris 50000000 --steps 8 --warmup 2 --depth 16 --dielectric 0.5 --textures 16 --texture-size 4096 || exit 1
fi
