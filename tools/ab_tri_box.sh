#!/bin/bash
# GPU box: two levels against the flattened tree on scenes with rotated instances, then the fuzzers.
set -o pipefail
out=gpurun_out/ab_tri_box
mkdir -p $out
run() { # name args...
    local name=$1; shift 1
    timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > $out/${name}.json 2> $out/${name}.err || { echo "$name failed"; tail -5 $out/${name}.err; return 1; }
    python - $out/${name}.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "value %.1f" % d["value"], "ms/step %.3f" % d["ms_per_step"], "flat_leaves", d["config"].get("flat_leaves"), "serial", d.get("kernel_ms", {}).get("one_draw_in_flight_ms_per_step"), "h2d", d.get("host_to_device", {}).get("seconds"))
PY
}
run s1m_two --workload synth_pt --tris 1000000 --traversal 2 --steps 24 --warmup 6 &&
run s1m_flat --workload synth_pt --tris 1000000 --traversal 3 --steps 24 --warmup 6 &&
run s1m_1080_two --workload synth_pt --tris 1000000 --width 1920 --height 1080 --traversal 2 --steps 48 --warmup 12 &&
run s1m_1080_flat --workload synth_pt --tris 1000000 --width 1920 --height 1080 --traversal 3 --steps 48 --warmup 12 &&
run s3m_two --workload synth_pt --tris 3000000 --traversal 2 --steps 24 --warmup 6 &&
run s3m_flat --workload synth_pt --tris 3000000 --traversal 3 --steps 24 --warmup 6 &&
timeout -k 10 500 python -u tools/flat_fuzz.py 90 20000 11 > $out/fuzz.log 2>&1; echo "fuzz rc $?"; tail -1 $out/fuzz.log
timeout -k 10 500 python -u tools/render_fuzz.py 40 > $out/render_fuzz.log 2>&1; echo "render fuzz rc $?"; tail -1 $out/render_fuzz.log
