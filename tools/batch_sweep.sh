#!/bin/bash
# casa 1080p depth 8, HIP's default hardware queues: samples per draw x draws in flight.
# usage: tools/batch_sweep.sh OUT_PREFIX [extra bench.py arguments]
out=$1; shift
# environment: CFGS="batch:pipeline ..." STEPS WARMUP
for cfg in ${CFGS:-1:1 1:3 4:1 16:1 4:3 8:2 8:3 16:2 16:3}; do
  b=${cfg%%:*}; pl=${cfg##*:}
  timeout -k 10 120 python bench.py --hw-queues 0 --batch $b --pipeline $pl --steps ${STEPS:-256} --warmup ${WARMUP:-32} --no-cpu-baseline --no-profile "$@" \
     > ${out}_b${b}_p${pl}.json 2>> ${out}.err || exit 1
  python - <<PY
import json
j = json.load(open("${out}_b${b}_p${pl}.json"))
print("batch %2d pipeline %d: %.3f ms/sample  %.0f Mrays/s" % ($b, $pl, j["ms_per_step"], j["value"]), flush=True)
PY
done
