#!/bin/bash
# On the GPU box: the parity tests that cover every kernel variant, then the batch benchmark and its kernel statistics.
# usage: tools/quick_check.sh TAG
tag=$1
timeout -k 10 600 python -u -m pytest tests/test_gpu_render.py tests/test_gpu_variants.py tests/test_gpu_batch.py tests/test_gpu_tail.py tests/test_gpu_golden.py tests/test_gpu_raster_first.py tests/test_gpu_flat.py -x -q --timeout 200 --timeout-method thread > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -4 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && exit $rc
for cfg in "16 1" "8 2"; do
  set -- $cfg
  timeout -k 10 120 python bench.py --batch $1 --pipeline $2 --steps 64 --warmup 16 --no-cpu-baseline --no-profile > gpurun_out/${tag}_b$1_p$2.json 2>> gpurun_out/${tag}.err || exit 1
  python -c "import json; j=json.load(open('gpurun_out/${tag}_b$1_p$2.json')); print('batch $1 pipeline $2: %.3f ms/sample %.0f Mrays/s' % (j['ms_per_step'], j['value']))"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --batch 16 --pipeline 1 --steps 64 --warmup 16 --no-cpu-baseline --no-profile > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err || exit 1
find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${tag}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:8]:
    print("%-60s calls %4s  %7.3f ms/sample  %5s%%" % (r["Name"].replace("rta::(anonymous namespace)::", "")[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6 / 80, r["Percentage"]))
print("kernels: %.3f ms/sample" % (tot / 1e6 / 80))
PY
