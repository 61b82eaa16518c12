#!/bin/bash
# GPU box: per-kernel device time (rocprofv3 --kernel-trace --stats) of the batched casa benchmark, the library against
# the variant build named "old", same box, alternating.  usage: tools/ab_kernels.sh [bench.py arguments]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ab_kernels
rm -rf $out; mkdir -p $out
for rep in ${REPS:-1 2}; do
for lib in ${LIBS:-default old}; do
  if [ "$lib" = default ]; then unset RAYTRACING_AMD_LIB; else export RAYTRACING_AMD_LIB=$PWD/raytracing_amd/libraytracing_amd.$lib.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${lib}$rep -- python3 bench.py --batch 16 --pipeline 1 --steps 128 --warmup 16 --no-cpu-baseline --no-profile "$@" > $out/${lib}$rep.json 2> $out/${lib}$rep.err || { tail -5 $out/${lib}$rep.err; exit 1; }
  f=$(find $out/${lib}$rep -name "*kernel_stats.csv" | head -1)
  python - $f $lib <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
n = 144.0
print(sys.argv[2], " ".join("%s %.3f" % (r["Name"].replace("rta::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:22], float(r["TotalDurationNs"]) / 1e6 / n) for r in rows[:6]), "| all %.3f ms/sample" % (tot / 1e6 / n), flush=True)
PY
done
done
