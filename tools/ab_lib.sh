#!/bin/bash
# GPU box: the library against the variant build named "old" on every bench workload, alternating.
out=gpurun_out/ab_lib
mkdir -p $out
run() { # name lib args...
    local name=$1 lib=$2; shift 2
    if [ "$lib" = default ]; then unset RAYTRACING_AMD_LIB; else export RAYTRACING_AMD_LIB=$PWD/raytracing_amd/libraytracing_amd.$lib.so; fi
    timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > $out/${name}_$lib.json 2> $out/${name}_$lib.err || { echo "$name $lib failed"; tail -5 $out/${name}_$lib.err; return 1; }
    python - $out/${name}_$lib.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "Mrays/s %.1f" % d["value"], "Msamples/s %.1f" % (d.get("msamples_per_s") or 0), "ms/step %.4f" % d["ms_per_step"], "serial", (d.get("kernel_ms") or {}).get("one_draw_in_flight_ms_per_step"))
PY
}
for lib in default old default old; do
  run casa_pt $lib --steps 64 --warmup 16 || exit 1
done
for lib in default old default old; do
  run synth_pt $lib --workload synth_pt --steps 12 --warmup 4 || exit 1
done
for lib in default old; do
  run config5 $lib --workload synth_pt --steps 8 --warmup 2 --depth 16 --dielectric 0.5 --textures 16 --texture-size 4096 || exit 1
  run isect_two $lib --workload casa_isect --traversal 2 --steps 8 || exit 1
done
