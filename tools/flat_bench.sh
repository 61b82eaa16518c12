set -e
for t in 2 0; do
  timeout -k 10 200 python bench.py --workload casa_isect --steps 4 --warmup 1 --traversal $t --no-cpu-baseline > gpurun_out/fb_isect_$t.json
  timeout -k 10 200 python bench.py --steps 64 --warmup 16 --traversal $t --no-cpu-baseline > gpurun_out/fb_pt_$t.json
done
timeout -k 10 400 python bench.py --workload synth_pt --steps 8 --warmup 2 --traversal 2 --no-cpu-baseline > gpurun_out/fb_synth_2.json
timeout -k 10 600 python bench.py --workload synth_pt --steps 8 --warmup 2 --traversal 3 --no-cpu-baseline > gpurun_out/fb_synth_3.json
