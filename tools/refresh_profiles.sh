set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
r=${1:-r03}
bash tools/profile_all.sh
# counter passes of the launches the roofline leg of bench.py times: the timed region's batches, one draw in flight, no tail
# launch, every stage a launch of its own (RTA_FUSED_PRIMARY=0: what the library does while per-kernel profiling is on) ...
RTA_FUSED_PRIMARY=0 bash tools/run_pmc.sh gpurun_out/pmc_casa_pt --steps 32 --warmup 16 --pipeline 1 --tail 0 --batch 16
# ... and of the same draws as shipped (camera rays and vertex 0 in one launch, coherent bounce rays as groups): per-kernel
# figures of those kernels and the HBM bytes per sample
bash tools/run_pmc.sh gpurun_out/pmc_casa_pt_shipped --steps 32 --warmup 16 --pipeline 1 --tail 0 --batch 16
python3 tools/summarize_pmc.py gpurun_out/pmc_casa_pt_shipped --json gpurun_out/${r}_pmc_casa_pt_shipped.json --samples 48 --batch 16 > gpurun_out/${r}_pmc_casa_pt_shipped.txt
bash tools/run_pmc.sh gpurun_out/pmc_casa_isect --workload casa_isect --steps 4 --warmup 1
bash tools/run_pmc.sh gpurun_out/pmc_synth_pt --workload synth_pt --steps 12 --warmup 6 --pipeline 1 --tail 0 --batch 6
python3 tools/summarize_pmc.py gpurun_out/pmc_casa_pt --json gpurun_out/${r}_pmc_casa_pt.json --samples 48 --batch 16 > gpurun_out/${r}_pmc_casa_pt.txt
python3 tools/summarize_pmc.py gpurun_out/pmc_casa_isect --json gpurun_out/${r}_pmc_casa_isect.json > gpurun_out/${r}_pmc_casa_isect.txt
python3 tools/summarize_pmc.py gpurun_out/pmc_synth_pt --json gpurun_out/${r}_pmc_synth_pt.json --samples 18 --batch 6 > gpurun_out/${r}_pmc_synth_pt.txt
