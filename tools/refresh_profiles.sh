set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile_all.sh
bash tools/run_pmc.sh gpurun_out/pmc_casa_pt --steps 16 --warmup 2 --pipeline 1 --tail 0
bash tools/run_pmc.sh gpurun_out/pmc_casa_isect --workload casa_isect --steps 4 --warmup 1
bash tools/run_pmc.sh gpurun_out/pmc_synth_pt --workload synth_pt --steps 4 --warmup 1 --pipeline 1 --tail 0
for w in casa_pt casa_isect synth_pt; do python3 tools/summarize_pmc.py gpurun_out/pmc_$w --json gpurun_out/r02_pmc_$w.json > gpurun_out/r02_pmc_$w.txt; done
