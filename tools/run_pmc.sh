#!/bin/bash
# Collects rocprofv3 PMC counters for one bench workload, one pass per counter group
# (rocprofv3 does not split counters over passes).  Usage on the GPU box:
#   tools/run_pmc.sh <outdir> <bench args...>
# For the path-tracing workloads pass --pipeline 1 --tail 0: draws one at a time, every vertex in launches of its
# own, which is what bench.py's roofline leg times (the profiler serialises kernels anyway; what matters is that
# the launches have the grids and the rays of the ones the launch time is taken from).
set -u
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
# bench.py asks for 16 hardware queues, but under rocprofv3 the HIP runtime starts before bench.py runs

groups=(
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
  "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
  "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
  "FETCH_SIZE"
  "WRITE_SIZE"
)
i=0
for g in "${groups[@]}"; do
  timeout -s KILL 120 rocprofv3 --pmc $g --output-format csv -d "$out/pass$i" -- python3 bench.py "$@" --no-cpu-baseline --no-profile > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out/pass$i.log"; exit 1; }
  i=$((i+1))
done
echo "pmc ok"
