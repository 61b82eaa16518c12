#!/bin/bash
# Builds locally, then runs GPU tests + the two casa benches on an MI355X box.  Stops at the first failure.
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1 || { tail -30 /tmp/build.log; exit 1; }
EXTRA="${1:-}"
/usr/local/graft/bin/gpurun --timeout 600 -- "timeout -k 10 300 python -u -m pytest tests -x -q --timeout 120 --timeout-method thread -m gpu 2>&1 | tail -5 && timeout -k 10 300 python -u bench.py --steps 64 --warmup 4 $EXTRA 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(\"casa_pt\", round(d[\"value\"],1), \"Mrays/s\", round(d[\"ms_per_step\"],3), \"ms/step\", d.get(\"parity_vs_oracle\"), \"frac\", round(d[\"roofline\"][\"frac\"],3), {k:round(v,1) for k,v in d[\"kernel_ms\"].items() if v is not None})' && timeout -k 10 200 python -u bench.py --workload casa_isect --steps 4 $EXTRA 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(\"casa_isect\", round(d[\"value\"],1), \"Mrays/s\", round(d[\"ms_per_step\"],3), \"ms/step\", d.get(\"parity_vs_oracle\"), \"frac\", round(d[\"roofline\"][\"frac\"],3))'"
