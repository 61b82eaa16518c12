#!/bin/bash
# GPU box: timeline of ONE single-sample draw (a caller that draws, waits, consumes) from rocprofv3's kernel trace.
# usage: tools/lone_timeline.sh [bench.py arguments, e.g. --width 1280 --height 720 --depth 2]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lone_timeline
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --pipeline 1 --batch 1 --steps 24 --warmup 4 --no-cpu-baseline --no-profile "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python3 - $out <<'PY'
import csv, glob, re, sys, json
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
fin = [i for i, r in enumerate(rows) if 'finalize_kernel' in r['Kernel_Name']]
a, b = fin[-3] + 1, fin[-2] + 1      # a draw of the timed region
t0 = int(rows[a]['Start_Timestamp'])
tot = 0
for r in rows[a:b]:
    n = re.sub(r'\(.*', '', r['Kernel_Name'].replace('rta::(anonymous namespace)::', '').replace('void ', ''))
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    tot += e - s
    print('%9.1f us +%8.1f us  %s' % ((s - t0) / 1e3, (e - s) / 1e3, n[:44]))
print('sum of kernels %.1f us, span %.1f us' % (tot / 1e3, (int(rows[b - 1]['End_Timestamp']) - t0) / 1e3))
print('bench: %.3f ms per draw' % json.load(open(sys.argv[1] + '/bench.json'))['ms_per_step'])
PY
