#!/usr/bin/env python3
"""Per-basic-block instruction histogram of one kernel in a hipcc -S listing:
   asm_blocks.py listing.s kernel_name_substring"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if key in l and l.startswith('_Z') and ': ' in l][0]
end = [i for i, l in enumerate(lines) if i > start and '.amdhsa_kernel' in l][0]
blk = ['entry', 0, 0, 0, 0, 0]
stats = [blk]
for l in lines[start + 1:end]:
    l = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blk = [m.group(1), 0, 0, 0, 0, 0]
        stats.append(blk)
        continue
    if not l or l.startswith(('.', ';', '//')):
        continue
    op = l.split()[0]
    if op.startswith('v_'): blk[1] += 1
    elif op.startswith('s_'): blk[2] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): blk[3] += 1
    elif op.startswith('ds_'): blk[4] += 1
    if op.startswith(('v_div', 'v_rcp')): blk[5] += 1
print("block valu salu vmem lds div")
tot = [0] * 5
for b in stats:
    print(*b)
    for i in range(5): tot[i] += b[i + 1]
print("total", tot)
