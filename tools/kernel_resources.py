#!/usr/bin/env python3
"""Registers, spills, scratch, LDS and occupancy of every kernel of the library, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage; device code only, nothing is written).  No GPU needed.
Usage: python tools/kernel_resources.py [substring of a kernel name] [-DMACRO=VALUE ...]"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracing_amd import build as B  # noqa: E402

want = next((a for a in sys.argv[1:] if not a.startswith("-")), "")
defines = [a for a in sys.argv[1:] if a.startswith("-D")]
keys = ("VGPRs", "AGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]")
for rel in B.DEVICE_SOURCES:
    if not rel.startswith("device/"):
        continue
    cmd = [B._hipcc(), "--offload-arch=gfx950", *B.COMMON, *defines, *B.EXTRA.get(rel, []), "-Rpass-analysis=kernel-resource-usage",
           "--cuda-device-only", "-c", os.path.join(B.CSRC, rel), "-o", os.devnull]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr)
    cur = None
    rows = []
    for line in r.stderr.splitlines():
        m = re.search(r"remark: \s*(.*?): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            name = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
            name = name.replace("rta::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            cur = {"name": name}
            rows.append(cur)
        elif cur is not None:
            cur[m.group(1)] = m.group(2)
    for c in rows:
        if want in c["name"]:
            print("%-42s %s" % (c["name"][:42], "  ".join("%s %s" % (k.split(" [")[0], c.get(k, "?")) for k in keys)))
