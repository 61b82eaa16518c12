"""Sums rocprofv3 --pmc counter CSVs per kernel name.
Usage: summarize_pmc.py <dir> [kernel substring] [--json out.json] [--samples N] [--batch B]
(--samples: how many samples per pixel the profiled run drew, warm-up included; stored so that bytes per sample can be quoted;
 --batch: how many samples each draw of the profiled run carried, i.e. how fat its launches were; stored for the same reason)
With --json, also writes {kernel: {counter: {"total":, "launches":}}} plus HBM bytes per launch
(FETCH_SIZE and WRITE_SIZE are in KiB; for this library's access pattern -- 64-byte records
gathered with four 16-byte loads per lane -- tools/calib/fetch_calib.hip measured FETCH_SIZE
equal to the known byte count, so no gfx950 doubling is applied)."""
import csv
import glob
import json
import sys
from collections import defaultdict

def int_flag(name):
    if name not in sys.argv:
        return None
    k = sys.argv.index(name)
    v = int(sys.argv[k + 1])
    del sys.argv[k:k + 2]
    return v


samples = int_flag("--samples")
batch = int_flag("--batch")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
d = args[0]
sub = args[1] if len(args) > 1 and not args[1].endswith(".json") else ""
jout = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
if jout in args:
    args.remove(jout)
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(d + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub and sub not in k:
            continue
        k = k.replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1]
        k = k.split("<")[0]  # instantiations of one kernel template count together
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
for k in tot:
    print(k)
    for c in sorted(tot[k]):
        print("  %-32s total %.6g  launches %d  per-launch %.6g" % (c, tot[k][c], calls[k][c], tot[k][c] / calls[k][c]))
if jout:
    out = {}
    for k in tot:
        out[k] = {c: {"total": tot[k][c], "launches": calls[k][c]} for c in tot[k]}
        if "FETCH_SIZE" in tot[k] and "WRITE_SIZE" in tot[k]:
            out[k]["hbm_bytes_per_launch"] = (tot[k]["FETCH_SIZE"] / calls[k]["FETCH_SIZE"] +
                                              tot[k]["WRITE_SIZE"] / calls[k]["WRITE_SIZE"]) * 1024.0
    # which tree the counters were measured on (bench.py compares it with the tree it runs from)
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from raytracing_amd.build import kernel_sources_digest
    out["_meta"] = {"kernel_sources_sha256": kernel_sources_digest()}
    if batch:
        out["_meta"]["samples_per_draw"] = batch
    if samples:
        # every kernel of the run, the library's and the runtime's fills / copies: HBM bytes per sample per pixel-frame
        fetch = sum(tot[k].get("FETCH_SIZE", 0.0) for k in tot) * 1024.0
        write = sum(tot[k].get("WRITE_SIZE", 0.0) for k in tot) * 1024.0
        out["_meta"].update({"samples": samples, "hbm_bytes_per_sample": (fetch + write) / samples,
                             "hbm_read_bytes_per_sample": fetch / samples, "hbm_written_bytes_per_sample": write / samples})
    json.dump(out, open(jout, "w"), indent=1, sort_keys=True)
