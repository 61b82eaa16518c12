#!/usr/bin/env python3
"""GPU check of the flattened tree: hits of traversal mode 3 (flattened tree) against mode 1 (reference order) and
mode 2 (ordered, two levels) on casa and on the hostile test scene, then the intersect rate of modes 2 and 3.
Usage: python tools/flat_check.py [n_rays_log2]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracing_amd import device, scene  # noqa: E402


def same(a, b):
    return all(np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k], b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k])
               for k in ("instance", "triangle", "t")) and np.array_equal(a["u"][a["triangle"] >= 0].view(np.uint32), b["u"][b["triangle"] >= 0].view(np.uint32))


def run(ctx, sd, name, n):
    lo, hi = sd.world_box()
    rays = scene.random_rays(lo, hi, n, 11)
    sb = device.SceneBuffers(ctx, sd)
    out = {}
    for mode in (1, 2, 3):
        p = device.PathTracingPass(ctx, 16, 16)
        p.bind_scene(sb)
        p.set_traversal(mode)
        hits = device.trace_rays(p, rays)
        rb = device.Buffer(ctx, rays)
        hb = device.Buffer(ctx, nbytes=len(rays) * device.HIT_DTYPE.itemsize)
        p.intersect(rb, len(rays), hb)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            p.intersect(rb, len(rays), hb)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 3
        st = p.stats()
        out[mode] = hits
        print(f"{name}: mode {mode}: {len(rays) / dt / 1e6:9.1f} Mrays/s  flat_leaves {st['flat_leaves']}  retraced {st['rays_retraced']}", flush=True)
        rb.release(); hb.release(); p.destroy()
    sb.release()
    print(f"{name}: mode 2 == mode 1: {same(out[2], out[1])}   mode 3 == mode 1: {same(out[3], out[1])}", flush=True)
    if not same(out[3], out[1]):
        a, b = out[3], out[1]
        bad = np.flatnonzero((a["triangle"] != b["triangle"]) | (a["instance"] != b["instance"]) | (a["t"].view(np.uint32) != b["t"].view(np.uint32)))
        print("  differing rays:", len(bad), bad[:8], a[bad[:4]], b[bad[:4]])
    return same(out[3], out[1])


def main():
    n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 22)
    ctx = device.Context(0)
    ok = run(ctx, scene.casa_scene(os.path.join(ROOT, "assets", "casa_obj.glb")), "casa", n)
    from conftest import hostile_scene
    ok = run(ctx, hostile_scene(), "hostile", 1 << 18) and ok
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
