"""Experiment, GPU box: what is ray coherence worth to the traversal?  The 16 Mi random casa rays as they are,
sorted by direction octant, by origin cell (16^3 grid, Morton order), and by cell then octant."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import device, scene  # noqa: E402


def morton(c):
    out = np.zeros(len(c), np.uint32)
    for b in range(4):
        for a in range(3):
            out |= ((c[:, a] >> b) & 1).astype(np.uint32) << (3 * b + a)
    return out


def main():
    sd = scene.casa_scene(os.path.join(ROOT, "assets", "casa_obj.glb"))
    ctx = device.Context(0)
    sb = device.SceneBuffers(ctx, sd)
    p = device.PathTracingPass(ctx, 16, 16)
    p.bind_scene(sb)
    lo, hi = sd.world_box()
    n = 1 << 24
    rays = scene.random_rays(lo, hi, n, 7)
    octant = ((rays[:, 3] < 0).astype(np.uint32) | ((rays[:, 4] < 0).astype(np.uint32) << 1) | ((rays[:, 5] < 0).astype(np.uint32) << 2))
    cell = morton(np.minimum(15, ((rays[:, :3] - lo) / (hi - lo) * 16).astype(np.int64)).astype(np.uint32))
    orders = {"as generated": np.arange(n), "by octant": np.argsort(octant, kind="stable"), "by origin cell": np.argsort(cell, kind="stable"),
              "by cell, then octant": np.argsort(cell * 8 + octant, kind="stable"), "by octant, then cell": np.argsort(octant * 4096 + cell, kind="stable")}
    hb = device.Buffer(ctx, nbytes=n * device.HIT_DTYPE.itemsize)
    for name, order in orders.items():
        rb = device.Buffer(ctx, np.ascontiguousarray(rays[order]))
        for _ in range(2):
            p.intersect(rb, n, hb)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            p.intersect(rb, n, hb)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        print("%-22s %8.1f Mrays/s" % (name, 5 * n / dt / 1e6), flush=True)
        rb.release()


if __name__ == "__main__":
    main()
