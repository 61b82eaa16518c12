#!/usr/bin/env python3
"""How long a trace launch takes when it carries few rays: pt_pass_intersect over the first n of a
set of random casa rays, device time per launch from the library's own events."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracing_amd import device, scene

sd = scene.casa_scene(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets", "casa_obj.glb"))
ctx = device.Context(0)
sb = device.SceneBuffers(ctx, sd)
p = device.PathTracingPass(ctx, 16, 16)
p.bind_scene(sb)
lo, hi = sd.world_box()
rays = scene.random_rays(lo, hi, 1 << 20, 11)
rb = device.Buffer(ctx, rays)
hb = device.Buffer(ctx, nbytes=len(rays) * device.HIT_DTYPE.itemsize)
p.intersect(rb, len(rays), hb)
ctx.synchronize()
p.set_profiling(True)
for mode in (0, 1):
    p.set_traversal(mode)
    for n in (1, 64, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20):
        p.profile(reset=True)
        for _ in range(10):
            p.intersect(rb, n, hb)
        pr = p.profile()
        print("mode %d  n %8d  %.1f us per launch (both kernels)" % (mode, n, pr["intersect_ms"] / pr["intersect_launches"] * 1e3), flush=True)
