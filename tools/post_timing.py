#!/usr/bin/env python3
"""Times the element-wise and stencil passes behind the path tracer at 1920x1080 (device time per
call from 50 back-to-back calls) and prints their algorithmic bytes per pixel and GB/s."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracing_amd import device  # noqa: E402

W, H = 1920, 1080
n = W * H
ctx = device.Context(0)
rng = np.random.default_rng(0)
B = lambda c: device.Buffer(ctx, rng.random(n * c, np.float32))
normal = rng.standard_normal((n, 3)).astype(np.float32)
normal /= np.linalg.norm(normal, axis=1, keepdims=True)
b = dict(c3a=B(3), c3b=B(3), c3c=B(3), c3d=B(3), c2a=B(2), c2b=B(2), c1a=B(1), c1b=B(1), c1c=B(1), c1d=B(1),
         normal=device.Buffer(ctx, normal), motion=device.Buffer(ctx, np.zeros(n * 2, np.float32)))
b["c1a"].fill(3.0)  # short histories: the variance filter takes its 7x7 branch


def timed(name, bytes_per_pixel, fn, reps=50):
    fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("%-22s %7.3f ms  %4d B/pixel  %7.1f GB/s" % (name, ms, bytes_per_pixel, n * bytes_per_pixel / ms / 1e6))


timed("merge", 48, lambda: device.post_merge(ctx, b["c3a"], b["c3b"], b["c3c"], b["c3d"]))
timed("static_blend", 64, lambda: device.post_static_blend(ctx, 5, b["c3a"], b["c3b"], b["c2a"]))
timed("tonemap", 24, lambda: device.post_tonemap(ctx, b["c3a"], b["c3d"]))
timed("svgf_temporal", 12 + 12 + 4 + 8 + 24 + 36, lambda: device.post_svgf_temporal(
    ctx, W, H, False, b["c3a"], b["normal"], b["c1b"], b["motion"], b["c3d"], b["c2b"], b["c1c"], b["c3b"], b["c2a"], b["c1a"],
    b["normal"], b["c1b"]))
timed("svgf_variance 7x7", 8 + 12 + 4 + 4 + 4, lambda: device.post_svgf_variance(ctx, W, H, b["c2a"], b["normal"], b["c1d"], b["c1a"], b["c1c"]))
timed("svgf_spatial step 1", 12 + 12 + 4 + 4 + 12, lambda: device.post_svgf_spatial(ctx, W, H, 1, b["c3a"], b["normal"], b["c1d"], b["c1b"], b["c3d"]))
timed("svgf_spatial step 8", 12 + 12 + 4 + 4 + 12, lambda: device.post_svgf_spatial(ctx, W, H, 8, b["c3a"], b["normal"], b["c1d"], b["c1b"], b["c3d"]))
timed("taa", 12 + 12 + 8 + 12 + 4 + 12 + 4 + 12, lambda: device.post_taa(ctx, W, H, b["c3a"], b["c3b"], b["motion"], b["normal"], b["c1b"],
                                                                      b["normal"], b["c1b"], b["c3d"]))
