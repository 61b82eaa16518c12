#!/usr/bin/env python3
"""Makes tests/golden/transforms.npz: matrices that the REFERENCE's own compiled code produces --
Transform::matrix (src/Transform.cpp:20-53), Instance::matrix_to_global (src/instance/Instance.cpp:38-41),
Camera::w2v_matrix / v2w_matrix / projection (src/instance/Camera.cpp:13-43) run by oracle/_ref/ref_harness --
for the poses main.cpp uses and a seeded set of others (negative and zero scales, angles beyond a turn,
large translations).  Build container only (needs /root/reference).
Usage: python tests/golden/make_transforms.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402


def records():
    rng = np.random.default_rng(20261113)
    rows = []
    ident = [0, 0, 0, 0, 0, 0, 1, 1, 1]
    # main.cpp:384-386 (camera pose), main.cpp:366-372 (scene root, o1 rotated -90 about x)
    rows.append([np.pi / 3, 1000.0] + ident + [-16.9924, 8.50888, 17.9285, -23.19, -48.8484, 0, 1, 1, 1])
    rows.append([np.pi / 3, 1000.0] + [0, 0, 0, -90, 0, 0, 1, 1, 1] + ident)
    for k in range(62):
        fov = float(rng.uniform(0.2, 2.8))
        far = float(rng.choice([10.0, 100.0, 1000.0, 1e5]))
        pr = []
        for _ in range(2):
            pos = rng.uniform(-50, 50, 3) * (1000.0 if k % 9 == 0 else 1.0)
            rot = rng.uniform(-720, 720, 3) if k % 2 else rng.integers(-8, 9, 3) * 45.0
            scl = rng.choice([-2.0, -1.0, 0.25, 1.0, 1.0, 3.0, 1e3], 3)
            pr += list(pos) + list(rot) + list(scl)
        rows.append([fov, far] + pr)
    return np.array(rows, np.float32)


def main():
    assert oracle.have_ref(), "oracle/_ref/ref_harness is not built"
    rec = records()
    wh, mats = oracle.ref_transforms(rec)
    dst = os.path.join(HERE, "transforms.npz")
    np.savez_compressed(dst, records=rec, screen_wh=np.array(wh, np.int32), matrices=mats)
    print(dst, os.path.getsize(dst), "bytes", "screen", wh)


if __name__ == "__main__":
    main()
