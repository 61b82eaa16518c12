#!/usr/bin/env python3
"""TEST INFRASTRUCTURE, build container only.  A HIGH-SPP frame of the reference's own shader text run as C++ with the C
library's sin / cos / atan / asin / pow (standing for a GL driver's, whose transcendentals differ from this build's in the
last bits): the demo scene, 96x54, 16 spp accumulated by the shader's running mean (shader/pathtracing.glsl:657-670),
MAX_DEPTH 4 (every Sobol dimension the reference defines), per-sample jitter from std::mt19937(0) as main.cpp:170-172 draws it.
Writes tests/golden/glsl_spp16.npz: directLum, indirectLum, albedo of that frame.  tests/test_oracle.py (oracle) and
tests/test_gpu_golden.py (HIP pass) check that their 16-spp frames differ from it by far less than the Monte Carlo noise of
16 samples -- the one-ulp-of-cosine pixels that miss the 1e-3 bar of a single sample wash out.
Usage: python oracle/glsl_spp16.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != HERE]
sys.path.insert(0, ROOT)

W, H, DEPTH, SPP = 96, 54, 4, 16


def main():
    from oracle import glsl_frame, oracle
    from raytracing_amd import scene
    oracle.build()
    exe = glsl_frame.build(False)
    sky = scene.procedural_sky(256, 128)
    sd = scene.casa_scene(os.path.join(ROOT, "assets", "casa_obj.glb"))
    camera = scene.casa_camera()
    osc = oracle.OracleScene(sd, sky)
    jit = scene.mt19937_jitters(SPP)
    gbufs = []
    for s in range(SPP):
        prim, _ = osc.render(scene.make_params(W, H, camera, sky, spp=1, frame=1, jitter=jit[s], max_depth=0))
        gbufs.append((prim["depth"].reshape(H, W, 1), prim["normal"].reshape(H, W, 3), np.zeros((H, W, 2), np.float32),
                      prim["instanceIndex"].reshape(H, W, 1)))
    prm = scene.make_params(W, H, camera, sky, spp=1, frame=1, jitter=jit[0], max_depth=DEPTH)
    res = glsl_frame.run(exe, sd, sky, prm, jit, gbufs, os.path.join(HERE, "_ref"), processes=os.cpu_count() or 1)
    dst = os.path.join(ROOT, "tests", "golden", "glsl_spp16.npz")
    np.savez_compressed(dst, **{k: res[k] for k in ("directLum", "indirectLum", "albedo")})
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
