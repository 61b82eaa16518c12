"""GPU box: a soak of the draw path.  Random sequences of settings (draws in flight, tail threshold, traversal
mode, node format, material grouping, MAX_DEPTH, sampling switches, moving instances) on one pass object, many
frames of a few samples each; every frame is compared with the oracle bit for bit.  Looks for what short tests
miss: state carried from one configuration into the next, lanes reused while still in flight, stale snapshots.
Usage: python tools/soak.py [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GPU_MAX_HW_QUEUES"] = "16"
from raytracing_amd import device, scene  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))
    sky = scene.procedural_sky(256, 128)
    scenes = [("casa", scene.casa_scene(os.path.join(ROOT, "assets", "casa_obj.glb")), scene.casa_camera()),
              ("synth", scene.synthetic_scene(30000, dielectric_fraction=0.5, lights=2, textures=2, texture_size=64), scene.synthetic_camera())]
    ctx = device.Context(0)
    t_end = time.time() + budget
    frames = 0
    for name, sd, cam in scenes * 1000:
        if time.time() > t_end:
            break
        w, h = [(160, 90), (97, 61), (8, 8), (1, 1), (257, 3), (64, 129)][int(rng.integers(0, 6))]   # edge tiles, single tiles, one pixel
        sb = device.SceneBuffers(ctx, sd, sky)
        osc = oracle.OracleScene(sd, sky)
        p = device.PathTracingPass(ctx, w, h)
        p.bind_scene(sb)
        for _ in range(12):
            if time.time() > t_end:
                break
            pipeline = int(rng.choice([1, 2, 3, 6, 10]))
            p.set_pipeline(pipeline)
            p.set_tail_threshold(int(rng.choice([-1, 0, 64, 100000, 1 << 30])))
            p.set_traversal(int(rng.random() < 0.15))
            p.set_wide_nodes(int(rng.choice([-1, 0, 1])))
            p.set_material_grouping(int(rng.choice([-1, 0, 1])))
            depth = int(rng.choice([0, 1, 2, 3, 5, 8, 16]))
            spp = int(rng.integers(1, 5))
            lights = len(sd.lights) if rng.random() < 0.5 else 0
            sw = dict(brdf_sampling=bool(rng.random() < 0.8), skybox_sampling=bool(rng.random() < 0.8), skybox_lighting=bool(rng.random() < 0.9))
            if rng.random() < 0.3:
                p.prepare()
            jit = scene.mt19937_jitters(spp)
            ref = None
            for s in range(1, spp + 1):
                prm = scene.make_params(w, h, cam, sky, spp=s, frame=1 + frames, jitter=jit[s - 1], max_depth=depth, light_count=lights, **sw)
                p.set_params(prm)
                p.draw()
                ref, _ = osc.render(prm, ref)
            got = p.download()
            for k in ref:
                a, b = np.ascontiguousarray(got[k]), np.ascontiguousarray(ref[k])
                ok = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
                if not ok.all():
                    print("MISMATCH frame %d scene %s output %s: %d words; pipeline %d depth %d spp %d lights %d %s" %
                          (frames, name, k, int((~ok).sum()), pipeline, depth, spp, lights, sw), "at %dx%d" % (w, h), flush=True)
                    sys.exit(1)
            frames += 1
            if frames % 200 == 0:
                print("%d frames ok (%.0f s left)" % (frames, t_end - time.time()), flush=True)
        p.destroy()
        sb.release()
    print("soak ok: %d frames" % frames)


if __name__ == "__main__":
    main()
