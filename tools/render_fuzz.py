#!/usr/bin/env python3
"""Differential test of whole draws on the GPU against the CPU oracle: random scenes (tools/flat_fuzz.py's geometry
with random conductor / rough dielectric materials, sometimes textured, 0-3 analytic lights), a camera looking at
them, random MAX_DEPTH and sampling switches, three accumulated samples; every traversal mode, draws in flight or one at
a time, tail launch forced / off / the library's choice, single draws and batched ones.  All seven outputs must equal the oracle's word for word.
Usage: python tools/render_fuzz.py [n_scenes] [seed]"""
import os
import sys

import numpy as np

os.environ.setdefault("RTA_BOUNCE_GROUPS", "2")   # rays that leave a tile together walked as groups however small the draw
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from raytracing_amd import device, scene  # noqa: E402
from oracle import oracle  # noqa: E402
import flat_fuzz  # noqa: E402


def random_render_scene(rng):
    b = scene.SceneBuilder()
    tex = [b.add_texture(rng.integers(0, 256, (int(rng.integers(2, 40)), int(rng.integers(2, 40)), int(rng.choice([1, 3, 4]))), dtype=np.uint8))
           for _ in range(int(rng.integers(0, 3)))]
    meshes = []
    for _ in range(int(rng.integers(2, 5))):
        s = float(10.0 ** rng.uniform(-0.5, 0.7))
        meshes.append(b.add_mesh(flat_fuzz.soup(rng, int(rng.integers(1, 300)), s) if rng.random() < 0.4 else flat_fuzz.grid(rng, int(rng.integers(1, 12)), s)))
    rotated = rng.random() < 0.4
    for _ in range(int(rng.integers(2, 8))):
        m = meshes[int(rng.integers(0, len(meshes)))]
        pos = rng.uniform(-1, 1, 3) * 6.0
        rot = rng.uniform(0, 360, 3) if rotated else rng.integers(0, 4, 3) * 90.0
        sc = np.full(3, 10.0 ** rng.uniform(-0.5, 0.5))
        if rng.random() < 0.3:
            sc[int(rng.integers(0, 3))] *= -1.0
        mat = dict(type=int(rng.choice([1, 1, 2])), albedo=tuple(float(x) for x in rng.uniform(0.1, 0.95, 3)),
                   roughness=float(rng.choice([0.0, 0.001, 0.05, 0.3, 0.9])), metallic=float(rng.choice([0.0, 0.5, 1.0])),
                   ior=float(rng.uniform(1.05, 2.0)), tex=int(rng.choice(tex)) if tex and rng.random() < 0.6 else -1)
        b.add_instance(m, scene.transform_matrix(tuple(pos), tuple(rot), tuple(sc)), mat)
    for k in range(int(rng.choice([0, 0, 0, 1, 2, 3]))):
        if rng.random() < 0.5:
            b.add_point_light(scene.transform_matrix(tuple(rng.uniform(-8, 8, 3))), tuple(float(x) for x in rng.uniform(10, 400, 3)))
        else:
            b.add_directional_light(scene.transform_matrix((0, 0, 0), tuple(rng.choice([0.0, 90.0, -90.0, 37.0, -50.0], 3))), tuple(float(x) for x in rng.uniform(0.5, 3, 3)))
    return b.finish()


def bit_equal(a, b):
    a = np.ascontiguousarray(a, np.float32).ravel()
    b = np.ascontiguousarray(b, np.float32).ravel()
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def run(n_scenes, seed, ctx, verbose=True):
    """Returns the number of scenes with a mismatch."""
    rng = np.random.default_rng(seed)
    w, h = 64, 36
    bad = 0
    for k in range(n_scenes):
        sd = random_render_scene(rng)
        sky_rgb = rng.uniform(0, 2, (16, 32, 3)).astype(np.float32) ** 4
        if rng.random() < 0.3:
            sky_rgb[:] = 0
            sky_rgb[2:12, 16] = 50.0      # light samples with an exactly zero z component
        sky = scene.Sky(sky_rgb)
        lo, hi = sd.world_box()
        centre = (np.asarray(lo, np.float64) + np.asarray(hi, np.float64)) / 2
        away = rng.normal(0, 1, 3)
        cam_pos = centre + away / np.linalg.norm(away) * float(np.linalg.norm(np.asarray(hi) - np.asarray(lo))) * float(rng.uniform(0.05, 0.8))
        to = (centre - cam_pos) / max(np.linalg.norm(centre - cam_pos), 1e-9)
        yaw = float(np.degrees(np.arctan2(-to[0], -to[2])))
        pitch = float(np.degrees(np.arcsin(np.clip(to[1], -1, 1))))
        v2w = scene.transform_matrix(tuple(cam_pos), (pitch, yaw, 0.0), (1, 1, 1))
        depth = int(rng.choice([0, 1, 2, 4, 7, 12]))
        jit = scene.mt19937_jitters(3)
        prms = [scene.make_params(w, h, v2w, sky, spp=s, frame=1, jitter=jit[s - 1], max_depth=depth, light_count=len(sd.lights)) for s in (1, 2, 3)]
        osc = oracle.OracleScene(sd, sky)
        ref = None
        for prm in prms:
            ref, cnt = osc.render(prm, ref)
        sb = device.SceneBuffers(ctx, sd, sky)
        fails = []
        for traversal in (0, 1, 2, 3):
            # single draws; the three samples as one draw with the scene laid out ahead (camera rays as tiles, shaded where they
            # are traced -- with the tail launch off also the groups of rays that leave a tile together); one sample, then two
            # in one draw
            for pipeline, tail, how in ((3, -1, "single"), (1, 1 << 30, "single"), (1, 0, "single"), (1, -1, "batch"), (1, 0, "batch"), (3, -1, "1+2")):
                p = device.PathTracingPass(ctx, w, h)
                p.bind_scene(sb)
                p.set_traversal(traversal)
                p.set_pipeline(pipeline)
                p.set_tail_threshold(tail)
                if how == "single":
                    for prm in prms:
                        p.set_params(prm)
                        p.draw()
                        if pipeline == 1:
                            ctx.synchronize()
                elif how == "batch":
                    p.prepare()
                    p.draw_batch(prms)
                else:
                    p.draw_batch(prms[:1])
                    p.draw_batch(prms[1:])
                out = p.download()
                wrong = [key for key in ref if not bit_equal(out[key], ref[key])]
                if wrong:
                    fails.append((traversal, pipeline, tail, how, wrong))
                p.destroy()
        sb.release()
        covered = float((np.asarray(ref["instanceIndex"]) >= 0).mean())
        if verbose or fails:
            print(f"scene {k:3d} inst={len(sd.instances)} tris={len(sd.triangles)} lights={len(sd.lights)} depth={depth} covered={covered:.2f} "
                  f"{'ok' if not fails else 'MISMATCH ' + str(fails)}", flush=True)
        bad += bool(fails)
    return bad


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = run(n_scenes, seed, device.Context(0))
    print(f"{n_scenes} scenes, {bad} with mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
