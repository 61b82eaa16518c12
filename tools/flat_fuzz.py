#!/usr/bin/env python3
"""Differential test of the three traversal modes on the GPU against the CPU oracle: random scenes (meshes of random soups and grids; instances
translated up to 1e4, scaled 1e-3..1e3, unevenly scaled, mirrored, axis-permuted, and -- forced -- rotated),
rays aimed at the instances from near and from far (|o| up to 1e5), with direction components that are zero,
denormal-small or huge, and grazing rays inside triangle planes.  Hits of traversal modes 1 (the reference's order), 2 (ordered, two levels) and 3
(flattened tree) must equal the oracle's bit for bit.
Usage: python tools/flat_fuzz.py [n_scenes] [rays_per_instance] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import device, scene  # noqa: E402
from oracle import oracle  # noqa: E402


def soup(rng, n, scale):
    t = np.zeros((n, 24), np.float32)
    c = rng.uniform(-1, 1, (n, 1, 3)) * scale
    t[:, :9] = (c + rng.normal(0, 0.08 * scale, (n, 3, 3))).reshape(n, 9)
    t[:, 9:18] = np.tile([0, 0, 1], 3)
    return t


def grid(rng, k, scale):
    u = np.linspace(-1, 1, k + 1) * scale
    x, z = np.meshgrid(u, u, indexing="ij")
    y = 0.1 * scale * np.sin(3 * x / scale) * np.cos(2 * z / scale) if rng.random() < 0.7 else np.zeros_like(x)   # some exactly flat
    return scene._grid_mesh(np.stack([x, y, z], -1), k, k, 1.0)


def random_scene(rng, rotated):
    b = scene.SceneBuilder()
    meshes = []
    for _ in range(int(rng.integers(2, 6))):
        s = float(10.0 ** rng.uniform(-2, 2))
        meshes.append(b.add_mesh(soup(rng, int(rng.integers(1, 400)), s) if rng.random() < 0.5 else grid(rng, int(rng.integers(1, 14)), s)))
    for _ in range(int(rng.integers(2, 10))):
        m = meshes[int(rng.integers(0, len(meshes)))]
        pos = rng.uniform(-1, 1, 3) * float(10.0 ** rng.uniform(-1, 4))
        if rotated:
            rot = rng.uniform(0, 360, 3)
        else:
            rot = rng.integers(0, 4, 3) * 90.0      # axis permutations (cos 90 degrees is 4e-8 in fp32, not 0)
        sc = np.full(3, 10.0 ** rng.uniform(-3, 3))
        if rng.random() < 0.4:
            sc = sc * 10.0 ** rng.uniform(-0.7, 0.7, 3)   # uneven
        if rng.random() < 0.3:
            sc[int(rng.integers(0, 3))] *= -1.0           # mirrored
        mat = scene.transform_matrix(tuple(pos), tuple(rot), tuple(sc))
        b.add_instance(m, mat)
        if rng.random() < 0.25:
            b.add_instance(m, mat)          # an exact duplicate: every hit ties with its twin in another instance
    return b.finish()


def rays_for(sd, rng, per_instance):
    rays = []
    for k in range(len(sd.instances)):
        leaf = np.flatnonzero(sd.scene_nodes["inst"] == k)[0]
        lo, hi = sd.scene_nodes["aa"][leaf][:3].astype(np.float64), sd.scene_nodes["bb"][leaf][:3].astype(np.float64)
        ext = np.maximum(hi - lo, 1e-30)
        n = per_instance
        target = lo + rng.uniform(0, 1, (n, 3)) * ext
        far = 10.0 ** rng.uniform(-1, 4, (n, 1))
        origin = (lo + hi) / 2 + rng.normal(0, 1, (n, 3)) * ext * far
        d = target - origin
        d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-300)
        # direction lengths from 1e-6 to 1e6 (the reference does not normalise what it is given)
        d *= 10.0 ** rng.uniform(-6, 6, (n, 1)) * (rng.random((n, 1)) < 0.2) + (rng.random((n, 1)) >= 0.2)
        # axis-parallel rays through the box, zero / tiny components
        m = rng.random(n) < 0.15
        ax = rng.integers(0, 3, n)
        for a in range(3):
            sel = m & (ax == a)
            dd = np.zeros((int(sel.sum()), 3))
            dd[:, a] = rng.choice([-1.0, 1.0], int(sel.sum()))
            tiny = rng.choice([0.0, 1e-30, -1e-30, 1e-42, 1e-12], (int(sel.sum()), 3))
            tiny[:, a] = 0.0
            d[sel] = dd + tiny
            origin[sel] = target[sel] - dd * ext[a] * 2.0
        rays.append(np.concatenate([origin, d], 1))
        # rays through vertices and edge midpoints of the instance's triangles (shared by neighbours in a grid: ties)
        mi = int(sd.mesh_index[k])
        base = int(np.sum(sd.tri_count[:mi]))
        tri = sd.triangles[base + rng.integers(0, int(sd.tri_count[mi]), per_instance // 4)]
        ver = np.asarray(tri["ver"] if tri.dtype.names else tri[:, :9], np.float64).reshape(-1, 3, 3)
        wgt = rng.choice([[1, 0, 0], [0, 1, 0], [0.5, 0.5, 0], [0, 0.5, 0.5], [1 / 3, 1 / 3, 1 / 3]], len(ver))
        local = np.einsum("nk,nkc->nc", np.asarray(wgt, np.float64), ver)
        m4 = np.asarray(sd.l2w[k], np.float64).reshape(4, 4)        # column major: m4[c][r]
        world = local @ m4[:3, :3] + m4[3, :3]
        org = world + rng.normal(0, 1, world.shape) * ext * 10.0 ** rng.uniform(-1, 2, (len(world), 1))
        dd = world - org
        dd /= np.maximum(np.linalg.norm(dd, axis=1, keepdims=True), 1e-300)
        rays.append(np.concatenate([org, dd], 1))
    return np.ascontiguousarray(np.concatenate(rays), np.float32)


def same(a, b):
    h = b["triangle"] >= 0
    return (np.array_equal(a["triangle"], b["triangle"]) and np.array_equal(a["instance"], b["instance"])
            and np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32))
            and np.array_equal(a["u"][h].view(np.uint32), b["u"][h].view(np.uint32))
            and np.array_equal(a["v"][h].view(np.uint32), b["v"][h].view(np.uint32)))


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rng = np.random.default_rng(seed)
    ctx = device.Context(0)
    bad = flat_built = 0
    for k in range(n_scenes):
        rotated = k % 3 == 2
        sd = random_scene(rng, rotated)
        rays = rays_for(sd, rng, per)
        sb = device.SceneBuffers(ctx, sd)
        out = {}
        for mode in (1, 2, 3):
            p = device.PathTracingPass(ctx, 16, 16)
            p.bind_scene(sb)
            p.set_traversal(mode)
            out[mode] = device.trace_rays(p, rays)
            st = p.stats()
            p.destroy()
        sb.release()
        truth, _ = oracle.OracleScene(sd).intersect(rays)
        ok1 = same(out[1], truth)
        out[1] = truth
        ok = same(out[3], out[1])
        ok2 = same(out[2], out[1])
        flat_built += st["flat_leaves"] > 0
        print(f"scene {k:3d} rotated={int(rotated)} inst={len(sd.instances)} tris={len(sd.triangles)} rays={len(rays)} hits={float((out[1]['triangle'] >= 0).mean()):.3f} "
              f"flat_leaves={st['flat_leaves']} retraced={st['rays_retraced']} {'ok' if ok else 'MISMATCH'} two-level {'ok' if ok2 else 'MISMATCH'} reference-order {'ok' if ok1 else 'MISMATCH'}", flush=True)
        for which, good in ((3, ok), (2, ok2)):
            if good:
                continue
            bad += 1
            a, b = out[which], out[1]
            print("    mode", which)
            w = np.flatnonzero((a["triangle"] != b["triangle"]) | (a["instance"] != b["instance"]) | (a["t"].view(np.uint32) != b["t"].view(np.uint32)))
            np.set_printoptions(precision=9, linewidth=200)
            print("   ", len(w), "rays differ; first:", rays[w[:3]].tolist(), a[w[:3]], b[w[:3]])
            if os.environ.get("FUZZ_DUMP"):
                np.savez(os.path.join(ROOT, "gpurun_out", f"fuzz_scene_{k}_mode{which}.npz"), rays=rays[w], got=a[w], want=b[w], triangles=sd.triangles,
                         tri_count=sd.tri_count, mesh_index=sd.mesh_index, l2w=sd.l2w)
    print(f"{n_scenes} scenes, {flat_built} with a flat tree, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
