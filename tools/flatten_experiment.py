"""Experiment, GPU box: how much of the traversal cost is the two-level structure?  The casa scene as it is
(19 instances over 19 meshes) against the same triangles moved to world space in ONE mesh under an identity
instance (results are then not the reference's bit for bit -- this only bounds what restructuring the scene
level could win).  Prints Mrays/s of pt_pass_intersect on random rays and of path-traced frames."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import device, scene  # noqa: E402


def flattened(sd):
    out = []
    base = 0
    for i, inst in enumerate(sd.instances):
        m = int(sd.mesh_index[i])
        n = int(sd.tri_count[m])
        first = int(np.sum(sd.tri_count[:m]))
        t = sd.triangles[first:first + n].copy()
        M = sd.l2w[i].reshape(4, 4).T.astype(np.float64)     # rows of l2w = columns of the matrix
        P = t[:, :9].reshape(-1, 3).astype(np.float64)
        t[:, :9] = (P @ M[:3, :3].T + M[:3, 3]).astype(np.float32).reshape(-1, 9)
        N = t[:, 9:18].reshape(-1, 3).astype(np.float64) @ np.linalg.inv(M[:3, :3])
        N /= np.maximum(np.linalg.norm(N, axis=1, keepdims=True), 1e-20)
        t[:, 9:18] = N.astype(np.float32).reshape(-1, 9)
        out.append(t)
    b = scene.SceneBuilder()
    for tex in sd.textures:
        b.add_texture(tex)
    m = b.add_mesh(np.concatenate(out, 0))
    b.add_instance(m, scene.transform_matrix(), dict(type=1, albedo=(0.8, 0.8, 0.8), roughness=np.float32(0.3), metallic=np.float32(1.0)))
    return b.finish(0)


def run(ctx, sd, sky, tag, n_rays=1 << 24):
    sb = device.SceneBuffers(ctx, sd, sky)
    p = device.PathTracingPass(ctx, 1920, 1080)
    p.bind_scene(sb)
    lo, hi = sd.world_box()
    rays = scene.random_rays(lo, hi, n_rays, 7)
    rb = device.Buffer(ctx, rays)
    hb = device.Buffer(ctx, nbytes=n_rays * device.HIT_DTYPE.itemsize)
    for _ in range(2):
        p.intersect(rb, n_rays, hb)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        p.intersect(rb, n_rays, hb)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    hits = np.frombuffer(hb.download(), device.HIT_DTYPE)
    print("%-10s intersect %8.1f Mrays/s   hit fraction %.3f" % (tag, 6 * n_rays / dt / 1e6, float((hits["triangle"] >= 0).mean())), flush=True)
    jit = scene.mt19937_jitters(40)
    def draw(s):
        p.set_params(scene.make_params(1920, 1080, scene.casa_camera(), sky, spp=s, frame=1, jitter=jit[s - 1], max_depth=8))
        p.draw()
    for s in range(1, 9):
        draw(s)
    ctx.synchronize()
    st0 = p.stats()
    t0 = time.perf_counter()
    for s in range(9, 41):
        draw(s)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    st = p.stats()
    rays_n = st["rays_extend"] + st["rays_shadow"] - st0["rays_extend"] - st0["rays_shadow"]
    print("%-10s path trace %7.1f Mrays/s   %.3f ms per draw, %.2f rays per sample" % (tag, rays_n / dt / 1e6, dt / 32 * 1e3, rays_n / 32 / (1920 * 1080)), flush=True)
    p.destroy()
    rb.release(); hb.release()
    sb.release()


def main():
    sd = scene.casa_scene(os.path.join(ROOT, "assets", "casa_obj.glb"))
    sky = scene.procedural_sky(2048, 1024)
    ctx = device.Context(0)
    run(ctx, sd, sky, "two-level")
    run(ctx, flattened(sd), sky, "one mesh")


if __name__ == "__main__":
    main()
