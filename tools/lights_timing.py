"""GPU box: a scene lit by the sky and three analytic lights (their occlusion queries run in the reference's order):
time per sample and ray counts.  Usage: python tools/lights_timing.py [triangles] [draws]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("RTA_HW_QUEUES", "16")
from raytracing_amd import device, scene  # noqa: E402


def main():
    tris = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
    draws = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    sd = scene.synthetic_scene(tris, seed=5, dielectric_fraction=0.3, lights=3, textures=4, texture_size=256)
    sky = scene.procedural_sky(1024, 512)
    w, h = 1920, 1080
    ctx = device.Context(0)
    sb = device.SceneBuffers(ctx, sd, sky)
    p = device.PathTracingPass(ctx, w, h)
    p.bind_scene(sb)
    p.set_pipeline(6)
    jit = scene.mt19937_jitters(draws + 4)

    def draw(s):
        p.set_params(scene.make_params(w, h, scene.synthetic_camera(), sky, spp=s, frame=1, jitter=jit[s - 1], max_depth=6, light_count=3))
        p.draw()

    for s in range(1, 5):
        draw(s)
    ctx.synchronize()
    p.stats(reset=True)
    t0 = time.perf_counter()
    for s in range(5, 5 + draws):
        draw(s)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    st = p.stats()
    print("%d triangles, 3 lights: %.3f ms per sample; per sample %.2f extend rays, %.2f occlusion queries of which %.2f not traced"
          % (len(sd.triangles), dt / draws * 1e3, st["rays_extend"] / st["samples"], st["rays_shadow"] / st["samples"],
             st["rays_shadow_null"] / st["samples"]), flush=True)


if __name__ == "__main__":
    main()
