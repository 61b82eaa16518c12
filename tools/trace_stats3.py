"""Traversal steps per ray by level (RT_TRACE_STATS=3 build of the library): mesh-level wide nodes,
scene-level wide nodes, instance entries.  Usage on the GPU box:
   RAYTRACING_AMD_LIB=raytracing_amd/libraytracing_amd.stats3.so python3 tools/trace_stats3.py [synth]"""
import os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from raytracing_amd import device, scene
synth = len(sys.argv) > 1 and sys.argv[1] == "synth"
sd = scene.synthetic_scene(10_000_000) if synth else scene.casa_scene(os.path.join(ROOT, "assets", "casa_obj.glb"))
sky = scene.procedural_sky(512, 256)
ctx = device.Context(0); sb = device.SceneBuffers(ctx, sd, sky)
p = device.PathTracingPass(ctx, 16, 16); p.bind_scene(sb)
lo, hi = sd.world_box(); n = 1 << 22
rays = scene.random_rays(lo, hi, n, 7); rb = device.Buffer(ctx, rays); hb = device.Buffer(ctx, nbytes=n * device.HIT_DTYPE.itemsize)
p.stats(reset=True); p.intersect(rb, n, hb); st = p.stats(reset=True)
d = st["debug"]; print("random rays: per ray mesh nodes %.2f scene nodes %.2f instance entries %.2f" % (d[0]/n, d[1]/n, d[2]/n))
W, H = (3840, 2160) if synth else (1920, 1080)
p2 = device.PathTracingPass(ctx, W, H); p2.bind_scene(sb); p2.set_tail_threshold(0)
cam = scene.synthetic_camera() if synth else scene.casa_camera()
for depth in (0, 8):
    prm = scene.make_params(W, H, cam, sky, spp=1, max_depth=depth)
    p2.set_params(prm); p2.stats(reset=True); p2.draw(); st = p2.stats()
    d = st["debug"]; r = st["rays_extend"] + st["rays_shadow"]
    print("depth %d: rays %d  per ray mesh nodes %.2f scene nodes %.2f instance entries %.2f" % (depth, r, d[0]/r, d[1]/r, d[2]/r))
