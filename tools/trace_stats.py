import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from raytracing_amd import device, scene
sd = scene.casa_scene('/root/repo/assets/casa_obj.glb'); sky = scene.procedural_sky(2048, 1024)
ctx = device.Context(0); sb = device.SceneBuffers(ctx, sd, sky)
print("instances", len(sd.instances), "tris", len(sd.triangles))
p = device.PathTracingPass(ctx, 16, 16); p.bind_scene(sb)
lo, hi = sd.world_box(); n = 1 << 22
rays = scene.random_rays(lo, hi, n, 7); rb = device.Buffer(ctx, rays); hb = device.Buffer(ctx, nbytes=n * device.HIT_DTYPE.itemsize)
p.stats(reset=True); p.intersect(rb, n, hb); st = p.stats(reset=True)
d = st["debug"]; print("isect: per ray lane-steps %.1f inner+instance %.1f leaf %.1f  busy %.2f" % (d[0]/n, d[1]/n, d[2]/n, (d[1]+d[2])/max(d[0],1)))
W, H = 1920, 1080
p2 = device.PathTracingPass(ctx, W, H); p2.bind_scene(sb)
prm = scene.make_params(W, H, scene.casa_camera(), sky, spp=1, frame=1, jitter=(0.5, 0.5), max_depth=8, light_count=len(sd.lights))
p2.set_params(prm); p2.stats(reset=True); p2.draw(); st = p2.stats()
d = st["debug"]; r = st["rays_extend"] + st["rays_shadow"]
print("pt: rays", r, "retraced", st["rays_retraced"], "per ray lane-steps %.1f inner+instance %.1f leaf %.1f  busy %.2f" % (d[0]/r, d[1]/r, d[2]/r, (d[1]+d[2])/max(d[0],1)))
