import sys
sys.path.insert(0, '/root/repo')
from raytracing_amd import device, scene
sd = scene.casa_scene('/root/repo/assets/casa_obj.glb'); sky = scene.procedural_sky(2048, 1024)
ctx = device.Context(0); sb = device.SceneBuffers(ctx, sd, sky)
W, H = 1920, 1080
p2 = device.PathTracingPass(ctx, W, H); p2.bind_scene(sb)
prm = scene.make_params(W, H, scene.casa_camera(), sky, spp=1, frame=1, jitter=(0.5, 0.5), max_depth=8, light_count=len(sd.lights))
p2.set_params(prm); p2.stats(reset=True); p2.draw(); st = p2.stats()
d = st["debug"]; r = st["rays_extend"] + st["rays_shadow"]
print("wave iterations", d[0], "per ray*64 %.1f" % (d[0] * 64 / r), "leaf code ran in %.2f of them, instance code in %.2f" % (d[1] / d[0], d[2] / d[0]))
