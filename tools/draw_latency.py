"""Latency of one draw the caller waits for, and cost per draw of a burst, with three and with one draw in flight."""
import sys, time
sys.path.insert(0, '/root/repo')
from raytracing_amd import device, scene
sd = scene.casa_scene('/root/repo/assets/casa_obj.glb'); sky = scene.procedural_sky(2048, 1024)
ctx = device.Context(0); sb = device.SceneBuffers(ctx, sd, sky)
W, H = 1920, 1080
p = device.PathTracingPass(ctx, W, H); p.bind_scene(sb)
prm = scene.make_params(W, H, scene.casa_camera(), sky, spp=1, frame=1, jitter=(0.5, 0.5), max_depth=8, light_count=0)
p.set_params(prm)
for lanes in (3, 1):
    p.set_pipeline(lanes)
    for k in range(3): p.draw(); ctx.synchronize()
    t = time.perf_counter()
    for k in range(20): p.draw(); ctx.synchronize()
    print("lanes", lanes, "draw + wait: %.3f ms" % ((time.perf_counter() - t) / 20 * 1e3))
    t = time.perf_counter()
    for k in range(60): p.draw()
    ctx.synchronize()
    print("lanes", lanes, "burst of 60: %.3f ms per draw" % ((time.perf_counter() - t) / 60 * 1e3))
