import sys, time, os
os.environ["GPU_MAX_HW_QUEUES"] = "16"
sys.path.insert(0, '/root/repo')
from raytracing_amd import device, scene
sd = scene.casa_scene('/root/repo/assets/casa_obj.glb'); sky = scene.procedural_sky(2048, 1024)
ctx = device.Context(0); sb = device.SceneBuffers(ctx, sd, sky)
for (W, H, depth) in ((640, 480, 2), (640, 480, 8), (1920, 1080, 8)):
    p = device.PathTracingPass(ctx, W, H); p.bind_scene(sb)
    for lanes in (1, 6):
        p.set_pipeline(lanes)
        jit = scene.mt19937_jitters(200)
        prms = [scene.make_params(W, H, scene.casa_camera(), sky, spp=s + 1, jitter=jit[s], max_depth=depth) for s in range(200)]
        for s in range(8):
            p.set_params(prms[s]); p.draw()
        ctx.synchronize()
        t0 = time.perf_counter(); tc = 0.0
        for s in range(200):
            p.set_params(prms[s])
            a = time.perf_counter(); p.draw(); tc += time.perf_counter() - a
        ctx.synchronize()
        dt = time.perf_counter() - t0
        print("%dx%d depth %d lanes %d: %.3f ms per draw wall, %.3f ms host time inside pt_pass_draw" % (W, H, depth, lanes, dt / 200 * 1e3, tc / 200 * 1e3))
    p.destroy()
