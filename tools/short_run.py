"""GPU box: where does the fixed cost of a short timed region go?  Times K pipelined casa draws
after W warm-up draws for several (W, K), each bracketed by synchronisation, in one process."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("RTA_HW_QUEUES", "16")   # (the box exports 4)
from raytracing_amd import device, scene  # noqa: E402


def main():
    sd = scene.casa_scene(os.path.join(ROOT, "assets", "casa_obj.glb"))
    sky = scene.procedural_sky(2048, 1024)
    ctx = device.Context(0)
    sb = device.SceneBuffers(ctx, sd, sky)
    p = device.PathTracingPass(ctx, 1920, 1080)
    p.bind_scene(sb)
    p.set_pipeline(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
    p.prepare()
    jit = scene.mt19937_jitters(4096)
    s = [0]

    def draw():
        s[0] += 1
        p.set_params(scene.make_params(1920, 1080, scene.casa_camera(), sky, spp=s[0], frame=1, jitter=jit[s[0] % 4096], max_depth=8))
        p.draw()

    for W, K in ((5, 20), (5, 20), (5, 20), (5, 64), (5, 64), (5, 10)):
        for _ in range(W):
            draw()
        ctx.synchronize()
        t0 = time.perf_counter()
        tq = []
        for _ in range(K):
            draw()
            tq.append(time.perf_counter() - t0)
        t_queued = time.perf_counter() - t0
        ctx.synchronize()
        dt = time.perf_counter() - t0
        print("W %3d K %3d: %.3f ms/step  total %.2f ms, host had queued everything after %.2f ms (draw 6 returned at %.2f ms)"
              % (W, K, dt / K * 1e3, dt * 1e3, t_queued * 1e3, tq[min(5, K - 1)] * 1e3), flush=True)


if __name__ == "__main__":
    main()
