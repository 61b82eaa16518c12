"""Full-size checks (the sizes of the benchmark configurations) through properties that need no oracle run:
the ordered traversal and the reference-order traversal are two independent kernels and must
return the same hits for all 16 Mi rays of the intersection microbench, and the same seven
images for a 1920x1080 depth-8 frame, whatever the number of draws in flight; a bounded sample
is also compared with the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_16mi_rays_ordered_equals_reference_order(gpu_ctx, casa, oracle_mod):
    from raytracing_amd import device, scene
    n = 1 << 24
    lo, hi = casa.world_box()
    rays = scene.random_rays(lo, hi, n, 7)
    sb = device.SceneBuffers(gpu_ctx, casa)
    p = device.PathTracingPass(gpu_ctx, 16, 16)
    p.bind_scene(sb)
    rb = device.Buffer(gpu_ctx, rays)
    hb = device.Buffer(gpu_ctx, nbytes=n * device.HIT_DTYPE.itemsize)
    got = {}
    for mode in (0, 1):
        p.set_traversal(mode)
        p.intersect(rb, n, hb)
        got[mode] = hb.download(np.uint8).view(device.HIT_DTYPE).copy()
    assert np.array_equal(got[0].view(np.uint8), got[1].view(np.uint8))
    hit = got[0]["triangle"] >= 0
    assert 0.1 < hit.mean() < 0.9
    assert np.all(got[0]["t"][hit] >= np.float32(0.001)) and np.all(got[0]["t"][~hit] == np.float32(1e18))
    u, v = got[0]["u"][hit], got[0]["v"][hit]
    assert np.all(u >= 0) and np.all(v >= 0) and np.all(u + v <= 1)
    ref, _ = oracle_mod.OracleScene(casa).intersect(rays[-(1 << 18):])          # a bounded sample, from the far end
    assert np.array_equal(got[0][-(1 << 18):].view(np.uint8), ref.view(np.uint8))
    rb.release(); hb.release(); p.destroy(); sb.release()


def test_1080p_frame_same_in_both_modes_and_any_pipeline_depth(gpu_ctx, casa, small_sky):
    from raytracing_amd import device, scene
    w, h, depth, spp = 1920, 1080, 8, 4
    sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
    jit = scene.mt19937_jitters(spp)
    images = []
    for mode, lanes in ((1, 1), (0, 1), (0, 6)):
        p = device.PathTracingPass(gpu_ctx, w, h)
        p.bind_scene(sb)
        p.set_traversal(mode)
        p.set_pipeline(lanes)
        for s in range(1, spp + 1):
            p.set_params(scene.make_params(w, h, scene.casa_camera(), small_sky, spp=s, jitter=jit[s - 1], max_depth=depth))
            p.draw()
        images.append(p.download())
        st = p.stats()
        assert st["samples"] == w * h * spp
        p.destroy()
    for other in images[1:]:
        for k in images[0]:
            a, b = images[0][k].view(np.uint32), other[k].view(np.uint32)
            assert np.array_equal(a, b), k
    assert np.isfinite(images[0]["directLum"]).all()
    sb.release()
