"""The largest benchmark configuration at its stated size on one device: a 50 M-triangle scene, sixteen 4096x4096 RGBA8
albedo maps (1 GiB), half of the instances rough dielectrics (shader/materials/RoughDielectric.glsl:7-133), MAX_DEPTH 16,
3840x2160, 1 spp.  No oracle run is affordable at that size, so:
  - the ordered traversal and the reference-order traversal (two independent kernels over two node formats) must
    produce the same seven 4K images, word for word;
  - a 480x270 window of the same scene and depth must equal the oracle's, word for word;
  - no ray runs out of traversal stack, the buffers (4.8 GB of triangles, 4.8 GB of nodes) are addressed correctly,
    and the reference's order is not forced on the scene.
About 35 s of host work to make and lay out the scene (10.7 GB of host arrays), then a few seconds of GPU time."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_50m_triangles_4k_textures_dielectrics_depth16(gpu_ctx, small_sky, oracle_mod, capsys):
    from raytracing_amd import device, scene
    t0 = time.time()

    def say(msg):   # (a line now and then: the test is silent for a minute otherwise)
        with capsys.disabled():
            print("[50 M triangles] %s (%.0f s)" % (msg, time.time() - t0), flush=True)
    sd = scene.synthetic_scene(50_000_000, dielectric_fraction=0.5, textures=16, texture_size=4096)
    assert len(sd.triangles) > 49_000_000 and len(sd.textures) == 16 and sd.textures[0].shape == (4096, 4096, 4)
    assert (np.round(sd.materials[sd.material_ptr]) == 2).sum() >= 20
    say("scene made")
    w, h, depth = 3840, 2160, 16
    sb = device.SceneBuffers(gpu_ctx, sd, small_sky)
    prm = scene.make_params(w, h, scene.synthetic_camera(), small_sky, spp=1, max_depth=depth)
    images, stats = [], []
    for mode in (0, 1):
        p = device.PathTracingPass(gpu_ctx, w, h)
        p.bind_scene(sb)
        p.set_traversal(mode)
        p.set_params(prm)
        p.draw()
        images.append(p.download())     # (raises if a ray ran out of stack)
        stats.append(p.stats())
        p.destroy()
        say("4K frame drawn, traversal mode %d" % mode)
    for k in images[0]:
        assert np.array_equal(images[0][k].view(np.uint32), images[1][k].view(np.uint32)), k
    assert stats[0]["reference_order_forced"] == 0 and stats[0]["samples"] == w * h
    assert stats[0]["rays_extend"] == stats[1]["rays_extend"] and stats[0]["rays_shadow"] == stats[1]["rays_shadow"]
    assert stats[0]["rays_extend"] > 2 * w * h and stats[1]["rays_retraced"] == 0
    assert stats[0]["rays_retraced"] < 1e-3 * stats[0]["rays_extend"]
    assert (images[0]["instanceIndex"] >= 0).mean() > 0.5 and np.isfinite(images[0]["albedo"]).all()
    images = None
    # a window the oracle finishes in seconds: same scene, same depth, every output word
    sw, sh = 480, 270
    p = device.PathTracingPass(gpu_ctx, sw, sh)
    p.bind_scene(sb)
    prm = scene.make_params(sw, sh, scene.synthetic_camera(), small_sky, spp=1, max_depth=depth)
    p.set_params(prm)
    p.draw()
    got = p.download()
    ref, cnt = oracle_mod.OracleScene(sd, small_sky).render(prm)
    say("oracle window rendered")
    for k in ref:
        a, b = np.ascontiguousarray(got[k]), np.ascontiguousarray(ref[k])
        assert np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))), k
    st = p.stats()
    assert st["rays_extend"] == cnt["rays_extend"] and st["rays_shadow"] == cnt["rays_shadow"]
    p.destroy()
    sb.release()
