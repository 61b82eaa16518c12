"""RasterizaionFor1st (the reference's default, Config.h:66): the first path vertex is rebuilt
from the raster pass's four images (pathtracing.glsl:606-624) instead of a traced primary ray.
The raster pass itself is not part of this library; the images here are made from traced
primary hits (what a rasteriser would have drawn) and from arbitrary smooth content sampled
off texel centres.  GPU against the oracle, every output word."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bit_equal(a, b):
    a = np.ascontiguousarray(a, np.float32).ravel()
    b = np.ascontiguousarray(b, np.float32).ravel()
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def gbuffer_from_traced_hits(sd, osc, prm_traced, w, h):
    """What RasterPass would hand over: distance, (unnormalised) world normal, uv, instance."""
    ref, _ = osc.render(prm_traced)
    depth = ref["depth"].reshape(h, w, 1).copy()                       # 1e6 where nothing is hit
    normal = (ref["normal"].reshape(h, w, 3) * np.float32(2.5)).astype(np.float32)
    inst = ref["instanceIndex"].reshape(h, w, 1).copy()
    miss = inst[..., 0] < 0
    inst[miss] = 1000000.0                                              # the clear colour
    normal[miss] = 1000000.0
    rng = np.random.default_rng(4)
    uv = rng.random((h, w, 2), np.float32)                              # any uv will do: both sides read the same
    uv[miss] = 1000000.0
    return depth, normal, uv, inst, ref


def run_both(gpu_ctx, oracle_mod, sd, sky, v2w, w, h, gb, window=None, **kw):
    from raytracing_amd import device, scene
    sb = device.SceneBuffers(gpu_ctx, sd, sky)
    p = device.PathTracingPass(gpu_ctx, w, h)
    p.bind_scene(sb)
    tex = [device.Texture(gpu_ctx, np.ascontiguousarray(a, np.float32)) for a in gb]
    for name, t in zip(("depthGBufferTexture", "normalGBufferTexture", "uvGBufferTexture", "instanceIndexGBufferTexture"), tex):
        p.bind_texture(name, t)
    prm = scene.make_params(w, h, v2w, sky, raster_first=True, light_count=len(sd.lights), **kw)
    if window:
        prm.WINDOW_W, prm.WINDOW_H = window
    p.set_params(prm)
    p.draw()
    out, st = p.download(), p.stats()
    osc = oracle_mod.OracleScene(sd, sky)
    osc.set_gbuffer(*gb)
    ref, cnt = osc.render(prm)
    p.destroy()
    for t in tex:
        t.release()
    sb.release()
    return out, ref, st, cnt


def test_raster_first_from_traced_hits(gpu_ctx, oracle_mod, casa, small_sky):
    from raytracing_amd import scene
    w, h = 128, 72
    v2w = scene.casa_camera()
    osc = oracle_mod.OracleScene(casa, small_sky)
    traced_prm = scene.make_params(w, h, v2w, small_sky, max_depth=3)
    depth, normal, uv, inst, traced = gbuffer_from_traced_hits(casa, osc, traced_prm, w, h)
    out, ref, st, cnt = run_both(gpu_ctx, oracle_mod, casa, small_sky, v2w, w, h, (depth, normal, uv, inst), max_depth=3)
    for k in ref:
        assert bit_equal(out[k], ref[k]), k
    assert st["rays_extend"] == cnt["rays_extend"] and st["rays_shadow"] == cnt["rays_shadow"]
    assert cnt["rays_extend"] < w * h * 3                       # no primary rays among them
    # Same first hits as the traced render.  The images are sampled with LINEAR filtering at
    # uv = (x + 0.5) / W computed in fp32, a few 1e-6 texels off the texel centre, so a pixel next
    # to an uncovered one (1e6) picks up a little of it, as in the reference: compare away from those.
    hit = (traced["instanceIndex"] >= 0).reshape(h, w)
    inner = hit.copy()
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            inner &= np.roll(np.roll(hit, dy, 0), dx, 1)
    d_r, d_t = ref["depth"].reshape(h, w)[inner], traced["depth"].reshape(h, w)[inner]
    assert inner.sum() > 1000 and np.allclose(d_r, d_t, rtol=1e-4)
    assert np.array_equal(ref["instanceIndex"].reshape(h, w)[hit], traced["instanceIndex"].reshape(h, w)[hit])
    assert np.all(ref["instanceIndex"].reshape(h, w)[~hit] == 1000000.0)   # uncovered pixels keep the clear value (:688)


def test_raster_first_window_scaling_and_bad_indices(gpu_ctx, oracle_mod, small_sky):
    """A window larger than the render target (the images are sampled off texel centres, with
    LINEAR / NEAREST filters), instance indices outside the scene, NaN and negative depths."""
    from raytracing_amd import scene
    sd = scene.synthetic_scene(20_000, seed=5, dielectric_fraction=0.3, lights=1)
    w, h, ww, wh = 96, 54, 150, 80
    rng = np.random.default_rng(8)
    yy, xx = np.mgrid[0:wh, 0:ww].astype(np.float32)
    depth = (30.0 + 10.0 * np.sin(xx / 9.0) * np.cos(yy / 7.0)).astype(np.float32)[..., None]
    depth[rng.random((wh, ww)) < 0.1] = 1000000.0
    depth[3, 5] = np.nan
    depth[4, 6] = -2.0
    normal = rng.standard_normal((wh, ww, 3)).astype(np.float32)
    normal[7, 7] = 0.0                                           # normalize(0) = NaN
    uv = (rng.random((wh, ww, 2)) * 3 - 1).astype(np.float32)
    inst = rng.integers(0, len(sd.instances), (wh, ww)).astype(np.float32)[..., None]
    inst[rng.random((wh, ww)) < 0.05] = 5000.0                   # names no instance
    inst[10, 10] = -3.0
    out, ref, st, cnt = run_both(gpu_ctx, oracle_mod, sd, small_sky, scene.synthetic_camera(), w, h,
                                 (depth, normal, uv, inst), window=(ww, wh), max_depth=2)
    for k in ref:
        assert bit_equal(out[k], ref[k]), k
    assert st["rays_extend"] == cnt["rays_extend"]


def test_raster_first_needs_its_textures(gpu_ctx, casa, small_sky):
    from raytracing_amd import device, scene, _capi
    p = device.PathTracingPass(gpu_ctx, 32, 18)
    sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
    p.bind_scene(sb)
    p.set_params(scene.make_params(32, 18, scene.casa_camera(), small_sky, raster_first=True))
    with pytest.raises(_capi.PtError, match="depthGBufferTexture"):
        p.draw()
    t3 = device.Texture(gpu_ctx, np.zeros((4, 4, 3), np.float32))
    with pytest.raises(_capi.PtError, match="channels"):
        p.bind_texture("uvGBufferTexture", device.Texture(gpu_ctx, np.zeros((4, 4, 1), np.float32)))
    p.bind_texture("normalGBufferTexture", t3)
    p.destroy()
    sb.release()


def test_raster_images_with_rasterization_off(gpu_ctx, oracle_mod, casa, small_sky):
    """RasterizaionFor1st off: the shader traces the first hit but still writes the RASTER depth and looks the albedo
    up through the RASTER instance index (shader/pathtracing.glsl:615-616, :641, :684).  With those two images bound
    the pass does the same -- here with images that disagree with the traced hit: depth scaled and shifted, indices
    rotated, some naming no instance (out of bounds in the reference; the traced index stands in), a window larger
    than the frame -- over two samples and with the tail launch and draws in flight in play; the oracle, which
    tests/test_oracle.py pins on this very behaviour against the reference's shader run as C++, is the judge."""
    from raytracing_amd import device, scene
    w, h = 160, 90
    osc = oracle_mod.OracleScene(casa, small_sky)
    prim, _ = osc.render(scene.make_params(w, h, scene.casa_camera(), small_sky, max_depth=0))
    depth = prim["depth"].reshape(h, w, 1)
    inst = prim["instanceIndex"].reshape(h, w, 1)
    n_inst = len(casa.instances)
    g_depth = np.where(depth <= 100000.0, depth * np.float32(1.25) + np.float32(0.5), depth).astype(np.float32)
    g_inst = ((inst.astype(np.int64) + 3) % n_inst).astype(np.float32)
    g_inst[::7, ::5] = 1000000.0          # names no instance
    g_inst[3::11, 2::9] = -4.0
    sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
    p = device.PathTracingPass(gpu_ctx, w, h)
    p.bind_scene(sb)
    td, ti = device.Texture(gpu_ctx, g_depth), device.Texture(gpu_ctx, g_inst)
    p.bind_texture("depthGBufferTexture", td)
    p.bind_texture("instanceIndexGBufferTexture", ti)
    osc.set_gbuffer(g_depth, np.zeros((h, w, 3), np.float32), np.zeros((h, w, 2), np.float32), g_inst)
    jit = scene.mt19937_jitters(2)
    ref = None
    for s in (1, 2):
        prm = scene.make_params(w, h, scene.casa_camera(), small_sky, spp=s, jitter=jit[s - 1], max_depth=3)
        prm.WINDOW_W, prm.WINDOW_H = w + 32, h + 10
        p.set_params(prm)
        p.draw()
        ref, _ = osc.render(prm, ref)
    got = p.download()
    for k in ref:
        a, b = np.ascontiguousarray(got[k]), np.ascontiguousarray(ref[k])
        assert np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))), k
    # the images really were in play: depth is the raster's, albedo differs from the traced-only frame
    plain = oracle_mod.OracleScene(casa, small_sky)
    prm1 = scene.make_params(w, h, scene.casa_camera(), small_sky, spp=1, jitter=jit[0], max_depth=3)
    base, _ = plain.render(prm1)
    p.set_params(prm1)
    p.draw()
    one = p.download()
    assert not np.array_equal(one["depth"], base["depth"]) and not np.array_equal(one["albedo"], base["albedo"])
    assert np.array_equal(one["instanceIndex"], base["instanceIndex"]) and np.array_equal(one["normal"].view(np.uint32), base["normal"].view(np.uint32))
    p.destroy()
    td.release(); ti.release()
    sb.release()
