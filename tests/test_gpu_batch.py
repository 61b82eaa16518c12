"""pt_pass_draw_batch: the reference's host loop `for spp in 1..SPP` (main.cpp:169-208: new jitter and currentspp,
then Renderer::draw) as one call.  Its outputs must be, word for word, those of the same samples drawn one by one
-- which tests/test_gpu_render.py and tests/test_gpu_golden.py tie to the oracle and to the reference's shader --
and, here too, the oracle's running mean over the same samples."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def words_equal(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def params_of(scene, sd, sky, camera, w, h, spps, depth, **kw):
    jit = scene.mt19937_jitters(max(spps))
    return [scene.make_params(w, h, camera, sky, spp=s, frame=1, jitter=jit[s - 1], max_depth=depth,
                              light_count=len(sd.lights), **kw) for s in spps]


def one_by_one(p, prms):
    for prm in prms:
        p.set_params(prm)
        p.draw()
    return p.download(), p.stats(reset=True)


def batched(p, prms, sizes):
    k = 0
    for n in sizes:
        p.draw_batch(prms[k:k + n])
        k += n
    assert k == len(prms)
    return p.download(), p.stats(reset=True)


@pytest.mark.parametrize("pipeline,tail", [(1, -1), (1, 0), (3, -1)])
def test_batches_equal_single_draws_and_the_oracle(gpu_ctx, casa, small_sky, oracle_mod, pipeline, tail):
    from raytracing_amd import device, scene
    w, h, depth = 160, 90, 5
    sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
    p = device.PathTracingPass(gpu_ctx, w, h)
    p.bind_scene(sb)
    p.set_pipeline(pipeline)
    p.set_tail_threshold(tail)
    prms = params_of(scene, casa, small_sky, scene.casa_camera(), w, h, list(range(1, 12)), depth)
    p.stats(reset=True)
    ref, st_ref = one_by_one(p, prms)
    # 11 samples as 4 + 1 + 6 (a batch, a batch of one, a larger batch: the path state grows), then as 3 + 8, then 11 x 1
    for sizes in ([4, 1, 6], [3, 8], [1] * 11):
        got, st = batched(p, prms, sizes)
        for k in ref:
            assert words_equal(got[k], ref[k]), (sizes, k)
        for k in ("samples", "draws", "rays_extend", "rays_shadow", "rays_shadow_null"):
            assert st[k] == st_ref[k], (sizes, k)
    # and the oracle's running mean over the same samples
    osc = oracle_mod.OracleScene(casa, small_sky)
    acc = None
    for prm in prms[:5]:
        acc, _ = osc.render(prm, acc)
    got, _ = batched(p, prms[:5], [5])
    for k in acc:
        assert words_equal(got[k], acc[k].reshape(got[k].shape)), k
    p.destroy()
    sb.release()


def test_batch_with_lights_dielectrics_and_sums(gpu_ctx, small_sky):
    """Analytic lights (the launches that trace occlusion queries apart from the extend rays), rough dielectrics and
    albedo maps; accum_mode 1 (sums, as the multi-GPU slices keep them); a frame counter other than 1."""
    from raytracing_amd import device, scene
    sd = scene.synthetic_scene(20_000, seed=3, dielectric_fraction=0.5, lights=2, textures=2, texture_size=64)
    w, h = 128, 72
    sb = device.SceneBuffers(gpu_ctx, sd, small_sky)
    p = device.PathTracingPass(gpu_ctx, w, h)
    p.bind_scene(sb)
    jit = scene.mt19937_jitters(6)
    for accum_mode in (0, 1):
        prms = [scene.make_params(w, h, scene.synthetic_camera(), small_sky, spp=s, frame=3, jitter=jit[s - 1], max_depth=6,
                                  light_count=len(sd.lights), accum_mode=accum_mode) for s in range(1, 7)]
        ref, _ = one_by_one(p, prms)
        got, _ = batched(p, prms, [6])
        for k in ref:
            assert words_equal(got[k], ref[k]), (accum_mode, k)
        got, _ = batched(p, prms, [2, 4])
        for k in ref:
            assert words_equal(got[k], ref[k]), (accum_mode, k)
    p.destroy()
    sb.release()


def test_batch_with_raster_first_inputs(gpu_ctx, casa, small_sky, oracle_mod):
    from raytracing_amd import device, scene
    from test_gpu_raster_first import gbuffer_from_traced_hits
    w, h = 96, 54
    osc = oracle_mod.OracleScene(casa, small_sky)
    gb = gbuffer_from_traced_hits(casa, osc, scene.make_params(w, h, scene.casa_camera(), small_sky, max_depth=2), w, h)[:4]
    sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
    p = device.PathTracingPass(gpu_ctx, w, h)
    p.bind_scene(sb)
    tex = [device.Texture(gpu_ctx, np.ascontiguousarray(a, np.float32)) for a in gb]
    for name, t in zip(("depthGBufferTexture", "normalGBufferTexture", "uvGBufferTexture", "instanceIndexGBufferTexture"), tex):
        p.bind_texture(name, t)
    jit = scene.mt19937_jitters(4)
    prms = [scene.make_params(w, h, scene.casa_camera(), small_sky, spp=s, jitter=jit[s - 1], max_depth=3, raster_first=True)
            for s in range(1, 5)]
    ref, _ = one_by_one(p, prms)
    got, _ = batched(p, prms, [4])
    for k in ref:
        assert words_equal(got[k], ref[k]), k
    p.destroy()
    for t in tex:
        t.release()
    sb.release()


def test_batch_arguments_are_checked(gpu_ctx, casa, small_sky):
    from raytracing_amd import device, scene, _capi
    w, h = 32, 18
    sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
    p = device.PathTracingPass(gpu_ctx, w, h)
    p.bind_scene(sb)
    prm = scene.make_params(w, h, scene.casa_camera(), small_sky, spp=1, max_depth=2)
    with pytest.raises(_capi.PtError, match="1..16"):
        p.draw_batch([prm] * 17)
    with pytest.raises(_capi.PtError, match="1..16"):
        _capi.check(_capi.lib.pt_pass_draw_batch(p.h, (_capi.PtParams * 1)(prm), 0))
    other = scene.make_params(w, h, scene.casa_camera(), small_sky, spp=2, max_depth=3)
    with pytest.raises(_capi.PtError, match="jitter, currentspp and frameCounter only"):
        p.draw_batch([prm, other])
    bad = scene.make_params(w + 1, h, scene.casa_camera(), small_sky, spp=1, max_depth=2)
    with pytest.raises(_capi.PtError, match="SCREEN_W"):
        p.draw_batch([bad])
    with pytest.raises(_capi.PtError, match="1..16"):
        p.reserve_batch(0)
    p.reserve_batch(4)
    p.prepare()
    p.draw_batch([prm] * 4)
    gpu_ctx.synchronize()
    p.destroy()
    sb.release()


@pytest.mark.parametrize("lights", [0, 2])
def test_camera_rays_and_first_vertex_in_one_launch(gpu_ctx, casa, small_sky, oracle_mod, lights):
    """With the scene laid out ahead of the first draw (pt_pass_prepare) the camera rays walk the flattened tree as tiles and
    their hits are shaded in the same launch; with per-kernel profiling on, every stage is a launch of its own.  Both must
    give the oracle's outputs and the same ray counts -- with analytic lights too (several shadow entries per vertex)."""
    from raytracing_amd import device, scene
    if lights:
        sd = scene.synthetic_scene(20_000, seed=5, dielectric_fraction=0.3, lights=lights, textures=2, texture_size=64)
        camera = scene.synthetic_camera()
    else:
        sd, camera = casa, scene.casa_camera()
    w, h, depth = 136, 76, 4   # (edge tiles: 136 = 17 x 8, 76 = 9.5 x 8)
    sb = device.SceneBuffers(gpu_ctx, sd, small_sky)
    p = device.PathTracingPass(gpu_ctx, w, h)
    p.bind_scene(sb)
    if lights:
        p.set_traversal(3)   # (its instances are rotated: the flattened tree is built only when asked for)
    p.prepare()
    prms = params_of(scene, sd, small_sky, camera, w, h, [1, 2, 3, 4, 5], depth)
    osc = oracle_mod.OracleScene(sd, small_sky)
    acc = None
    for prm in prms:
        acc, _ = osc.render(prm, acc)
    p.stats(reset=True)
    got, st = batched(p, prms, [5])
    assert st["flat_leaves"] > 0
    for k in acc:
        assert words_equal(got[k], acc[k].reshape(got[k].shape)), ("one launch", k)
    p.set_profiling(True)
    got2, st2 = batched(p, prms, [2, 3])
    p.set_profiling(False)
    for k in acc:
        assert words_equal(got2[k], acc[k].reshape(got2[k].shape)), ("stages apart", k)
    for k in ("samples", "rays_extend", "rays_shadow", "rays_shadow_null", "rays_retraced"):
        assert st[k] == st2[k], k
    p.destroy()
    sb.release()


def test_rays_that_leave_a_tile_together_are_walked_together(gpu_ctx, casa, oracle_mod):
    """Vertex 1 of the demo scene: the tiles on its polished surfaces (roughness 0.001) send their paths on in nearly one
    direction; those are listed as groups and walked one wave per group -- and so are the light samples of a tile that
    point at the sun of the benchmark's sky (at 2048x1024 its bright texels lie within a few degrees; the small sky of
    the other tests spreads them too far).  Outputs and ray counts must not know."""
    from raytracing_amd import device, scene
    import os
    small_sky = scene.procedural_sky(2048, 1024)
    w, h, depth = 320, 180, 5
    os.environ["RTA_BOUNCE_GROUPS"] = "2"   # (the library makes groups only in draws of millions of paths)
    try:
        sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
        p = device.PathTracingPass(gpu_ctx, w, h)
        p.bind_scene(sb)
        p.set_tail_threshold(0)   # every vertex in launches of its own (a small frame would hand vertex 1 to the tail launch)
        p.prepare()
        prms = params_of(scene, casa, small_sky, scene.casa_camera(), w, h, [1, 2, 3, 4], depth)
        osc = oracle_mod.OracleScene(casa, small_sky)
        acc, cnt = None, dict(rays_extend=0, rays_shadow=0)
        for prm in prms:
            acc, c = osc.render(prm, acc)
            for k in cnt:
                cnt[k] += c[k]
        p.stats(reset=True)
        got, st = batched(p, prms, [4])
        assert st["rays_extend_grouped"] > 0.1 * w * h * 4, st   # (a third of this view's first hits lie on such surfaces)
        assert st["rays_shadow_grouped"] > 0.2 * w * h * 4, st
        assert st["rays_extend"] == cnt["rays_extend"] and st["rays_shadow"] == cnt["rays_shadow"]
        for k in acc:
            assert words_equal(got[k], acc[k].reshape(got[k].shape)), k
        # single draws and the per-kernel profile take other routes to the same frame
        got1, st1 = one_by_one(p, prms)
        p.set_profiling(True)
        got2, st2 = batched(p, prms, [4])
        p.set_profiling(False)
        assert st2["rays_extend_grouped"] == 0 and st2["rays_shadow_grouped"] == 0
        for k in acc:
            assert words_equal(got1[k], acc[k].reshape(got1[k].shape)) and words_equal(got2[k], acc[k].reshape(got2[k].shape)), k
        for k in ("rays_extend", "rays_shadow", "rays_shadow_null"):
            assert st[k] == st1[k] == st2[k], k
    finally:
        del os.environ["RTA_BOUNCE_GROUPS"]
    p.destroy()
    sb.release()
