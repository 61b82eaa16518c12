"""The launch that finishes the thin tail of a draw (pt_pass_set_tail_threshold): whatever vertex it
takes over from, the seven outputs and the ray counts are those of the vertex-by-vertex chain."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render(gpu_ctx, sb, sky, w, h, depth, spp, threshold, camera, lanes=1):
    from raytracing_amd import device, scene
    p = device.PathTracingPass(gpu_ctx, w, h)
    p.bind_scene(sb)
    p.set_pipeline(lanes)
    p.set_tail_threshold(threshold)
    jit = scene.mt19937_jitters(spp)
    for s in range(1, spp + 1):
        p.set_params(scene.make_params(w, h, camera, sky, spp=s, jitter=jit[s - 1], max_depth=depth))
        p.draw()
        if lanes == 1:
            gpu_ctx.synchronize()       # the next draw then sees this draw's queue lengths
    out = p.download()
    st = p.stats()
    prof_free = {k: st[k] for k in ("samples", "rays_extend", "rays_shadow", "rays_retraced", "draws")}
    p.destroy()
    return out, prof_free


@pytest.mark.parametrize("depth", [4, 8, 16])
def test_tail_launch_gives_the_same_frame(gpu_ctx, casa, small_sky, depth):
    from raytracing_amd import device, scene
    w, h, spp = 480, 270, 4
    sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
    ref, ref_st = _render(gpu_ctx, sb, small_sky, w, h, depth, spp, 0, scene.casa_camera())
    # thresholds: everything from vertex 1 on; a mid value; the library's choice; several draws in flight
    for threshold, lanes in ((1 << 30, 1), (20000, 1), (-1, 1), (1 << 30, 3)):
        got, st = _render(gpu_ctx, sb, small_sky, w, h, depth, spp, threshold, scene.casa_camera(), lanes)
        for k in ref:
            assert np.array_equal(ref[k].view(np.uint32), got[k].view(np.uint32)), (k, threshold, lanes)
        assert st == ref_st, (threshold, lanes)
    sb.release()


def test_tail_launch_with_dielectrics_and_textures_vs_oracle(gpu_ctx, small_sky, oracle_mod):
    from raytracing_amd import device, scene
    sd = scene.synthetic_scene(30000, dielectric_fraction=0.5, textures=3, texture_size=64)
    w, h, depth = 200, 120, 12
    sb = device.SceneBuffers(gpu_ctx, sd, small_sky)
    p = device.PathTracingPass(gpu_ctx, w, h)
    p.bind_scene(sb)
    p.set_tail_threshold(1 << 30)
    osc = oracle_mod.OracleScene(sd, small_sky)
    ref = None
    jit = scene.mt19937_jitters(3)
    for s in (1, 2, 3):
        prm = scene.make_params(w, h, scene.synthetic_camera(), small_sky, spp=s, jitter=jit[s - 1], max_depth=depth)
        p.set_params(prm)
        p.draw()
        gpu_ctx.synchronize()
        ref, _ = osc.render(prm, ref)
    got = p.download()
    prof = p.profile()
    for k in ref:
        a, b = got[k], ref[k]
        assert np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))), k
    p.destroy()
    sb.release()


def test_tail_really_runs(gpu_ctx, casa, small_sky):
    """After the first draw has left its queue lengths, later draws use the tail launch."""
    from raytracing_amd import device, scene
    w, h = 320, 180
    sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
    p = device.PathTracingPass(gpu_ctx, w, h)
    p.bind_scene(sb)
    p.set_pipeline(1)
    p.set_profiling(True)
    p.set_tail_threshold(1 << 30)
    for s in (1, 2, 3):
        p.set_params(scene.make_params(w, h, scene.casa_camera(), small_sky, spp=s, max_depth=8))
        p.draw()
        gpu_ctx.synchronize()
    prof = p.profile()
    assert prof["tail_launches"] == 2 and prof["tail_ms"] > 0
    assert prof["shade_launches"] == 9 + 1 + 1      # the first draw shades vertex by vertex, the others only vertex 0
    p.set_tail_threshold(0)
    p.profile(reset=True)
    p.set_params(scene.make_params(w, h, scene.casa_camera(), small_sky, spp=4, max_depth=8))
    p.draw()
    assert p.profile()["tail_launches"] == 0
    p.destroy()
    sb.release()


def test_material_grouping_changes_nothing(gpu_ctx, small_sky, oracle_mod):
    """pt_pass_set_material_grouping: the shade launches sort the paths of every block into two groups (0: not at
    all, 1: dielectrics behind conductors at every vertex, -1: escaped paths behind the others from vertex 1 on);
    the frame (checked against the oracle) and the ray counts stay the same."""
    from raytracing_amd import device, scene
    sd = scene.synthetic_scene(30000, dielectric_fraction=0.5, textures=3, texture_size=64, lights=1)
    w, h, depth = 333, 187, 6          # (sizes that leave ragged blocks and tiles)
    sb = device.SceneBuffers(gpu_ctx, sd, small_sky)
    prm = scene.make_params(w, h, scene.synthetic_camera(), small_sky, spp=1, max_depth=depth, light_count=1)
    ref, _ = oracle_mod.OracleScene(sd, small_sky).render(prm)
    stats = []
    for mode in (0, 1, -1):
        p = device.PathTracingPass(gpu_ctx, w, h)
        p.bind_scene(sb)
        p.set_material_grouping(mode)
        p.set_params(prm)
        p.draw()
        got = p.download()
        for k in ref:
            a, b = got[k], ref[k]
            assert np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))), (k, mode)
        st = p.stats()
        stats.append((st["rays_extend"], st["rays_shadow"], st["rays_retraced"]))
        p.destroy()
    assert stats[0] == stats[1] == stats[2]
    sb.release()
