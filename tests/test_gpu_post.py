"""GPU parity of the post-processing entry points (pt_post_merge, pt_post_static_blend,
pt_post_tonemap, pt_buf_save_as_image) against the oracle: every word identical."""
import numpy as np
import pytest

from test_post import decode_png

pytestmark = pytest.mark.gpu


def bit_equal(a, b):
    a = np.ascontiguousarray(a, np.float32).ravel()
    b = np.ascontiguousarray(b, np.float32).ravel()
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def test_post_chain_matches_oracle(gpu_ctx, oracle_mod, tmp_path):
    from raytracing_amd import device
    rng = np.random.default_rng(11)
    w, h = 321, 123
    n = w * h
    di = (rng.random((n, 3), np.float32) ** 4 * 50).astype(np.float32)
    idi = (rng.random((n, 3), np.float32) ** 2 * 3).astype(np.float32)
    al = rng.random((n, 3), np.float32)
    di[17, 0] = np.nan; idi[99, 1] = np.inf; al[1234, 2] = -np.inf; di[5] = 0; idi[5] = 0; al[5, 1] = np.inf
    b_di, b_idi, b_al = (device.Buffer(gpu_ctx, a) for a in (di, idi, al))
    b_col = device.Buffer(gpu_ctx, nbytes=n * 12)
    device.post_merge(gpu_ctx, b_di, b_idi, b_al, b_col)
    want_col = oracle_mod.post_merge(di, idi, al)
    assert bit_equal(b_col.download(), want_col)

    hist_c = np.zeros((n, 3), np.float32); hist_m = np.zeros((n, 2), np.float32)
    b_hc, b_hm = device.Buffer(gpu_ctx, hist_c), device.Buffer(gpu_ctx, hist_m)
    for frame in (1, 2, 3):
        col = (want_col.reshape(n, 3) * rng.random((n, 1), np.float32)).astype(np.float32)
        if frame == 3:
            col[40:60] = 800.0  # fireflies
        b_col.copy_from_host(col)
        device.post_static_blend(gpu_ctx, frame, b_col, b_hc, b_hm)
        hist_c, hist_m = oracle_mod.post_static_blend(frame, col, hist_c, hist_m)
        assert bit_equal(b_hc.download(), hist_c), frame
        assert bit_equal(b_hm.download(), hist_m), frame

    b_out = device.Buffer(gpu_ctx, nbytes=n * 12)
    device.post_tonemap(gpu_ctx, b_hc, b_out)
    want_ldr = oracle_mod.post_tonemap(hist_c)
    assert bit_equal(b_out.download(), want_ldr)
    device.post_tonemap(gpu_ctx, b_out, b_out)  # in place
    assert bit_equal(b_out.download(), oracle_mod.post_tonemap(want_ldr))

    b_out.copy_from_host(want_ldr)
    path = tmp_path / "frame.png"
    b_out.save_as_image(w, h, 3, path)
    assert np.array_equal(decode_png(str(path)), oracle_mod.quantize_image(want_ldr, w, h, 3))

    from raytracing_amd import _capi
    assert _capi.lib.pt_buf_save_as_image(b_out.h, w, h, 4, str(path).encode()) != 0   # size mismatch
    small = device.Buffer(gpu_ctx, nbytes=24)
    assert _capi.lib.pt_post_merge(gpu_ctx.h, b_di.h, b_idi.h, b_al.h, small.h) != 0
    assert _capi.lib.pt_post_static_blend(gpu_ctx.h, 0, b_col.h, b_hc.h, b_hm.h) != 0
    for b in (b_di, b_idi, b_al, b_col, b_hc, b_hm, b_out, small):
        b.release()


def test_rendered_frame_to_png(gpu_ctx, casa, small_sky, oracle_mod, tmp_path):
    """draw -> merge -> tonemap -> save_as_image, the reference's frame loop without the
    denoisers (main.cpp:206-315), against the oracle end to end."""
    from raytracing_amd import device, scene
    w, h = 96, 54
    sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
    p = device.PathTracingPass(gpu_ctx, w, h)
    p.bind_scene(sb)
    prm = scene.make_params(w, h, scene.casa_camera(), small_sky, spp=1, frame=1, jitter=(0.5, 0.5), max_depth=2,
                            light_count=len(casa.lights))
    p.set_params(prm)
    p.draw()
    col = device.Buffer(gpu_ctx, nbytes=w * h * 12)
    device.post_merge(gpu_ctx, p.directLumGBufferSSBO, p.indirectLumGBufferSSBO, p.albedoGBufferSSBO, col)
    device.post_tonemap(gpu_ctx, col, col)
    path = tmp_path / "casa.png"
    col.save_as_image(w, h, 3, path)
    ref, _ = oracle_mod.OracleScene(casa, small_sky).render(prm)
    want = oracle_mod.post_tonemap(oracle_mod.post_merge(ref["directLum"], ref["indirectLum"], ref["albedo"]))
    img = decode_png(str(path))
    assert np.array_equal(img, oracle_mod.quantize_image(want, w, h, 3))
    assert img.std() > 5  # an actual picture, not a constant
    col.release(); p.destroy(); sb.release()
