"""GPU parity of the denoising chain (pt_post_svgf_temporal / _variance / _spatial, pt_post_taa;
SVGF_*.glsl, TAA.glsl) against the oracle: a three-frame sequence of casa renders with a moving
camera (non-zero motion vectors, disocclusions), every intermediate buffer compared word for
word, plus synthetic inputs with NaN / Inf / out-of-range motion."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bit_equal(a, b):
    a = np.ascontiguousarray(a, np.float32).ravel()
    b = np.ascontiguousarray(b, np.float32).ravel()
    return a.size == b.size and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


class Chain:
    """The buffers of SVGFTemporalFilter + SVGFVarianceFilter + SVGFSpatialFilterPass + TAA on the
    device, driven as main.cpp:213-300 drives them, with a numpy/oracle twin."""

    def __init__(self, ctx, oracle_mod, w, h, levels):
        from raytracing_amd import device
        self.ctx, self.o, self.w, self.h, self.levels, self.dev = ctx, oracle_mod, w, h, levels, device
        n = w * h
        z = lambda c: np.zeros(n * c, np.float32)
        self.host = dict(h_color=z(3), h_moment=z(2), h_num=z(1), h_normal=z(3), h_inst=z(1),
                         taa_color=z(3), taa_normal=z(3), taa_inst=z(1))
        self.buf = {k: device.Buffer(ctx, v) for k, v in self.host.items()}
        for k, c in (("o_color", 3), ("o_moment", 2), ("o_num", 1), ("variance", 1), ("ping", 3), ("pong", 3), ("taa_out", 3)):
            self.buf[k] = device.Buffer(ctx, nbytes=n * c * 4)
        self.taa_first = True

    def svgf(self, color_buf, color_np, g, gnp):
        d, o, w, h, B, H = self.dev, self.o, self.w, self.h, self.buf, self.host
        d.post_svgf_temporal(self.ctx, w, h, False, color_buf, g["normal"], g["instanceIndex"], g["motion"], B["o_color"], B["o_moment"],
                             B["o_num"], B["h_color"], B["h_moment"], B["h_num"], B["h_normal"], B["h_inst"])
        oc, om, on = o.svgf_temporal(w, h, False, color_np, gnp["normal"], gnp["instanceIndex"], gnp["motion"], H["h_color"],
                                     H["h_moment"], H["h_num"], H["h_normal"], H["h_inst"])
        assert bit_equal(B["o_color"].download(), oc) and bit_equal(B["o_moment"].download(), om) and bit_equal(B["o_num"].download(), on)
        for dst, src, val in (("h_color", "o_color", oc), ("h_moment", "o_moment", om), ("h_num", "o_num", on)):
            B[dst].copy_from(B[src])
            H[dst] = val
        B["h_normal"].copy_from(g["normal"]); H["h_normal"] = gnp["normal"].ravel().copy()
        B["h_inst"].copy_from(g["instanceIndex"]); H["h_inst"] = gnp["instanceIndex"].ravel().copy()
        d.post_svgf_variance(self.ctx, w, h, B["o_moment"], g["normal"], g["depth"], B["o_num"], B["variance"])
        var = o.svgf_variance(w, h, om, gnp["normal"], gnp["depth"], on)
        assert bit_equal(B["variance"].download(), var)
        B["ping"].copy_from(B["o_color"])
        cur, cur_np, other = "ping", oc, "pong"
        for lvl in range(self.levels):
            d.post_svgf_spatial(self.ctx, w, h, 1 << lvl, B[cur], g["normal"], g["depth"], B["variance"], B[other])
            cur_np = o.svgf_spatial(w, h, 1 << lvl, cur_np, gnp["normal"], gnp["depth"], var)
            cur, other = other, cur
            assert bit_equal(B[cur].download(), cur_np), lvl
        B["h_color"].copy_from(B[cur]); H["h_color"] = cur_np       # update_historycolor
        return B[cur], cur_np

    def taa(self, ldr_buf, ldr_np, g, gnp):
        d, o, w, h, B, H = self.dev, self.o, self.w, self.h, self.buf, self.host
        if self.taa_first:                                           # TAA.cpp:19-25
            self.taa_first = False
            out_buf, out_np = ldr_buf, ldr_np
            B["taa_color"].copy_from(ldr_buf); H["taa_color"] = ldr_np.copy()
        else:
            d.post_taa(self.ctx, w, h, ldr_buf, B["taa_color"], g["motion"], g["normal"], g["instanceIndex"], B["taa_normal"],
                       B["taa_inst"], B["taa_out"])
            out_np = o.taa(w, h, ldr_np, H["taa_color"], gnp["motion"], gnp["normal"], gnp["instanceIndex"], H["taa_normal"], H["taa_inst"])
            assert bit_equal(B["taa_out"].download(), out_np)
            out_buf = B["taa_out"]
            B["taa_color"].copy_from(B["taa_out"]); H["taa_color"] = out_np
        B["taa_normal"].copy_from(g["normal"]); H["taa_normal"] = gnp["normal"].ravel().copy()
        B["taa_inst"].copy_from(g["instanceIndex"]); H["taa_inst"] = gnp["instanceIndex"].ravel().copy()
        return out_buf, out_np

    def release(self):
        for b in self.buf.values():
            b.release()


def test_svgf_and_taa_over_a_camera_move(gpu_ctx, oracle_mod, casa, small_sky):
    from raytracing_amd import device, scene
    w, h, levels = 160, 90, 3
    sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
    p = device.PathTracingPass(gpu_ctx, w, h)
    p.bind_scene(sb)
    chain_di = Chain(gpu_ctx, oracle_mod, w, h, levels)
    chain_idi = Chain(gpu_ctx, oracle_mod, w, h, levels)
    merged = device.Buffer(gpu_ctx, nbytes=w * h * 12)
    ldr = device.Buffer(gpu_ctx, nbytes=w * h * 12)
    prev_v2w = None
    fov = None
    saw_history = False
    for frame in range(1, 4):
        v2w = scene.transform_matrix((-16.9924 + 0.35 * frame, 8.50888, 17.9285 - 0.2 * frame), (-23.19, -48.8484 + 1.5 * frame, 0.0), (1, 1, 1))
        prm0 = scene.make_params(w, h, v2w, small_sky, spp=1, frame=frame, max_depth=2)
        # back-projection of the PREVIOUS frame's camera (main.cpp:160), so that motion vectors point at the history
        if prev_v2w is not None:
            bp = scene.mat4_mul(scene.camera_projection(prm0.fov, 1000.0, w, h), scene.mat4_inverse(prev_v2w))
            prm = scene.make_params(w, h, v2w, small_sky, spp=1, frame=frame, max_depth=2, backproj=bp)
        else:
            prm = prm0
        prev_v2w = v2w
        p.set_params(prm)
        p.draw()
        gnp = p.download()
        g = p.outputs
        di_buf, di_np = chain_di.svgf(g["directLum"], gnp["directLum"].ravel(), g, gnp)
        idi_buf, idi_np = chain_idi.svgf(g["indirectLum"], gnp["indirectLum"].ravel(), g, gnp)
        device.post_merge(gpu_ctx, di_buf, idi_buf, g["albedo"], merged)
        device.post_tonemap(gpu_ctx, merged, ldr)
        ldr_np = oracle_mod.post_tonemap(oracle_mod.post_merge(di_np, idi_np, gnp["albedo"]))
        assert bit_equal(ldr.download(), ldr_np)
        out_buf, out_np = chain_di.taa(ldr, ldr_np, g, gnp)
        if frame > 1:
            saw_history = saw_history or float(chain_di.host["h_num"].max()) > 1.0
    assert saw_history                                        # the temporal filter did find its history through the motion vectors
    assert np.isfinite(out_np).all() and out_np.std() > 0.01
    for c in (chain_di, chain_idi):
        c.release()
    merged.release(); ldr.release(); p.destroy(); sb.release()


def test_denoise_kernels_on_hostile_inputs(gpu_ctx, oracle_mod):
    from raytracing_amd import device, _capi
    w, h = 67, 41
    n = w * h
    rng = np.random.default_rng(21)
    color = (rng.random((n, 3)) ** 3 * 20).astype(np.float32)
    normal = rng.standard_normal((n, 3)).astype(np.float32)
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    normal[::7] = normal[0]                                        # patches that pass the geometry test
    inst = rng.integers(0, 3, n).astype(np.float32)
    motion = ((rng.random((n, 2)) - 0.5) * 0.2).astype(np.float32)
    depth = (10 + rng.random(n) * 5).astype(np.float32)
    color[3] = np.nan; color[50, 1] = np.inf; normal[9] = 0; normal[11] = np.nan
    motion[5] = np.nan; motion[6] = (5.0, -7.0); motion[7] = (np.inf, 0.0); motion[8] = (-1e30, 1e30)
    depth[13] = 1e6; depth[14] = np.nan
    h_color = (rng.random((n, 3)) * 5).astype(np.float32); h_moment = (rng.random((n, 2)) * 30).astype(np.float32)
    h_num = rng.integers(1, 12, n).astype(np.float32)
    h_normal = normal.copy(); h_normal[::5] = -h_normal[::5]
    h_inst = inst.copy(); h_inst[::11] += 1
    B = lambda a: device.Buffer(gpu_ctx, np.ascontiguousarray(a, np.float32))
    O = lambda c: device.Buffer(gpu_ctx, nbytes=n * c * 4)
    b = dict(color=B(color), normal=B(normal), inst=B(inst), motion=B(motion), depth=B(depth), h_color=B(h_color), h_moment=B(h_moment),
             h_num=B(h_num), h_normal=B(h_normal), h_inst=B(h_inst), oc=O(3), om=O(2), on=O(1), var=O(1), sp=O(3), taa=O(3))
    for first in (True, False):
        device.post_svgf_temporal(gpu_ctx, w, h, first, b["color"], b["normal"], b["inst"], b["motion"], b["oc"], b["om"], b["on"],
                                  b["h_color"], b["h_moment"], b["h_num"], b["h_normal"], b["h_inst"])
        oc, om, on = oracle_mod.svgf_temporal(w, h, first, color, normal, inst, motion, h_color, h_moment, h_num, h_normal, h_inst)
        assert bit_equal(b["oc"].download(), oc) and bit_equal(b["om"].download(), om) and bit_equal(b["on"].download(), on)
    assert (on > 1).any() and (on == 1).any()
    device.post_svgf_variance(gpu_ctx, w, h, b["om"], b["normal"], b["depth"], b["on"], b["var"])
    var = oracle_mod.svgf_variance(w, h, om, normal, depth, on)
    assert bit_equal(b["var"].download(), var)
    for step in (1, 2, 16, 1 << 20):
        device.post_svgf_spatial(gpu_ctx, w, h, step, b["oc"], b["normal"], b["depth"], b["var"], b["sp"])
        assert bit_equal(b["sp"].download(), oracle_mod.svgf_spatial(w, h, step, oc, normal, depth, var)), step
    ldr = np.clip(color / 20, 0, 1).astype(np.float32); ldr[3] = np.nan
    b["ldr"] = B(ldr)
    device.post_taa(gpu_ctx, w, h, b["ldr"], b["h_color"], b["motion"], b["normal"], b["inst"], b["h_normal"], b["h_inst"], b["taa"])
    assert bit_equal(b["taa"].download(), oracle_mod.taa(w, h, ldr, h_color, motion, normal, inst, h_normal, h_inst))
    # argument checks
    assert _capi.lib.pt_post_svgf_spatial(gpu_ctx.h, w, h, 0, b["oc"].h, b["normal"].h, b["depth"].h, b["var"].h, b["sp"].h) != 0
    assert _capi.lib.pt_post_svgf_spatial(gpu_ctx.h, w, h, 1, b["oc"].h, b["normal"].h, b["depth"].h, b["var"].h, b["oc"].h) != 0   # aliasing
    assert _capi.lib.pt_post_svgf_variance(gpu_ctx.h, w + 1, h, b["om"].h, b["normal"].h, b["depth"].h, b["on"].h, b["var"].h) != 0
    for x in b.values():
        x.release()
