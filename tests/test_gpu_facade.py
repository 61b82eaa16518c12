"""The C++ facade (include/raytracing_amd.hpp: SSBOBuffer, Renderer, SVGFMergePass,
StaticBlender, ToneMappingGamma, ResourceManager with the reference's names) driven by a C++
program shaped like the reference's update() (main.cpp:169-315), checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from test_post import decode_png

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "facade_frame")


def bit_equal(a, b):
    a = np.ascontiguousarray(a, np.float32).ravel()
    b = np.ascontiguousarray(b, np.float32).ravel()
    return a.size == b.size and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("from_glb", [False, True])
def test_facade_frame_loop(casa, small_sky, oracle_mod, tmp_path, from_glb):
    from raytracing_amd import scene
    assert os.path.exists(EXE), "tests/cpp/facade_frame is built by __graft_entry__.build() / raytracing_amd.build.build_facade_demo()"
    w, h, spp_n, depth = 112, 63, 3, 3
    d = str(tmp_path)
    casa.triangles.tofile(d + "/tris.bin"); casa.instances.tofile(d + "/inst.bin")
    casa.mesh_nodes.tofile(d + "/mesh_nodes.bin"); casa.scene_nodes.tofile(d + "/scene_nodes.bin")
    np.ascontiguousarray(casa.materials, np.float32).tofile(d + "/materials.bin")
    casa.lights.tofile(d + "/lights.bin")
    meta = ["%d %d %d %d" % (w, h, spp_n, len(casa.textures))]
    for k, t in enumerate(casa.textures):
        np.ascontiguousarray(t, np.uint8).tofile(d + "/tex%d.bin" % k)
        meta.append("%d %d %d" % (t.shape[1], t.shape[0], t.shape[2]))
    meta.append("%d %d" % (small_sky.rgb.shape[1], small_sky.rgb.shape[0]))
    open(d + "/meta.txt", "w").write("\n".join(meta) + "\n")
    small_sky.rgb.astype(np.float32).tofile(d + "/sky.bin")
    small_sky.cache.astype(np.float32).tofile(d + "/sky_cache.bin")
    jit = scene.mt19937_jitters(spp_n)
    v2w = scene.casa_camera()
    prms = [scene.make_params(w, h, v2w, small_sky, spp=s, frame=1, jitter=jit[s - 1], max_depth=depth,
                              light_count=len(casa.lights)) for s in range(1, spp_n + 1)]
    open(d + "/params.bin", "wb").write(b"".join(bytes(p) for p in prms))

    if from_glb:
        # the program loads casa_obj.glb and builds every scene buffer itself (pt_host_glb_*, pt_host_build_*)
        for f in ("tris", "inst", "mesh_nodes", "scene_nodes", "materials"):
            os.remove(d + "/%s.bin" % f)
        r = subprocess.run([EXE, d, os.path.join(ROOT, "assets", "casa_obj.glb")], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "glb_scene 34689 triangles, 19 instances" in r.stdout
        for f, a in (("tris", casa.triangles), ("inst", casa.instances), ("mesh_nodes", casa.mesh_nodes), ("scene_nodes", casa.scene_nodes),
                     ("materials", np.ascontiguousarray(casa.materials, np.float32))):
            assert open(d + "/glb_%s.bin" % f, "rb").read() == a.tobytes(), f
    else:
        r = subprocess.run([EXE, d], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
    assert "ssbo_ok" in r.stdout

    osc = oracle_mod.OracleScene(casa, small_sky)
    hist_c = np.zeros(w * h * 3, np.float32); hist_m = np.zeros(w * h * 2, np.float32)
    n = w * h
    z = lambda c: np.zeros(n * c, np.float32)
    svgf = [dict(h_color=z(3), h_moment=z(2), h_num=z(1), h_normal=z(3), h_inst=z(1)) for _ in range(2)]
    taa_hist = None

    def denoise(acc):
        nonlocal taa_hist
        outs = []
        for k, name in enumerate(("directLum", "indirectLum")):
            H = svgf[k]
            oc, om, on = oracle_mod.svgf_temporal(w, h, False, acc[name], acc["normal"], acc["instanceIndex"], acc["motion"],
                                                  H["h_color"], H["h_moment"], H["h_num"], H["h_normal"], H["h_inst"])
            var = oracle_mod.svgf_variance(w, h, om, acc["normal"], acc["depth"], on)
            cur = oc
            for lvl in range(3):
                cur = oracle_mod.svgf_spatial(w, h, 1 << lvl, cur, acc["normal"], acc["depth"], var)
            H.update(h_color=cur, h_moment=om, h_num=on, h_normal=acc["normal"].ravel().copy(), h_inst=acc["instanceIndex"].ravel().copy())
            outs.append(cur)
        ldr = oracle_mod.post_tonemap(oracle_mod.post_merge(outs[0], outs[1], acc["albedo"]))
        if taa_hist is None:
            out = None
            taa_hist = dict(color=ldr)
        else:
            out = oracle_mod.taa(w, h, ldr, taa_hist["color"], acc["motion"], acc["normal"], acc["instanceIndex"], taa_hist["normal"], taa_hist["inst"])
            taa_hist["color"] = out
        taa_hist["normal"] = acc["normal"].ravel().copy()
        taa_hist["inst"] = acc["instanceIndex"].ravel().copy()
        return outs[0], outs[1], svgf[0]["h_num"], out

    for frame in (1, 2):
        acc = None
        for s in range(1, spp_n + 1):
            prm = scene.make_params(w, h, v2w, small_sky, spp=s, frame=frame, jitter=jit[s - 1], max_depth=depth,
                                    light_count=len(casa.lights))
            acc, _ = osc.render(prm, accum=acc)
        svgf_di, svgf_idi, svgf_num, taa_out = denoise(acc)
        color = oracle_mod.post_merge(acc["directLum"], acc["indirectLum"], acc["albedo"])
        hist_c, hist_m = oracle_mod.post_static_blend(frame, color, hist_c, hist_m)
        if frame == 1:
            assert bit_equal(np.fromfile(d + "/out_directLum_f1.bin", np.float32), acc["directLum"])
            assert bit_equal(np.fromfile(d + "/out_color_f1.bin", np.float32), color)
    for name in ("directLum", "indirectLum", "albedo", "normal", "depth", "instanceIndex", "motion"):
        assert bit_equal(np.fromfile(d + "/out_%s.bin" % name, np.float32), acc[name]), name
    assert bit_equal(np.fromfile(d + "/out_history.bin", np.float32), hist_c)
    ldr = oracle_mod.post_tonemap(hist_c)
    assert bit_equal(np.fromfile(d + "/out_ldr.bin", np.float32), ldr)
    assert np.array_equal(decode_png(d + "/frame.png"), oracle_mod.quantize_image(ldr, w, h, 3))
    # the denoising branch (SVGFTemporalFilter / SVGFVarianceFilter / SVGFSpatialFilterPass x2, merge, tone map, TAA), frame 2
    assert bit_equal(np.fromfile(d + "/out_svgf_di.bin", np.float32), svgf_di)
    assert bit_equal(np.fromfile(d + "/out_svgf_idi.bin", np.float32), svgf_idi)
    assert bit_equal(np.fromfile(d + "/out_svgf_num.bin", np.float32), svgf_num)
    assert bit_equal(np.fromfile(d + "/out_taa.bin", np.float32), taa_out)
