"""Error behaviour of the boundary on a real device.  The reference has no error channel
(SSBOBuffer.h and RenderPass.cpp print and carry on); this library returns a status and a
message, and must refuse malformed scene buffers on the host instead of letting a kernel
follow a bad index."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def fails(fn, *needles):
    from raytracing_amd import _capi
    with pytest.raises(_capi.PtError) as e:
        fn()
    msg = str(e.value)
    for n in needles:
        assert n in msg, msg
    return msg


def test_draw_preconditions(gpu_ctx, casa, small_sky):
    from raytracing_amd import device, scene
    w, h = 32, 18
    p = device.PathTracingPass(gpu_ctx, w, h)
    fails(p.draw, "set_params")
    prm = scene.make_params(w, h, scene.casa_camera(), small_sky)
    p.set_params(prm)
    fails(p.draw, "slot")                       # nothing bound
    sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
    for slot, buf in enumerate(sb.buffers()):
        p.bind(slot, buf)
    fails(p.draw, "skybox")                     # sky texture not bound
    p.bind_texture("skybox", sb.sky_tex)
    fails(p.draw, "skybox_samplecache")
    p.bind_texture("skybox_samplecache", sb.sky_cache_tex)
    p.draw()                                    # complete now
    bad = scene.make_params(w + 1, h, scene.casa_camera(), small_sky)
    fails(lambda: p.set_params(bad), "SCREEN_W")
    bad = scene.make_params(w, h, scene.casa_camera(), small_sky, max_depth=65)
    fails(lambda: p.set_params(bad), "MAX_DEPTH")
    bad = scene.make_params(w, h, scene.casa_camera(), small_sky, light_count=2)
    p.set_params(bad)
    fails(p.draw, "lightCount")                 # more lights than the bound buffer holds
    fails(lambda: p.set_pipeline(0), "1..")
    fails(lambda: p.set_traversal(4), "mode")
    fails(lambda: p.bind_texture("albedoTexture", sb.sky_tex), "unknown sampler")
    p.destroy()
    sb.release()


def test_malformed_scene_buffers_are_refused(gpu_ctx, casa, small_sky):
    from raytracing_amd import device, scene
    w, h = 32, 18
    prm = scene.make_params(w, h, scene.casa_camera(), small_sky)

    def attempt(mutate, *needles):
        import copy
        sd = copy.copy(casa)
        for name in ("mesh_nodes", "scene_nodes", "instances", "materials", "triangles"):
            setattr(sd, name, getattr(casa, name).copy())
        mutate(sd)
        sb = device.SceneBuffers(gpu_ctx, sd, small_sky)
        p = device.PathTracingPass(gpu_ctx, w, h)
        p.bind_scene(sb)
        p.set_params(prm)
        try:
            fails(p.draw, *needles)
        finally:
            p.destroy()
            sb.release()

    def child_out_of_range(sd):
        sd.mesh_nodes["ls"][0] = len(sd.mesh_nodes) + 5
    attempt(child_out_of_range, "rejected")

    def triangle_out_of_range(sd):
        leaf = np.nonzero(sd.mesh_nodes["tri"] >= 0)[0][0]
        sd.mesh_nodes["tri"][leaf] = len(sd.triangles)
    attempt(triangle_out_of_range, "rejected")

    def cycle(sd):
        inner = np.nonzero(sd.mesh_nodes["ls"] >= 0)[0]
        sd.mesh_nodes["ls"][inner[3]] = 0       # back edge to a root
    attempt(cycle, "rejected")

    def bad_instance_mesh(sd):
        sd.instances["meshIndex"][1] = 10_000_000
    attempt(bad_instance_mesh, "rejected")

    def bad_material_pointer(sd):
        sd.instances["materialPtr"][0] = len(sd.materials) - 2
    attempt(bad_material_pointer, "material")

    def truncated_nodes(sd):
        sd.mesh_nodes = sd.mesh_nodes[:-7].copy()
    attempt(truncated_nodes, "rejected")


def test_buffer_and_intersect_argument_checks(gpu_ctx, casa, small_sky):
    from raytracing_amd import device, _capi
    b = device.Buffer(gpu_ctx, np.arange(16, dtype=np.float32))
    fails(lambda: b.copy_from_host(np.zeros(8, np.float32)), "size")
    other = device.Buffer(gpu_ctx, nbytes=32)
    fails(lambda: other.copy_from(b), "sizes differ")
    out = np.zeros(32, np.float32)
    assert _capi.lib.pt_buf_download(b.h, out.ctypes.data, 128) != 0   # more than the buffer holds
    empty = device.Buffer(gpu_ctx, nbytes=0)                            # zero-length buffers are legal (no lights)
    assert empty.nbytes == 0 and not empty.isempty()
    sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
    p = device.PathTracingPass(gpu_ctx, 16, 16)
    p.bind_scene(sb)
    rays = device.Buffer(gpu_ctx, np.zeros(6 * 10, np.float32))
    hits = device.Buffer(gpu_ctx, nbytes=20 * 10)
    fails(lambda: p.intersect(rays, 11, hits), "fewer than n rays")
    small_hits = device.Buffer(gpu_ctx, nbytes=20 * 5)
    fails(lambda: p.intersect(rays, 10, small_hits), "fewer than n records")
    p.intersect(rays, 0, hits)                                          # n = 0 is a no-op
    for x in (b, other, empty, rays, hits, small_hits):
        x.release()
    p.destroy()
    sb.release()


def test_released_resources_are_unbound_not_dangling(gpu_ctx, casa, small_sky):
    """Releasing a buffer or texture that a pass still has bound (the reference's release();
    x = SSBOBuffer(...) pattern, ResourceManager.cpp:53-54, forgets to rebind) makes the next draw
    fail with a message instead of reading freed memory."""
    from raytracing_amd import device, scene
    w, h = 32, 18
    sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
    p = device.PathTracingPass(gpu_ctx, w, h)
    p.bind_scene(sb)
    p.set_params(scene.make_params(w, h, scene.casa_camera(), small_sky))
    p.draw()
    sb.textures[0].release()                    # an albedo map named by the handle buffer
    fails(p.draw, "live texture")
    sb.sky_cache_tex.release()
    sb.sky_cache_tex = None
    fails(p.draw, "live texture")               # still the first problem found
    sb.materialSSBO.release()
    fails(p.draw, "slot 1")
    p.destroy()
    sb.textures = sb.textures[1:]
    sb.release()


def test_stream_handed_out_is_ordered_behind_draws(gpu_ctx, casa, small_sky):
    """Draws run on streams of their own.  pt_ctx_stream() orders the context's main stream behind
    the draws issued so far, so a device-to-device copy of an output queued on it right after
    pt_pass_draw (no synchronisation in between) holds the finished frame."""
    import ctypes
    from raytracing_amd import device, scene
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    w, h = 640, 360
    sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
    p = device.PathTracingPass(gpu_ctx, w, h)
    p.bind_scene(sb)
    p.set_pipeline(3)
    copies = {k: device.Buffer(gpu_ctx, nbytes=p.outputs[k].nbytes) for k in ("directLum", "indirectLum", "normal")}
    for b in copies.values():
        b.fill(-7.0)
    jit = scene.mt19937_jitters(3)
    gpu_ctx.synchronize()
    for s in (1, 2, 3):
        p.set_params(scene.make_params(w, h, scene.casa_camera(), small_sky, spp=s, max_depth=4, jitter=jit[s - 1]))
        p.draw()
    stream = gpu_ctx.stream   # joins the three draws
    assert stream
    for k, b in copies.items():
        assert hip.hipMemcpyAsync(b.device_ptr, p.outputs[k].device_ptr, b.nbytes, 3, stream) == 0
    assert hip.hipStreamSynchronize(stream) == 0
    got = {k: b.download() for k, b in copies.items()}     # (download waits again; the copies are done by now)
    ref = p.download()
    for k in copies:
        assert np.array_equal(got[k].view(np.uint32), ref[k].reshape(-1).view(np.uint32)), k
    for b in copies.values():
        b.release()
    p.destroy()
    sb.release()


_OVERFLOW_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from raytracing_amd import _capi, device, scene
casa = scene.casa_scene(sys.argv[1] + "/assets/casa_obj.glb")
ctx = device.Context(0)
lo, hi = casa.world_box()
rays = scene.random_rays(lo, hi, 1 << 18, 5)
sb = device.SceneBuffers(ctx, casa)
rb = device.Buffer(ctx, rays)
hb = device.Buffer(ctx, nbytes=len(rays) * device.HIT_DTYPE.itemsize)

def reports(fn):
    try:
        fn()
    except _capi.PtError as e:
        assert "stack overflow" in str(e), str(e)
        return True
    return False

for mode in (0, 1):
    p = device.PathTracingPass(ctx, 16, 16)
    p.bind_scene(sb)
    p.set_traversal(mode)
    p.intersect(rb, len(rays), hb)
    assert reports(ctx.synchronize), "synchronize did not report (mode %d)" % mode
    assert not reports(ctx.synchronize), "reported twice"
    p.intersect(rb, len(rays), hb)
    assert reports(lambda: hb.download(np.uint8)), "download did not report"
    p.intersect(rb, len(rays), hb)
    assert reports(p.stats), "stats did not report"
    p.destroy()
print("overflow reported")
"""


def test_stack_overflow_is_reported_by_the_calls_that_wait():
    """A build of the library with three stack entries per ray in LDS (libraytracing_amd.smallstack.so,
    made by __graft_entry__.build) and the spill area capped to nothing: deep rays run out of stack
    in both traversals.  The kernels drop the entry (no out-of-bounds access) and raise a flag that
    pt_ctx_synchronize / pt_buf_download / pt_pass_stats report, once.  Runs in a child process
    because the library is chosen when raytracing_amd._capi is imported."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    lib = os.path.join(ROOT, "raytracing_amd", "libraytracing_amd.smallstack.so")
    assert os.path.exists(lib), "build it with __graft_entry__.build()"
    env = dict(os.environ, RAYTRACING_AMD_LIB=lib, RTA_SPILL_DEPTH_CAP="0")
    r = subprocess.run([sys.executable, "-c", _OVERFLOW_SCRIPT, ROOT], env=env, capture_output=True, text=True, timeout=100)
    assert r.returncode == 0 and "overflow reported" in r.stdout, r.stdout + r.stderr


def test_prepare_builds_the_traversal_arrays_ahead_of_the_first_draw(casa, gpu_ctx, small_sky):
    """pt_pass_prepare does the eager half of ResourceManager::reload_meshes / reload_scene
    (src/ResourceManager.cpp:95-224): after it, the first draw finds nothing to rebuild; with a slot
    unbound it reports that instead of the draw."""
    from raytracing_amd import _capi, device, scene
    p = device.PathTracingPass(gpu_ctx, 64, 36)
    with pytest.raises(_capi.PtError, match="not bound"):
        p.prepare()
    sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
    p.bind_scene(sb)
    p.prepare()
    st = p.stats()
    assert st["mesh_level_builds"] == 1 and st["scene_level_builds"] == 1 and st["draws"] == 0
    p.set_params(scene.make_params(64, 36, scene.casa_camera(), small_sky, spp=1, max_depth=2))
    p.draw()
    p.prepare()
    st = p.stats()
    assert st["mesh_level_builds"] == 1 and st["scene_level_builds"] == 1 and st["draws"] == 1
    p.destroy()
    sb.release()
