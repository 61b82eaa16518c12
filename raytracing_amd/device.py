"""Thin Python objects over the C ABI (include/raytracing_amd.h), shaped like the
reference's classes so that tests read like code written against them:

  Buffer            SSBOBuffer<T>            src/SSBOBuffer.h
  Texture           Texture / sky textures   src/texture/Texture.cpp, Skybox.cpp
  PathTracingPass   Renderer                 src/renderpass/Renderer.{h,cpp}
  SceneBuffers      the seven SSBO members of ResourceManager (src/ResourceManager.h:84-90)

Every call goes through libraytracing_amd.so; there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import PtProfile, PtStats, check, lib

HIT_DTYPE = np.dtype([("instance", "<i4"), ("triangle", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])

SLOT_TEXTURE_HANDLES, SLOT_MATERIAL, SLOT_TRIANGLE, SLOT_INSTANCE_INFO, SLOT_MESH_BVH, SLOT_SCENE_BVH, SLOT_LIGHT = range(7)
OUT_ALBEDO, OUT_MOTION, OUT_DEPTH, OUT_NORMAL, OUT_INSTANCE_INDEX, OUT_DIRECT_LUM, OUT_INDIRECT_LUM = 7, 9, 11, 12, 13, 14, 15
OUTPUTS = {"albedo": (OUT_ALBEDO, 3), "motion": (OUT_MOTION, 2), "depth": (OUT_DEPTH, 1), "normal": (OUT_NORMAL, 3),
           "instanceIndex": (OUT_INSTANCE_INDEX, 1), "directLum": (OUT_DIRECT_LUM, 3),
           "indirectLum": (OUT_INDIRECT_LUM, 3)}


class Context:
    def __init__(self, device_id=0):
        h = C.c_void_p()
        check(lib.pt_ctx_create(device_id, C.byref(h)))
        self.h = h

    def synchronize(self):
        check(lib.pt_ctx_synchronize(self.h))

    @property
    def stream(self):
        return lib.pt_ctx_stream(self.h)

    def destroy(self):
        if self.h:
            lib.pt_ctx_destroy(self.h)
            self.h = None


class Buffer:
    """SSBOBuffer: construct from a size or from host data; release() explicitly."""

    def __init__(self, ctx, data=None, nbytes=None, _borrowed=None):
        self.ctx = ctx
        if _borrowed is not None:
            self.h = _borrowed
            self.owned = False
            return
        self.owned = True
        h = C.c_void_p()
        if data is not None:
            a = np.ascontiguousarray(data)
            check(lib.pt_buf_create(ctx.h, a.nbytes, a.ctypes.data if a.nbytes else None, C.byref(h)))
        else:
            check(lib.pt_buf_create(ctx.h, int(nbytes), None, C.byref(h)))
        self.h = h

    @property
    def nbytes(self):
        return lib.pt_buf_bytes(self.h)

    @property
    def device_ptr(self):
        return lib.pt_buf_device_ptr(self.h)

    def isempty(self):
        return self.h is None

    def copy_from_host(self, data):
        a = np.ascontiguousarray(data)
        check(lib.pt_buf_upload(self.h, a.ctypes.data, a.nbytes))

    def copy_from(self, other):
        check(lib.pt_buf_copy(self.h, other.h))

    def fill(self, value, dtype=np.float32):
        v = np.array([value], dtype)
        check(lib.pt_buf_fill(self.h, v.ctypes.data, v.nbytes))

    def download(self, dtype=np.float32, count=None):
        n = self.nbytes if count is None else count * np.dtype(dtype).itemsize
        out = np.zeros(n // np.dtype(dtype).itemsize, dtype)
        check(lib.pt_buf_download(self.h, out.ctypes.data, out.nbytes))
        return out

    def save_as_image(self, width, height, channel, path):
        check(lib.pt_buf_save_as_image(self.h, int(width), int(height), int(channel), str(path).encode()))

    def release(self):
        if self.h and self.owned:
            lib.pt_buf_release(self.h)
        self.h = None


def post_merge(ctx, direct, indirect, albedo, color):
    """SVGF_Merge::draw: color = (direct + indirect) * albedo."""
    check(lib.pt_post_merge(ctx.h, direct.h, indirect.h, albedo.h, color.h))


def post_static_blend(ctx, frame_counter, color, history_color, history_moment):
    """StaticBlender::draw with the frame counter it would pass (1 on the first frame)."""
    check(lib.pt_post_static_blend(ctx.h, int(frame_counter), color.h, history_color.h, history_moment.h))


def post_tonemap(ctx, color, out):
    """ToneMappingGamma::draw."""
    check(lib.pt_post_tonemap(ctx.h, color.h, out.h))


def post_svgf_temporal(ctx, w, h, first_frame, color, normal, instance_index, motion, out_color, out_moment, out_num_samples,
                       history_color, history_moment, history_num_samples, history_normal, history_instance_index):
    """One dispatch of SVGF_TemporalFilter.glsl (the history copies of SVGFTemporalFilter::draw are the caller's)."""
    check(lib.pt_post_svgf_temporal(ctx.h, w, h, int(first_frame), color.h, normal.h, instance_index.h, motion.h, out_color.h,
                                    out_moment.h, out_num_samples.h, history_color.h, history_moment.h, history_num_samples.h,
                                    history_normal.h, history_instance_index.h))


def post_svgf_variance(ctx, w, h, moment, normal, depth, num_samples, out_variance):
    check(lib.pt_post_svgf_variance(ctx.h, w, h, moment.h, normal.h, depth.h, num_samples.h, out_variance.h))


def post_svgf_spatial(ctx, w, h, step, color, normal, depth, variance, out_color):
    check(lib.pt_post_svgf_spatial(ctx.h, w, h, int(step), color.h, normal.h, depth.h, variance.h, out_color.h))


def post_taa(ctx, w, h, color, history_color, motion, normal, instance_index, history_normal, history_instance_index, out_color):
    check(lib.pt_post_taa(ctx.h, w, h, color.h, history_color.h, motion.h, normal.h, instance_index.h, history_normal.h,
                          history_instance_index.h, out_color.h))


class Texture:
    def __init__(self, ctx, pixels):
        a = np.ascontiguousarray(pixels)
        assert a.ndim == 3 and a.dtype in (np.uint8, np.float32)
        h = C.c_void_p()
        check(lib.pt_tex_create(ctx.h, a.shape[1], a.shape[0], a.shape[2], 0 if a.dtype == np.uint8 else 1,
                                a.ctypes.data, C.byref(h)))
        self.h = h

    @property
    def handle(self):
        return lib.pt_tex_handle(self.h)

    def release(self):
        if self.h:
            lib.pt_tex_release(self.h)
            self.h = None


class SceneBuffers:
    """Uploads a scene.SceneData (+ optional scene.Sky) the way ResourceManager does."""

    def __init__(self, ctx, scene_data, sky=None):
        sd = scene_data
        self.ctx = ctx
        self.textures = [Texture(ctx, t) for t in sd.textures]
        handles = np.array([t.handle for t in self.textures], np.uint64)
        self.textureHandleSSBO = Buffer(ctx, handles)
        self.materialSSBO = Buffer(ctx, sd.materials)
        self.triangleSSBO = Buffer(ctx, sd.triangles)
        self.instanceInfoSSBO = Buffer(ctx, sd.instances)
        self.meshBVHSSBO = Buffer(ctx, sd.mesh_nodes)
        self.sceneBVHSSBO = Buffer(ctx, sd.scene_nodes)
        self.lightSSBO = Buffer(ctx, sd.lights)
        self.light_count = len(sd.lights)
        self.sky = sky
        self.sky_tex = Texture(ctx, sky.rgb) if sky is not None else None
        self.sky_cache_tex = Texture(ctx, sky.cache) if sky is not None else None

    def buffers(self):
        return [self.textureHandleSSBO, self.materialSSBO, self.triangleSSBO, self.instanceInfoSSBO,
                self.meshBVHSSBO, self.sceneBVHSSBO, self.lightSSBO]

    def release(self):
        for b in self.buffers():
            b.release()
        for t in self.textures + [self.sky_tex, self.sky_cache_tex]:
            if t is not None:
                t.release()


class PathTracingPass:
    """Renderer: seven public output buffers, draw() = one sample per pixel."""

    def __init__(self, ctx, width, height):
        self.ctx = ctx
        self.width, self.height = width, height
        h = C.c_void_p()
        check(lib.pt_pass_create(ctx.h, width, height, C.byref(h)))
        self.h = h
        self.outputs = {}
        for name, (slot, _) in OUTPUTS.items():
            b = C.c_void_p()
            check(lib.pt_pass_output(self.h, slot, C.byref(b)))
            self.outputs[name] = Buffer(ctx, _borrowed=b)
        # the reference's member names (Renderer.h:19-25)
        self.directLumGBufferSSBO = self.outputs["directLum"]
        self.indirectLumGBufferSSBO = self.outputs["indirectLum"]
        self.motionGBufferSSBO = self.outputs["motion"]
        self.albedoGBufferSSBO = self.outputs["albedo"]
        self.depthGBufferSSBO = self.outputs["depth"]
        self.normalGBufferSSBO = self.outputs["normal"]
        self.instanceIndexGBufferSSBO = self.outputs["instanceIndex"]

    def bind_scene(self, sb):
        for slot, buf in enumerate(sb.buffers()):
            check(lib.pt_pass_bind(self.h, slot, buf.h))
        if sb.sky_tex is not None:
            self.bind_texture("skybox", sb.sky_tex)
            self.bind_texture("skybox_samplecache", sb.sky_cache_tex)

    def bind(self, slot, buf):
        check(lib.pt_pass_bind(self.h, slot, buf.h if buf is not None else None))

    def bind_texture(self, name, tex):
        check(lib.pt_pass_bind_texture(self.h, name.encode(), tex.h if tex is not None else None))

    def set_params(self, params):
        check(lib.pt_pass_set_params(self.h, C.byref(params)))

    def draw(self):
        check(lib.pt_pass_draw(self.h))

    def draw_batch(self, params_list):
        """pt_pass_draw_batch: the samples of one frame (parameter sets differing in jitter, currentspp, frameCounter)
        as one draw; outputs as after set_params + draw for each in turn."""
        arr = (_capi.PtParams * len(params_list))(*params_list)
        check(lib.pt_pass_draw_batch(self.h, arr, len(params_list)))

    def reserve_batch(self, n):
        """Samples per draw_batch call that prepare() sizes the path state for."""
        check(lib.pt_pass_reserve_batch(self.h, int(n)))

    def intersect(self, rays_buf, n, hits_buf):
        check(lib.pt_pass_intersect(self.h, rays_buf.h, n, hits_buf.h))

    def intersect_noalloc(self, rays_buf, n, hits_buf):
        """pt_pass_intersect_noalloc: the same closest hits, with no device allocation in the call."""
        check(lib.pt_pass_intersect_noalloc(self.h, rays_buf.h, n, hits_buf.h))

    def stats(self, reset=False):
        s = PtStats()
        check(lib.pt_pass_stats(self.h, C.byref(s), int(reset)))
        return dict(samples=int(s.samples), rays_extend=int(s.rays_extend), rays_shadow=int(s.rays_shadow),
                    draws=int(s.draws), rays_retraced=int(s.rays_retraced), mesh_level_builds=int(s.mesh_level_builds),
                    scene_level_builds=int(s.scene_level_builds), debug=[int(x) for x in s.debug],
                    reference_order_forced=int(s.reference_order_forced), flat_leaves=int(s.flat_leaves),
                    rays_shadow_null=int(s.rays_shadow_null), rays_extend_grouped=int(s.rays_extend_grouped),
                    rays_shadow_grouped=int(s.rays_shadow_grouped))

    def prepare(self):
        """Validates the bound scene buffers and builds the traversal arrays now instead of inside the next draw."""
        check(lib.pt_pass_prepare(self.h))

    def set_traversal(self, mode):
        """0 = ordered traversal with exact fallback (default), 1 = reference order only."""
        check(lib.pt_pass_set_traversal(self.h, int(mode)))

    def set_wide_nodes(self, fmt):
        """Node format of the ordered traversal: -1 the library's choice (default), 0 fp32 boxes, 1 quantised."""
        check(lib.pt_pass_set_wide_nodes(self.h, int(fmt)))

    def set_pipeline(self, draws_in_flight):
        """How many draws may overlap on the device (1..16); outputs do not depend on it."""
        check(lib.pt_pass_set_pipeline(self.h, int(draws_in_flight)))

    def set_tail_threshold(self, threshold):
        """Paths alive below which one launch finishes the draw (0 = never, -1 = the library's choice)."""
        check(lib.pt_pass_set_tail_threshold(self.h, int(threshold)))

    def set_material_grouping(self, mode):
        """Grouping of paths inside the shade launches: -1 the library's choice (in a lone draw: escaped paths behind the
        ones that hit something, from the second vertex on), 0 none, 1 by material type at every vertex."""
        check(lib.pt_pass_set_material_grouping(self.h, int(mode)))

    def set_profiling(self, enable=True):
        check(lib.pt_pass_set_profiling(self.h, int(enable)))

    def profile(self, reset=False):
        pr = PtProfile()
        check(lib.pt_pass_profile(self.h, C.byref(pr), int(reset)))
        return {n: getattr(pr, n) for n, _ in PtProfile._fields_}

    def download(self):
        out = {}
        for name, (_, c) in OUTPUTS.items():
            out[name] = self.outputs[name].download(np.float32).reshape(self.height, self.width, c)
        return out

    def destroy(self):
        if self.h:
            lib.pt_pass_destroy(self.h)
            self.h = None


class Comm:
    """pt_comm: this rank's end of the RCCL communicator the frame is combined over (one rank per GPU)."""

    ID_BYTES = 128

    @staticmethod
    def unique_id():
        """bytes to hand to every rank (rank 0 makes them)."""
        buf = C.create_string_buffer(Comm.ID_BYTES)
        check(lib.pt_comm_unique_id(buf))
        return buf.raw

    def __init__(self, ctx, unique_id, rank, world):
        assert len(unique_id) == Comm.ID_BYTES
        h = C.c_void_p()
        check(lib.pt_comm_create(ctx.h, C.c_char_p(unique_id), int(rank), int(world), C.byref(h)))
        self.h = h
        self.rank, self.world = rank, world

    def allreduce(self, pass_, total_spp, last_rank):
        """pt_pass_allreduce: sums of the accumulated outputs over the ranks / total_spp, last-write
        outputs from last_rank; queued behind the draws, does not wait."""
        check(lib.pt_pass_allreduce(pass_.h, self.h, int(total_spp), int(last_rank)))

    def destroy(self):
        if self.h:
            lib.pt_comm_destroy(self.h)
            self.h = None


def trace_rays(pass_, rays):
    """Convenience: closest hits (HIT_DTYPE array) for host rays (n, 6)."""
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    rb = Buffer(pass_.ctx, rays)
    hb = Buffer(pass_.ctx, nbytes=len(rays) * HIT_DTYPE.itemsize)
    pass_.intersect(rb, len(rays), hb)
    hits = hb.download(np.uint8).view(HIT_DTYPE)
    rb.release()
    hb.release()
    return hits
