"""ctypes binding of include/raytracing_amd.h.

The shared library is built in-tree by raytracing_amd.build; importing this module
fails loudly when it is missing -- there is no Python or CPU fallback for the pass.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAYTRACING_AMD_LIB selects another build of the library (tuning variants made by build.py --variant)
LIB_PATH = os.environ.get("RAYTRACING_AMD_LIB") or os.path.join(_HERE, "libraytracing_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "raytracing_amd: %s is missing. Build it with `python -m raytracing_amd.build` "
        "(needs hipcc; cross-compiles for gfx950 without a GPU)." % LIB_PATH)

lib = C.CDLL(LIB_PATH)

c_void_pp = C.POINTER(C.c_void_p)
f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)


class PtParams(C.Structure):
    _fields_ = [
        ("jitter", C.c_float * 2),
        ("currentspp", C.c_uint32),
        ("skybox_Light_SUM", C.c_float),
        ("v2wMat", C.c_float * 16),
        ("SCREEN_W", C.c_int32), ("SCREEN_H", C.c_int32),
        ("WINDOW_W", C.c_int32), ("WINDOW_H", C.c_int32),
        ("MAX_DEPTH", C.c_int32),
        ("cameraNear", C.c_float), ("cameraFar", C.c_float),
        ("frameCounter", C.c_uint32),
        ("fov", C.c_float),
        ("SKY_W", C.c_int32), ("SKY_H", C.c_int32),
        ("backprojMat", C.c_float * 16),
        ("BRDFSampling", C.c_int32), ("SkyboxSampling", C.c_int32),
        ("SkyboxLighting", C.c_int32), ("RasterizaionFor1st", C.c_int32),
        ("lightCount", C.c_int32),
        ("accum_mode", C.c_int32),
        ("reserved", C.c_int32 * 7),
    ]


class PtStats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64), ("rays_extend", C.c_uint64), ("rays_shadow", C.c_uint64),
        ("draws", C.c_uint64), ("rays_retraced", C.c_uint64), ("mesh_level_builds", C.c_uint64),
                ("scene_level_builds", C.c_uint64), ("debug", C.c_uint64 * 3), ("reference_order_forced", C.c_uint64), ("flat_leaves", C.c_uint64),
                ("rays_shadow_null", C.c_uint64), ("rays_extend_grouped", C.c_uint64), ("rays_shadow_grouped", C.c_uint64),
    ]


class PtGlbInfo(C.Structure):
    _fields_ = [("n_meshes", C.c_int32), ("n_materials", C.c_int32), ("n_textures", C.c_int32), ("n_instances", C.c_int32),
                ("n_triangles", C.c_int64)]


class PtGlbMaterial(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("roughness", C.c_float), ("metallic", C.c_float), ("texture", C.c_int32)]


class PtProfile(C.Structure):
    _fields_ = [("extend_ms", C.c_double), ("shadow_ms", C.c_double), ("shade_ms", C.c_double),
                ("other_ms", C.c_double), ("intersect_ms", C.c_double),
                ("extend_launches", C.c_uint64), ("shadow_launches", C.c_uint64), ("shade_launches", C.c_uint64),
                ("other_launches", C.c_uint64), ("intersect_launches", C.c_uint64),
                ("trace_ms", C.c_double), ("trace_launches", C.c_uint64),
                ("tail_ms", C.c_double), ("tail_launches", C.c_uint64)]


# name -> (restype, argtypes); also the list the symbol-export test walks
SIGNATURES = {
    "pt_last_error": (C.c_char_p, []),
    "pt_version": (C.c_char_p, []),
    "pt_ctx_create": (C.c_int, [C.c_int, c_void_pp]),
    "pt_ctx_destroy": (C.c_int, [C.c_void_p]),
    "pt_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "pt_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "pt_buf_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, c_void_pp]),
    "pt_buf_release": (C.c_int, [C.c_void_p]),
    "pt_buf_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "pt_buf_copy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pt_buf_fill": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "pt_buf_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "pt_buf_bytes": (C.c_size_t, [C.c_void_p]),
    "pt_buf_device_ptr": (C.c_void_p, [C.c_void_p]),
    "pt_tex_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, c_void_pp]),
    "pt_tex_release": (C.c_int, [C.c_void_p]),
    "pt_tex_handle": (C.c_uint64, [C.c_void_p]),
    "pt_pass_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_void_pp]),
    "pt_pass_destroy": (C.c_int, [C.c_void_p]),
    "pt_pass_bind": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "pt_pass_bind_texture": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    "pt_pass_set_params": (C.c_int, [C.c_void_p, C.POINTER(PtParams)]),
    "pt_pass_draw": (C.c_int, [C.c_void_p]),
    "pt_pass_draw_batch": (C.c_int, [C.c_void_p, C.POINTER(PtParams), C.c_int]),
    "pt_pass_reserve_batch": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_pass_output": (C.c_int, [C.c_void_p, C.c_int, c_void_pp]),
    "pt_pass_stats": (C.c_int, [C.c_void_p, C.POINTER(PtStats), C.c_int]),
    "pt_pass_set_traversal": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_pass_set_wide_nodes": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_pass_prepare": (C.c_int, [C.c_void_p]),
    "pt_pass_set_pipeline": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_pass_set_tail_threshold": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_pass_set_material_grouping": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_post_merge": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pt_post_static_blend": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pt_post_tonemap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pt_post_svgf_temporal": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 12),
    "pt_post_svgf_variance": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5),
    "pt_post_svgf_spatial": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5),
    "pt_post_taa": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 8),
    "pt_buf_save_as_image": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p]),
    "pt_host_quantize_image": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "pt_host_write_png": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "pt_pass_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_pass_profile": (C.c_int, [C.c_void_p, C.POINTER(PtProfile), C.c_int]),
    "pt_pass_intersect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pt_pass_intersect_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pt_comm_unique_id": (C.c_int, [C.c_void_p]),
    "pt_comm_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, c_void_pp]),
    "pt_comm_create_all": (C.c_int, [c_void_pp, C.c_int, c_void_pp]),
    "pt_comm_destroy": (C.c_int, [C.c_void_p]),
    "pt_comm_rank": (C.c_int, [C.c_void_p]),
    "pt_comm_world": (C.c_int, [C.c_void_p]),
    "pt_pass_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]),
    "pt_pass_allreduce_all": (C.c_int, [c_void_pp, c_void_pp, C.c_int, C.c_uint32, C.c_int]),
    "pt_host_glb_open": (C.c_int, [C.c_char_p, c_void_pp]),
    "pt_host_glb_open_memory": (C.c_int, [C.c_void_p, C.c_size_t, c_void_pp]),
    "pt_host_glb_close": (C.c_int, [C.c_void_p]),
    "pt_host_glb_info": (C.c_int, [C.c_void_p, C.POINTER(PtGlbInfo)]),
    "pt_host_glb_meshes": (C.c_int, [C.c_void_p, i32p, C.c_void_p]),
    "pt_host_glb_materials": (C.c_int, [C.c_void_p, C.POINTER(PtGlbMaterial)]),
    "pt_host_glb_texture": (C.c_int, [C.c_void_p, C.c_int32, c_void_pp, C.POINTER(C.c_size_t), i32p]),
    "pt_host_glb_instances": (C.c_int, [C.c_void_p, f32p, i32p, i32p, f32p]),
    "pt_host_glb_material_stream": (C.c_int64, [C.c_void_p, f32p, C.c_size_t, i32p]),
    "pt_host_build_mesh_bvh": (C.c_int, [C.c_void_p, i32p, C.c_int32, C.c_int32, C.c_void_p, f32p, i32p, i32p]),
    "pt_host_check_traversal_layout": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                 C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "pt_host_build_scene": (C.c_int, [f32p, i32p, i32p, C.c_int32, f32p, i32p, i32p, C.c_int32,
                                      C.c_void_p, C.c_void_p, i32p]),
    "pt_host_check_flat_layout": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.c_int32, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]),
    "pt_host_transform_matrix": (C.c_int, [f32p, f32p, f32p, f32p]),
    "pt_host_mat4_mul": (C.c_int, [f32p, f32p, f32p]),
    "pt_host_mat4_inverse": (C.c_int, [f32p, f32p]),
    "pt_host_camera_projection": (C.c_int, [C.c_float, C.c_float, C.c_int32, C.c_int32, f32p]),
    "pt_host_frame_jitters": (C.c_int, [C.c_uint64, C.c_int32, C.c_int32, f32p]),
    "pt_host_rgbe_decode": (C.c_int, [C.c_void_p, C.c_size_t, i32p, i32p, f32p]),
    "pt_host_rgbe_encode": (C.c_int64, [f32p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t]),
    "pt_host_sky_cache": (C.c_int, [f32p, C.c_int32, C.c_int32, f32p, C.POINTER(C.c_double)]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


class PtError(RuntimeError):
    pass


def check(code):
    """Raises PtError with the library's message when code is not PT_OK."""
    if code < 0:
        raise PtError("%s (status %d)" % (lib.pt_last_error().decode("utf-8", "replace"), code))
    return code
