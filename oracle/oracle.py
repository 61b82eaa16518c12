"""TEST INFRASTRUCTURE ONLY: ctypes loader for oracle/liboracle.so (the CPU restatement
of the reference's path-tracing pass, oracle/pt_oracle.cpp) and for the compiled
reference harness oracle/_ref/ref_harness.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes as C
import os
import struct
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_HARNESS = os.path.join(_HERE, "_ref", "ref_harness")


def build(verbose=False):
    """Compiles liboracle.so (and _ref/ref_harness when the reference tree exists)."""
    r = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(r.stdout)


class OrcTexture(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("channels", C.c_int32), ("format", C.c_int32)]


class OrcScene(C.Structure):
    _fields_ = [("materials", C.c_void_p), ("triangles", C.c_void_p), ("instances", C.c_void_p),
                ("mesh_nodes", C.c_void_p), ("scene_nodes", C.c_void_p), ("lights", C.c_void_p),
                ("textures", C.POINTER(OrcTexture)), ("n_textures", C.c_int32),
                ("skybox", OrcTexture), ("skybox_samplecache", OrcTexture),
                ("depthGBufferTexture", OrcTexture), ("normalGBufferTexture", OrcTexture),
                ("uvGBufferTexture", OrcTexture), ("instanceIndexGBufferTexture", OrcTexture),
                ("n_instances", C.c_int32)]


class OrcOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("directLum", "indirectLum", "albedo", "motion", "depth", "normal", "instanceIndex")]


COUNTER_FIELDS = ("rays", "scene_nodes", "mesh_inner_nodes", "leaf_tests", "hits", "max_stack", "samples",
                  "rays_extend", "rays_shadow", "shaded_vertices", "material_evals", "ldr_tex_fetches", "sky_fetches",
                  "ext_nodes", "ext_leaf_tests")


class OrcCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_FIELDS}


HIT_DTYPE = np.dtype([("instance", "<i4"), ("triangle", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_sobol.restype = C.c_float
        _lib.orc_sobol.argtypes = [C.c_uint32, C.c_uint32]
        _lib.orc_skybox_sample_pdf.restype = C.c_float
        _lib.orc_intersect.restype = None
        _lib.orc_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        _lib.orc_render.restype = None
        _lib.orc_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.orc_post_merge.restype = None
        _lib.orc_post_merge.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_post_static_blend.restype = None
        _lib.orc_post_static_blend.argtypes = [C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_post_tonemap.restype = None
        _lib.orc_post_tonemap.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p]
        for name, n_int, n_ptr in (("orc_svgf_temporal", 3, 12), ("orc_svgf_variance", 2, 5), ("orc_svgf_spatial", 3, 5), ("orc_taa", 2, 8)):
            f = getattr(_lib, name)
            f.restype = None
            f.argtypes = [C.c_int] * n_int + [C.c_void_p] * n_ptr
        _lib.orc_quantize_image.restype = None
        _lib.orc_quantize_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return _lib


def _tex(arr, fmt):
    t = OrcTexture()
    t.data = arr.ctypes.data
    t.height, t.width, t.channels = arr.shape
    t.format = fmt
    return t


class OracleScene:
    """Keeps the numpy arrays alive next to the C struct that points into them."""

    def __init__(self, scene_data, sky=None):
        sd = scene_data
        self._keep = [sd, sky]
        s = OrcScene()
        s.materials = sd.materials.ctypes.data
        s.triangles = sd.triangles.ctypes.data
        s.instances = sd.instances.ctypes.data
        s.mesh_nodes = sd.mesh_nodes.ctypes.data
        s.scene_nodes = sd.scene_nodes.ctypes.data
        s.lights = sd.lights.ctypes.data if len(sd.lights) else None
        self._tex = (OrcTexture * max(1, len(sd.textures)))()
        for i, t in enumerate(sd.textures):
            self._tex[i] = _tex(t, 0)
        s.textures = self._tex
        s.n_textures = len(sd.textures)
        if sky is not None:
            s.skybox = _tex(sky.rgb, 1)
            s.skybox_samplecache = _tex(sky.cache, 1)
        s.n_instances = len(sd.instances)
        self.c = s
        self.scene_data = sd
        self.sky = sky

    def set_gbuffer(self, depth, normal, uv, instance_index):
        """The raster pass's four float images (h x w x 1/3/2/1) for RasterizaionFor1st renders."""
        imgs = [np.ascontiguousarray(a, np.float32) for a in (depth, normal, uv, instance_index)]
        imgs = [a.reshape(a.shape[0], a.shape[1], -1) for a in imgs]
        self._keep.append(imgs)
        self.c.depthGBufferTexture = _tex(imgs[0], 1)
        self.c.normalGBufferTexture = _tex(imgs[1], 1)
        self.c.uvGBufferTexture = _tex(imgs[2], 1)
        self.c.instanceIndexGBufferTexture = _tex(imgs[3], 1)

    def intersect(self, rays, n_threads=0):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        hits = np.zeros(len(rays), HIT_DTYPE)
        cnt = OrcCounters()
        lib().orc_intersect(C.addressof(self.c), rays.ctypes.data, len(rays), hits.ctypes.data, C.addressof(cnt), n_threads)
        return hits, cnt.as_dict()

    def render(self, params, accum=None, n_threads=0):
        """One sample per pixel.  params: any ctypes struct laid out like pt_params.
        accum: dict of the previous outputs (needed when currentspp != 1).
        Returns (dict of 7 float32 arrays, counters dict)."""
        w, h = params.SCREEN_W, params.SCREEN_H
        shapes = dict(directLum=3, indirectLum=3, albedo=3, motion=2, depth=1, normal=3, instanceIndex=1)
        out = {}
        for k, c in shapes.items():
            if accum is not None and k in accum:
                out[k] = np.ascontiguousarray(accum[k], np.float32).copy()
            else:
                out[k] = np.zeros((h, w, c), np.float32)
        o = OrcOutputs()
        for k in shapes:
            setattr(o, k, out[k].ctypes.data)
        cnt = OrcCounters()
        lib().orc_render(C.addressof(self.c), C.addressof(params), C.addressof(o), C.addressof(cnt), n_threads)
        return out, cnt.as_dict()


def post_merge(di, idi, albedo):
    """SVGF_Merge: (di + idi) * albedo with the NaN/Inf replacement; arrays of 3 floats per pixel."""
    di, idi, albedo = (np.ascontiguousarray(a, np.float32).ravel() for a in (di, idi, albedo))
    out = np.zeros_like(di)
    lib().orc_post_merge(di.size // 3, di.ctypes.data, idi.ctypes.data, albedo.ctypes.data, out.ctypes.data)
    return out


def post_static_blend(frame, color, hist_color, hist_moment):
    """StaticBlender: returns the updated (history colour, history moment)."""
    color = np.ascontiguousarray(color, np.float32).ravel()
    hc = np.array(hist_color, np.float32).ravel().copy()
    hm = np.array(hist_moment, np.float32).ravel().copy()
    lib().orc_post_static_blend(color.size // 3, int(frame), color.ctypes.data, hc.ctypes.data, hm.ctypes.data)
    return hc, hm


def post_tonemap(color):
    color = np.ascontiguousarray(color, np.float32).ravel()
    out = np.zeros_like(color)
    lib().orc_post_tonemap(color.size, color.ctypes.data, out.ctypes.data)
    return out


def quantize_image(data, w, h, c):
    """save_as_image's bytes: (h, w, c) uint8, top row first."""
    data = np.ascontiguousarray(data, np.float32).ravel()
    assert data.size == w * h * c
    out = np.zeros(w * h * c, np.uint8)
    lib().orc_quantize_image(data.ctypes.data, w, h, c, out.ctypes.data)
    return out.reshape(h, w, c)


def _f32(a):
    return np.ascontiguousarray(a, np.float32).ravel()


def svgf_temporal(w, h, first_frame, color, normal, inst, motion, h_color, h_moment, h_num, h_normal, h_inst):
    """SVGF_TemporalFilter.glsl: returns (color, moment, num_samples)."""
    ins = [_f32(a) for a in (color, normal, inst, motion)]
    hist = [_f32(a) for a in (h_color, h_moment, h_num, h_normal, h_inst)]
    oc, om, on = np.zeros(w * h * 3, np.float32), np.zeros(w * h * 2, np.float32), np.zeros(w * h, np.float32)
    lib().orc_svgf_temporal(w, h, int(first_frame), *[a.ctypes.data for a in ins], oc.ctypes.data, om.ctypes.data, on.ctypes.data,
                            *[a.ctypes.data for a in hist])
    return oc, om, on


def svgf_variance(w, h, moment, normal, depth, num):
    ins = [_f32(a) for a in (moment, normal, depth, num)]
    out = np.zeros(w * h, np.float32)
    lib().orc_svgf_variance(w, h, *[a.ctypes.data for a in ins], out.ctypes.data)
    return out


def svgf_spatial(w, h, step, color, normal, depth, variance):
    ins = [_f32(a) for a in (color, normal, depth, variance)]
    out = np.zeros(w * h * 3, np.float32)
    lib().orc_svgf_spatial(w, h, int(step), *[a.ctypes.data for a in ins], out.ctypes.data)
    return out


def taa(w, h, color, h_color, motion, normal, inst, h_normal, h_inst):
    ins = [_f32(a) for a in (color, h_color, motion, normal, inst, h_normal, h_inst)]
    out = np.zeros(w * h * 3, np.float32)
    lib().orc_taa(w, h, *[a.ctypes.data for a in ins], out.ctypes.data)
    return out


def algorithmic_bytes_per_ray(cnt):
    """Bytes the REFERENCE data structures make one closest-hit query touch, from the
    reference-order visit counts: 48 B per node popped (scene, mesh inner and leaf nodes),
    36 B of vertex positions per triangle test, 24 B ray in, 12 B hit out."""
    rays = max(1, cnt["rays"])
    nodes = cnt["scene_nodes"] + cnt["mesh_inner_nodes"] + cnt["leaf_tests"]
    return (48.0 * nodes + 36.0 * cnt["leaf_tests"]) / rays + 24.0 + 12.0


def algorithmic_bytes_per_extend_ray(cnt):
    """Same figure restricted to the closest-hit rays of a render (primary + indirect)."""
    rays = max(1, cnt["rays_extend"])
    return (48.0 * cnt["ext_nodes"] + 36.0 * cnt["ext_leaf_tests"]) / rays + 24.0 + 12.0


# ------------------------------------------------------------------ compiled reference
def have_ref():
    return os.path.exists(REF_HARNESS)


def _write_meshes(path, scene_data):
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(scene_data.tri_count)))
        base = 0
        for n in scene_data.tri_count:
            f.write(struct.pack("<i", int(n)))
            f.write(np.ascontiguousarray(scene_data.triangles[base:base + n, :9]).tobytes())
            base += int(n)


def ref_mesh_bvh(scene_data):
    """Runs the reference's BVHNode::build on every mesh; returns (nodes bytes as uint8
    array of 48-byte records, siz per mesh, depth per mesh)."""
    with tempfile.TemporaryDirectory() as d:
        _write_meshes(d + "/m.bin", scene_data)
        subprocess.check_call([REF_HARNESS, "bvh", d + "/m.bin", d + "/o.bin"])
        raw = open(d + "/o.bin", "rb").read()
    n, = struct.unpack_from("<i", raw, 0)
    nodes = np.frombuffer(raw, np.uint8, n * 48, 4).copy()
    st = np.frombuffer(raw, np.int32, 2 * len(scene_data.tri_count), 4 + n * 48).reshape(-1, 2)
    return nodes, st[:, 0].copy(), st[:, 1].copy()


def ref_intersect(scene_data, rays):
    """Closest hits from the reference's CPU picking routine (BVHNode::rayIntersect)."""
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    with tempfile.TemporaryDirectory() as d:
        _write_meshes(d + "/m.bin", scene_data)
        with open(d + "/s.bin", "wb") as f:
            f.write(struct.pack("<i", len(scene_data.instances)))
            for i in range(len(scene_data.instances)):
                f.write(np.ascontiguousarray(scene_data.l2w[i], np.float32).tobytes())
                f.write(struct.pack("<i", int(scene_data.instances["meshIndex"][i])))
        with open(d + "/r.bin", "wb") as f:
            f.write(struct.pack("<i", len(rays)))
            f.write(rays.tobytes())
        subprocess.check_call([REF_HARNESS, "isect", d + "/m.bin", d + "/s.bin", d + "/r.bin", d + "/o.bin"])
        raw = open(d + "/o.bin", "rb").read()
    return np.frombuffer(raw, np.dtype([("instance", "<i4"), ("triangle", "<i4"), ("t", "<f4")]), len(rays)).copy()


def ref_intersect_scene(scene_data, nodes, rays):
    """The reference's CPU picking path run as the application runs it (main.cpp:125-146): real Instance / Transform
    objects in a hierarchy, the scene BVH of ResourceManager::reload_scene over them, BVHNode::rayIntersect entered at
    the SCENE root (src/BVH.cpp:89-114, scene-leaf branch included).  scene_data supplies the meshes (tri_count,
    triangles); nodes: list of (position, rotation_deg, scale, mesh or -1, parent or -1) with parents before children.
    Returns (hits, l2w (n_inst, 16), mesh_index (n_inst,)): the instances are the nodes with a mesh in pre-order and
    l2w the matrices the reference's hierarchy walk accumulated for them."""
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    with tempfile.TemporaryDirectory() as d:
        _write_meshes(d + "/m.bin", scene_data)
        with open(d + "/n.bin", "wb") as f:
            f.write(struct.pack("<i", len(nodes)))
            for pos, rot, scl, mesh, parent in nodes:
                f.write(np.asarray(list(pos) + list(rot) + list(scl), np.float32).tobytes())
                f.write(struct.pack("<2i", int(mesh), int(parent)))
        with open(d + "/r.bin", "wb") as f:
            f.write(struct.pack("<i", len(rays)))
            f.write(rays.tobytes())
        subprocess.check_call([REF_HARNESS, "isect_scene", d + "/m.bin", d + "/n.bin", d + "/r.bin", d + "/o.bin"])
        raw = open(d + "/o.bin", "rb").read()
    n_inst, = struct.unpack_from("<i", raw, 0)
    hits = np.frombuffer(raw, np.dtype([("instance", "<i4"), ("triangle", "<i4"), ("t", "<f4")]), len(rays), 4).copy()
    rec = np.frombuffer(raw, np.dtype([("l2w", "<f4", 16), ("mesh", "<i4")]), n_inst, 4 + 12 * len(rays))
    return hits, rec["l2w"].copy(), rec["mesh"].copy()


def ref_scene_level(scene_data):
    """The reference's scene level for the instances of scene_data (ResourceManager::reload_scene,
    reload_sceneBVH, reload_instances run over its own AABB / glm / BVHNode::build): returns
    (scene nodes as uint8 array of 48-byte records, world2local (n_inst, 16), siz, depth)."""
    n_inst = len(scene_data.instances)
    with tempfile.TemporaryDirectory() as d:
        _write_meshes(d + "/m.bin", scene_data)
        with open(d + "/s.bin", "wb") as f:
            f.write(struct.pack("<i", n_inst))
            for i in range(n_inst):
                f.write(np.ascontiguousarray(scene_data.l2w[i], np.float32).tobytes())
                f.write(struct.pack("<i", int(scene_data.mesh_index[i])))
        subprocess.check_call([REF_HARNESS, "scene", d + "/m.bin", d + "/s.bin", d + "/o.bin"])
        raw = open(d + "/o.bin", "rb").read()
    n, = struct.unpack_from("<i", raw, 0)
    nodes = np.frombuffer(raw, np.uint8, n * 48, 4).copy()
    w2l = np.frombuffer(raw, np.float32, n_inst * 16, 4 + n * 48).reshape(n_inst, 16).copy()
    siz, depth = struct.unpack_from("<2i", raw, 4 + n * 48 + n_inst * 64)
    return nodes, w2l, siz, depth


def ref_transforms(records):
    """records: (n, 20) float32 rows {fov_x, far, parent position/rotation_deg/scale, camera position/rotation_deg/
    scale}.  Runs the reference's Transform::matrix (src/Transform.cpp:20-53), Instance::matrix_to_global
    (src/instance/Instance.cpp:38-41) and Camera::w2v_matrix / v2w_matrix / projection (src/instance/Camera.cpp:13-43)
    and returns ((screen_w, screen_h), (n, 6, 16) float32): parent matrix, camera matrix, camera to global, w2v, v2w,
    projection, each as glm stores it (column major)."""
    records = np.ascontiguousarray(records, np.float32).reshape(-1, 20)
    with tempfile.TemporaryDirectory() as d:
        with open(d + "/i.bin", "wb") as f:
            f.write(struct.pack("<i", len(records)))
            f.write(records.tobytes())
        subprocess.check_call([REF_HARNESS, "transform", d + "/i.bin", d + "/o.bin"])
        raw = open(d + "/o.bin", "rb").read()
    wh = struct.unpack_from("<2i", raw, 0)
    return wh, np.frombuffer(raw, np.float32, len(records) * 96, 8).reshape(-1, 6, 16).copy()


def ref_mat_inverse(mats):
    mats = np.ascontiguousarray(mats, np.float32).reshape(-1, 16)
    with tempfile.TemporaryDirectory() as d:
        with open(d + "/i.bin", "wb") as f:
            f.write(struct.pack("<i", len(mats)))
            f.write(mats.tobytes())
        subprocess.check_call([REF_HARNESS, "mat", d + "/i.bin", d + "/o.bin"])
        raw = np.fromfile(d + "/o.bin", np.float32)
    return raw[:len(mats) * 16].reshape(-1, 16), raw[len(mats) * 16:]


def ref_rgbe_component_table():
    """convertComponent(exponent byte - 128, mantissa byte) of include/hdrloader.cpp for all
    256 x 256 inputs (the loader's entry point itself is not runnable on LP64)."""
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call([REF_HARNESS, "hdr", d + "/o.bin"])
        return np.fromfile(d + "/o.bin", np.float32).reshape(256, 256)


def ref_stb_decode(image_bytes):
    with tempfile.TemporaryDirectory() as d:
        open(d + "/i.img", "wb").write(image_bytes)
        subprocess.check_call([REF_HARNESS, "stb", d + "/i.img", d + "/o.bin"])
        raw = open(d + "/o.bin", "rb").read()
    w, h, c = struct.unpack_from("<iii", raw, 0)
    return np.frombuffer(raw, np.uint8, w * h * c, 12).reshape(h, w, c).copy()
