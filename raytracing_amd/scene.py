"""Host-side scene assembly for tests, smoke and bench.

Produces the seven scene arrays in the reference's SSBO byte layouts by calling the
library's host helpers (pt_host_*), mirroring what lxtyin/Raytracing does between
AssimpLoader::load_model and ResourceManager::reload_scene:

  src/tool/loader.cpp:44-158      glTF material/mesh/node mapping  -> load_glb, casa_scene
  src/ResourceManager.cpp:47-77   hierarchy walk, instance order   -> SceneBuilder.add_instance order
  src/ResourceManager.cpp:95-224  buffers                          -> SceneBuilder.finish
  src/texture/Skybox.cpp:12-53    sky sample cache                 -> Sky
  main.cpp:376-388, 169-205       demo camera pose and uniforms    -> casa_camera, make_params

Nothing in this module touches a GPU.
"""
import ctypes as C
import io
import json
import math
import os
import struct

import numpy as np

from . import _capi
from ._capi import PtParams, check, lib

NODE_DTYPE = np.dtype([("aa", "<f4", 4), ("bb", "<f4", 4), ("ls", "<i4"), ("rs", "<i4"),
                       ("inst", "<i4"), ("tri", "<i4")])
INSTANCE_DTYPE = np.dtype([("w2l", "<f4", 16), ("materialPtr", "<i4"), ("meshIndex", "<i4"), ("pad", "<i4", 10)])
LIGHT_DTYPE = np.dtype([("type", "<i4"), ("pos", "<f4", 3), ("rgb", "<f4", 3), ("pad", "<f4")])
assert NODE_DTYPE.itemsize == 48 and INSTANCE_DTYPE.itemsize == 112 and LIGHT_DTYPE.itemsize == 32


def _f32p(a):
    return a.ctypes.data_as(_capi.f32p)


def _i32p(a):
    return a.ctypes.data_as(_capi.i32p)


def transform_matrix(position=(0, 0, 0), rotation_deg=(0, 0, 0), scale=(1, 1, 1)):
    """Transform::matrix (src/Transform.cpp:20-53); returns 16 floats, column major."""
    p = np.asarray(position, np.float32)
    r = np.asarray(rotation_deg, np.float32)
    s = np.asarray(scale, np.float32)
    out = np.zeros(16, np.float32)
    check(lib.pt_host_transform_matrix(_f32p(p), _f32p(r), _f32p(s), _f32p(out)))
    return out


def mat4_mul(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.zeros(16, np.float32)
    check(lib.pt_host_mat4_mul(_f32p(a), _f32p(b), _f32p(out)))
    return out


def mat4_inverse(m):
    m = np.ascontiguousarray(m, np.float32)
    out = np.zeros(16, np.float32)
    check(lib.pt_host_mat4_inverse(_f32p(m), _f32p(out)))
    return out


def camera_projection(fov_x, far_plane, w, h):
    out = np.zeros(16, np.float32)
    check(lib.pt_host_camera_projection(fov_x, far_plane, w, h, _f32p(out)))
    return out


IDENTITY = np.eye(4, dtype=np.float32).reshape(16)


class SceneData:
    """The scene buffers as numpy arrays, in upload order (slots 0-6)."""

    def __init__(self):
        self.textures = []          # list of uint8 arrays (h, w, c); slot 0 holds their handles
        self.materials = None       # float32 stream                       slot 1
        self.triangles = None       # float32 (n, 24)                      slot 2
        self.instances = None       # INSTANCE_DTYPE                       slot 3
        self.mesh_nodes = None      # NODE_DTYPE                           slot 4
        self.scene_nodes = None     # NODE_DTYPE                           slot 5
        self.lights = None          # LIGHT_DTYPE                          slot 6
        self.tri_count = None       # per mesh
        self.mesh_root_box = None   # (n_mesh, 6)
        self.mesh_siz = None
        self.mesh_depth = None
        self.bvh_siz = 0            # two-level tree, as the reference prints (main.cpp:393-394)
        self.bvh_depth = 0
        self.l2w = None             # (n_inst, 16)

    def world_box(self):
        """World-space box of the whole scene = the scene-BVH root box."""
        return (self.scene_nodes["aa"][0][:3].copy(), self.scene_nodes["bb"][0][:3].copy())


class SceneBuilder:
    """Collects meshes, materials, instances and lights, then emits SceneData.

    Meshes are registered in call order (ResourceManager::add_mesh); instances are given
    in hierarchy pre-order with their accumulated local-to-world matrix
    (update_globalinstance_recursive); one material record is appended per instance,
    not per material (ResourceManager.cpp:149-158).
    """

    def __init__(self):
        self._meshes = []
        self._instances = []   # (l2w, mesh, material dict)
        self._lights = []
        self._textures = []

    def add_texture(self, pixels):
        pixels = np.ascontiguousarray(pixels, np.uint8)
        assert pixels.ndim == 3 and pixels.shape[2] in (1, 3, 4)
        self._textures.append(pixels)
        return len(self._textures) - 1

    def add_mesh(self, triangles):
        """triangles: float32 (n, 24): 9 positions, 9 normals, 6 uvs (Triangle.h:15-27)."""
        t = np.ascontiguousarray(triangles, np.float32).reshape(-1, 24)
        assert len(t) > 0
        self._meshes.append(t)
        return len(self._meshes) - 1

    def add_instance(self, mesh, l2w=IDENTITY, material=None):
        """material: dict(type=1|2, albedo, roughness, metallic, tex=-1, ior=1.5)."""
        m = dict(type=1, albedo=(1.0, 1.0, 1.0), roughness=1.0, metallic=0.0, tex=-1, ior=1.5)
        if material:
            m.update(material)
        self._instances.append((np.ascontiguousarray(l2w, np.float32).reshape(16), int(mesh), m))
        return len(self._instances) - 1

    def add_point_light(self, l2w, emission):
        l2w = np.asarray(l2w, np.float32).reshape(4, 4)       # rows of this array = matrix columns
        self._lights.append((0, l2w[3, :3].copy(), np.asarray(emission, np.float32)))

    def add_directional_light(self, l2w, emission):
        l2w = np.asarray(l2w, np.float32).reshape(4, 4)
        self._lights.append((1, l2w[2, :3].copy(), np.asarray(emission, np.float32)))

    def finish(self, n_threads=0):
        sd = SceneData()
        sd.textures = list(self._textures)
        n_mesh = len(self._meshes)
        sd.tri_count = np.array([len(m) for m in self._meshes], np.int32)
        sd.triangles = np.ascontiguousarray(np.concatenate(self._meshes, axis=0), np.float32)
        n_nodes = int((2 * sd.tri_count.astype(np.int64) - 1).sum())
        sd.mesh_nodes = np.zeros(n_nodes, NODE_DTYPE)
        sd.mesh_root_box = np.zeros((n_mesh, 6), np.float32)
        sd.mesh_siz = np.zeros(n_mesh, np.int32)
        sd.mesh_depth = np.zeros(n_mesh, np.int32)
        check(lib.pt_host_build_mesh_bvh(sd.triangles.ctypes.data, _i32p(sd.tri_count), n_mesh, n_threads,
                                         sd.mesh_nodes.ctypes.data, _f32p(sd.mesh_root_box),
                                         _i32p(sd.mesh_siz), _i32p(sd.mesh_depth)))
        # material stream: one record per instance (RoughConductor.cpp:26-38, RoughDielectric.cpp:23-36)
        stream = []
        mat_ptr = []
        for _, _, m in self._instances:
            mat_ptr.append(len(stream))
            a = m["albedo"]
            stream += [float(m["type"]), a[0], a[1], a[2], m["roughness"], m["metallic"],
                       float(m["tex"]) if m["tex"] >= 0 else -1.0]
            if m["type"] == 2:
                stream.append(m["ior"])
        sd.materials = np.array(stream, np.float32)
        n_inst = len(self._instances)
        sd.l2w = np.ascontiguousarray(np.stack([i[0] for i in self._instances]), np.float32)
        mesh_index = np.array([i[1] for i in self._instances], np.int32)
        material_ptr = np.array(mat_ptr, np.int32)
        sd.instances = np.zeros(n_inst, INSTANCE_DTYPE)
        sd.scene_nodes = np.zeros(2 * n_inst - 1, NODE_DTYPE)
        stats = np.zeros(2, np.int32)
        check(lib.pt_host_build_scene(_f32p(sd.mesh_root_box), _i32p(sd.mesh_siz), _i32p(sd.mesh_depth), n_mesh,
                                      _f32p(sd.l2w), _i32p(mesh_index), _i32p(material_ptr), n_inst,
                                      sd.instances.ctypes.data, sd.scene_nodes.ctypes.data, _i32p(stats)))
        sd.bvh_siz, sd.bvh_depth = int(stats[0]), int(stats[1])
        sd.mesh_index, sd.material_ptr = mesh_index, material_ptr
        sd.lights = np.zeros(len(self._lights), LIGHT_DTYPE)
        for k, (t, p, e) in enumerate(self._lights):
            sd.lights[k] = (t, p, e, 0.0)
        return sd


# --------------------------------------------------------------------------------------
# glTF binary reader reproducing the observable mapping of src/tool/loader.cpp
# --------------------------------------------------------------------------------------
_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _accessor(js, blob, idx):
    acc = js["accessors"][idx]
    bv = js["bufferViews"][acc["bufferView"]]
    dt = np.dtype(_COMP[acc["componentType"]]).newbyteorder("<")
    nc = _NCOMP[acc["type"]]
    off = bv.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = bv.get("byteStride", 0) or dt.itemsize * nc
    cnt = acc["count"]
    if stride == dt.itemsize * nc:
        a = np.frombuffer(blob, dt, cnt * nc, off).reshape(cnt, nc)
    else:
        a = np.lib.stride_tricks.as_strided(np.frombuffer(blob, dt, offset=off, count=(stride // dt.itemsize) * (cnt - 1) + nc),
                                            (cnt, nc), (stride, dt.itemsize))
    return np.array(a)


def decode_image(data, index=0):
    """Decodes an embedded image to uint8 (h, w, c), rows top to bottom, channels as
    stored -- what stbi_load_from_memory(..., 0) returns in loader.cpp:23-27."""
    from PIL import Image  # Pillow ships in the image; libjpeg's IDCT may differ from stb_image's by +-1..2 levels
    im = Image.open(io.BytesIO(data))
    if im.mode not in ("L", "RGB", "RGBA"):
        im = im.convert("RGB")
    a = np.asarray(im, np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return np.ascontiguousarray(a)


def load_glb(path, image_decoder=decode_image):
    """Returns (meshes, materials, textures, instances) for a .glb file.

    meshes:    list of float32 (n, 24) triangle arrays, one per glTF mesh primitive, in
               mesh order; triangle order = index order; normals as stored; uv as stored
               (assimp's internal V flip and aiProcess_FlipUVs cancel, loader.cpp:139-142).
    materials: list of dicts; every material becomes a RoughConductor with
               albedo = baseColorFactor rgb, roughness = max(0.001, roughnessFactor),
               metallic = 1.0 because the metallic key is always present in a glTF import
               (loader.cpp:53-65); tex = index into textures or -1.
    instances: list of (node name, mesh index, material index, l2w) in pre-order, the
               root node's own transform reset to identity (loader.cpp:152-156).
    """
    d = open(path, "rb").read()
    magic, ver, total = struct.unpack_from("<III", d, 0)
    assert magic == 0x46546C67 and ver == 2, "not a glTF 2.0 binary"
    clen, ctype = struct.unpack_from("<II", d, 12)
    js = json.loads(d[20:20 + clen])
    boff = 20 + clen
    blen, btype = struct.unpack_from("<II", d, boff)
    blob = d[boff + 8: boff + 8 + blen]

    textures = []
    tex_of_image = {}
    materials = []
    for m in js.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        col = pbr.get("baseColorFactor", [1, 1, 1, 1])
        mat = dict(type=1, name=m.get("name", ""),
                   albedo=(np.float32(col[0]), np.float32(col[1]), np.float32(col[2])),
                   roughness=max(np.float32(0.001), np.float32(pbr.get("roughnessFactor", 1.0))),
                   metallic=np.float32(1.0), tex=-1)
        bt = pbr.get("baseColorTexture")
        if bt is not None:
            src = js["textures"][bt["index"]]["source"]
            # loader.cpp:63-65 creates a new Texture per material that references one
            img = js["images"][src]
            bv = js["bufferViews"][img["bufferView"]]
            raw = blob[bv.get("byteOffset", 0): bv.get("byteOffset", 0) + bv["byteLength"]]
            textures.append(image_decoder(raw, src))
            mat["tex"] = len(textures) - 1
        materials.append(mat)

    meshes = []
    mesh_material = []
    first_prim_of_mesh = []
    for m in js["meshes"]:
        first_prim_of_mesh.append(len(meshes))
        assert len(m["primitives"]) == 1, "the reference asserts one mesh per node (loader.cpp:116)"
        for p in m["primitives"]:
            assert p.get("mode", 4) == 4
            pos = _accessor(js, blob, p["attributes"]["POSITION"]).astype(np.float32)
            nor = _accessor(js, blob, p["attributes"]["NORMAL"]).astype(np.float32)
            if "TEXCOORD_0" in p["attributes"]:
                uv = _accessor(js, blob, p["attributes"]["TEXCOORD_0"]).astype(np.float32)
            else:
                uv = np.zeros((len(pos), 2), np.float32)
            idx = _accessor(js, blob, p["indices"]).astype(np.int64).reshape(-1, 3)
            tri = np.concatenate([pos[idx].reshape(-1, 9), nor[idx].reshape(-1, 9), uv[idx].reshape(-1, 6)], axis=1)
            meshes.append(np.ascontiguousarray(tri, np.float32))
            mesh_material.append(p.get("material", -1))

    instances = []

    def node_matrix(n):
        if "matrix" in n:
            return np.asarray(n["matrix"], np.float32)
        t = n.get("translation", (0, 0, 0))
        s = n.get("scale", (1, 1, 1))
        assert "rotation" not in n, "quaternion node rotations are not mapped yet"
        return transform_matrix(t, (0, 0, 0), s)

    def walk(ni, parent, is_root):
        n = js["nodes"][ni]
        local = IDENTITY if is_root else node_matrix(n)
        world = mat4_mul(parent, local)
        if "mesh" in n:
            mi = first_prim_of_mesh[n["mesh"]]
            instances.append((n.get("name", ""), mi, mesh_material[mi], world))
        for c in n.get("children", []):
            walk(c, world, False)

    roots = js["scenes"][js.get("scene", 0)]["nodes"]
    assert len(roots) == 1, "multi-root scenes get a synthetic assimp root; not mapped yet"
    walk(roots[0], IDENTITY, True)
    return meshes, materials, textures, instances


def casa_scene(glb_path, n_threads=0, image_decoder=None):
    """The demo scene of main.cpp:380-383: casa_obj.glb under a root rotated (-90, 0, 0).

    The embedded JPEG is taken from casa_textures.npz next to the .glb when that file
    exists: it holds the pixels the reference's own decoder (stb_image) produces, which
    differ from libjpeg's by a level or two in places."""
    if image_decoder is None:
        npz = os.path.join(os.path.dirname(os.path.abspath(glb_path)), "casa_textures.npz")
        if os.path.exists(npz):
            stored = np.load(npz)

            def image_decoder(raw, index):
                return np.ascontiguousarray(stored["image%d" % index])
        else:
            image_decoder = decode_image
    meshes, materials, textures, instances = load_glb(glb_path, image_decoder)
    b = SceneBuilder()
    for t in textures:
        b.add_texture(t)
    for m in meshes:
        b.add_mesh(m)
    # scene(identity) * o1(rotation -90,0,0) * child transforms  (Instance::matrix_to_global order)
    scene_m = transform_matrix()
    root = mat4_mul(scene_m, transform_matrix((0, 0, 0), (-90, 0, 0), (1, 1, 1)))
    for name, mesh, mat, l2w in instances:
        b.add_instance(mesh, mat4_mul(root, l2w), materials[mat])
    return b.finish(n_threads)


def casa_camera():
    """Camera pose of main.cpp:384-386 -> v2w matrix (Camera::v2w_matrix)."""
    cam = transform_matrix((-16.9924, 8.50888, 17.9285), (-23.19, -48.8484, 0), (1, 1, 1))
    return mat4_mul(transform_matrix(), cam)


# --------------------------------------------------------------------------------------
# sky
# --------------------------------------------------------------------------------------
class Sky:
    """Environment map + importance-sampling cache (HDRTexture + Skybox)."""

    def __init__(self, rgb):
        self.rgb = np.ascontiguousarray(rgb, np.float32)
        self.height, self.width = self.rgb.shape[:2]
        self.cache = np.zeros_like(self.rgb)
        ls = C.c_double(0)
        check(lib.pt_host_sky_cache(_f32p(self.rgb), self.width, self.height, _f32p(self.cache), C.byref(ls)))
        self.light_sum = ls.value

    @staticmethod
    def from_hdr_bytes(data):
        w, h = C.c_int32(0), C.c_int32(0)
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        check(lib.pt_host_rgbe_decode(buf, len(data), C.byref(w), C.byref(h), None))
        rgb = np.zeros((h.value, w.value, 3), np.float32)
        check(lib.pt_host_rgbe_decode(buf, len(data), C.byref(w), C.byref(h), _f32p(rgb)))
        return Sky(rgb)


def rgbe_encode(rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w = rgb.shape[:2]
    n = lib.pt_host_rgbe_encode(_f32p(rgb), w, h, None, 0)
    check(int(n))
    out = (C.c_char * n)()
    check(int(lib.pt_host_rgbe_encode(_f32p(rgb), w, h, out, n)))
    return bytes(out)


def procedural_sky_rgb(width=2048, height=1024):
    """Stand-in for the demo's kloofendal_48d_partly_cloudy_puresky_2k.hdr, which is not
    redistributable with this repository: a blue gradient with a warm horizon band and a
    small bright sun, built from +,-,*,/ on float32 only so that every machine computes
    the same bits.  Row 0 is the top of the sky (file scanline order)."""
    f = np.float32
    v = ((np.arange(height, dtype=np.float32) + f(0.5)) / f(height))[:, None]   # 0 top .. 1 bottom
    u = ((np.arange(width, dtype=np.float32) + f(0.5)) / f(width))[None, :]
    up = np.maximum(f(0), f(1) - f(2) * v)            # 1 at zenith, 0 at and below horizon
    down = np.maximum(f(0), f(2) * v - f(1))
    hz = f(1) / (f(1) + f(40) * (v - f(0.5)) * (v - f(0.5)))    # horizon glow
    r = f(0.25) + f(0.55) * hz - f(0.10) * up
    g = f(0.40) + f(0.45) * hz + f(0.05) * up
    b = f(0.65) + f(0.30) * hz + f(0.25) * up
    gr = f(1) - f(0.85) * np.minimum(f(1), f(3) * down)         # dark ground below the horizon
    r = r * gr + u * f(0)
    g = g * gr + u * f(0)
    b = b * gr + u * f(0)
    du = u - f(0.62)
    du = du - np.round(du)
    dv = v - f(0.27)
    d2 = (du * du * f(4) + dv * dv) * f(width) * f(width) / f(4)    # texel-ish distance^2 (u spans 2x v)
    s = f(30) / (f(1) + d2 / f(60))
    sun = s * s * f(25)                                             # peak 22500, radius ~ 8 texels at 2k
    rgb = np.stack([r + sun * f(1.0), g + sun * f(0.9), b + sun * f(0.7)], axis=-1)
    return np.ascontiguousarray(rgb, np.float32)


def procedural_sky(width=2048, height=1024):
    """procedural_sky_rgb pushed through the Radiance RGBE codec, i.e. exactly what the
    pass would see had the picture been loaded from a .hdr file."""
    return Sky.from_hdr_bytes(rgbe_encode(procedural_sky_rgb(width, height)))


def random_rays(lo, hi, n, seed):
    """Incoherent test rays: origins uniform in the box [lo, hi], directions uniform on the
    sphere (z = 2u - 1, phi = 2 pi u'), from numpy's PCG64 seeded with `seed`; (n, 6) float32."""
    rng = np.random.default_rng(seed)
    o = lo + (hi - lo) * rng.random((n, 3))
    z = 2 * rng.random(n) - 1
    ph = 2 * np.pi * rng.random(n)
    r = np.sqrt(np.maximum(0.0, 1 - z * z))
    d = np.stack([r * np.cos(ph), r * np.sin(ph), z], 1)
    return np.concatenate([o, d], 1).astype(np.float32)


# --------------------------------------------------------------------------------------
# synthetic large scene
# --------------------------------------------------------------------------------------
def _grid_mesh(pos, nu, nv, uv_scale=1.0):
    """Triangulates an (nu+1) x (nv+1) vertex grid; smooth vertex normals from the faces."""
    pos = pos.astype(np.float32)
    iu, iv = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (iu * (nv + 1) + iv).ravel()
    b = a + (nv + 1)
    c = b + 1
    d = a + 1
    idx = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], 0)
    P = pos.reshape(-1, 3)
    fn = np.cross(P[idx[:, 1]] - P[idx[:, 0]], P[idx[:, 2]] - P[idx[:, 0]])
    vn = np.zeros(P.shape, np.float64)
    for k in range(3):
        for c in range(3):
            vn[:, c] += np.bincount(idx[:, k], weights=fn[:, c], minlength=len(P))
    vn = vn.astype(np.float32)
    vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-20)
    gu, gv = np.meshgrid(np.arange(nu + 1) / nu, np.arange(nv + 1) / nv, indexing="ij")
    UV = (np.stack([gu, gv], -1).reshape(-1, 2) * uv_scale).astype(np.float32)
    tri = np.concatenate([P[idx].reshape(-1, 9), vn[idx].reshape(-1, 9), UV[idx].reshape(-1, 6)], 1)
    return np.ascontiguousarray(tri, np.float32)


def procedural_texture(size, seed, channels=4):
    """size x size 8-bit texture: bands and cells with seeded colours (stands in for the 4K
    albedo maps of the texture-heavy configuration)."""
    rng = np.random.default_rng(1000 + seed)
    y, x = np.mgrid[0:size, 0:size].astype(np.float32) / np.float32(size)
    base = rng.random(3).astype(np.float32)
    f = rng.integers(2, 9, 3)
    img = np.empty((size, size, channels), np.float32)
    for c in range(3):
        img[..., c] = 0.15 + 0.8 * (0.5 + 0.5 * np.sin(6.2831853 * (f[c] * x + (c + 1) * y))) * base[c] \
                      + 0.1 * (((x * 16).astype(np.int32) + (y * 16).astype(np.int32)) & 1)
    if channels == 4:
        img[..., 3] = 1.0
    return np.ascontiguousarray(np.clip(img * 255.0, 0, 255).astype(np.uint8))


def synthetic_scene(n_tris=10_000_000, n_threads=0, seed=12345, dielectric_fraction=0.0, lights=0, textures=0,
                    texture_size=256):
    """Procedural stand-in for the large configurations: a 4 x 4 tiling of displaced terrain
    patches and 48 displaced spheres resting above it, 64 meshes / 64 instances in all,
    about n_tris triangles, conductor materials with seeded roughness 0.05..0.8 (a
    fraction may be rough dielectrics), optionally `lights` analytic lights (alternating
    point and directional) and `textures` procedural RGBA8 albedo maps spread over the
    instances.  Deterministic for a given argument list."""
    rng = np.random.default_rng(seed)
    per_mesh = max(8, n_tris // 64)
    k = max(2, int(round(math.sqrt(per_mesh / 2.0))))
    b = SceneBuilder()
    tile = 20.0
    tex_ids = [b.add_texture(procedural_texture(texture_size, k)) for k in range(textures)]
    n_mat = [0]

    def material():
        t = 2 if rng.random() < dielectric_fraction else 1
        n_mat[0] += 1
        return dict(type=t, albedo=tuple(np.float32(0.2 + 0.75 * rng.random(3))),
                    roughness=np.float32(0.05 + 0.75 * rng.random()),
                    metallic=np.float32(1.0 if rng.random() < 0.5 else 0.0),
                    tex=tex_ids[n_mat[0] % len(tex_ids)] if tex_ids and n_mat[0] % 4 != 0 else -1,
                    ior=np.float32(1.3 + 0.5 * rng.random()))

    def height(x, z):
        return (1.5 * np.sin(0.21 * x) * np.cos(0.17 * z) + 0.6 * np.sin(0.53 * x + 1.3) * np.sin(0.47 * z + 0.4)
                + 0.15 * np.sin(2.1 * x) * np.cos(1.9 * z))

    for ti in range(4):
        for tj in range(4):
            u = np.linspace(0, tile, k + 1)
            x, z = np.meshgrid(u + (ti - 2) * tile, u + (tj - 2) * tile, indexing="ij")
            y = height(x, z)
            # mesh-local coordinates are relative to the tile corner; the instance translates
            pos = np.stack([x - (ti - 2) * tile, y, z - (tj - 2) * tile], -1)
            m = b.add_mesh(_grid_mesh(pos, k, k, 4.0))
            b.add_instance(m, transform_matrix(((ti - 2) * tile, 0, (tj - 2) * tile)), material())
    ku = max(2, int(round(math.sqrt(per_mesh))))   # ku x (ku/2) quads x 2
    kv = max(2, ku // 2)
    for s in range(48):
        th = np.linspace(0, 2 * np.pi, ku + 1)
        ph = np.linspace(0.02, np.pi - 0.02, kv + 1)
        T, Pp = np.meshgrid(th, ph, indexing="ij")
        f1, f2, a1 = 3 + rng.integers(0, 6), 2 + rng.integers(0, 5), 0.05 + 0.15 * rng.random()
        rad = 1.0 + a1 * np.sin(f1 * T) * np.sin(f2 * Pp)
        pos = np.stack([rad * np.sin(Pp) * np.cos(T), rad * np.cos(Pp), rad * np.sin(Pp) * np.sin(T)], -1)
        m = b.add_mesh(_grid_mesh(pos, ku, kv, 1.0))
        cx, cz = (rng.random(2) * 4 - 2) * tile * 0.9
        scale = 1.5 + 3.0 * rng.random()
        cy = float(height(cx, cz)) + scale * (0.9 + 1.5 * rng.random())
        rot = tuple(rng.random(3) * 360.0)
        b.add_instance(m, transform_matrix((cx, cy, cz), rot, (scale, scale, scale)), material())
    for k in range(lights):
        if k % 2 == 0:
            b.add_point_light(transform_matrix((-10.0 + 9.0 * k, 14.0 + k, 6.0 - 5.0 * k)), (400.0, 350.0 + 20 * k, 300.0))
        else:
            b.add_directional_light(transform_matrix((0, 0, 0), (-50.0 - 7 * k, 30.0 * k, 0.0)), (2.0, 2.0, 1.6 + 0.1 * k))
    return b.finish(n_threads)


def rebuild_scene_level(sd, l2w):
    """ResourceManager::reload_scene (src/ResourceManager.cpp:200-224) after instances moved:
    new instance records and scene BVH for local-to-world matrices `l2w` (n_inst x 16), meshes
    untouched.  Returns a shallow copy of `sd` with instances / scene_nodes / l2w replaced."""
    import copy
    out = copy.copy(sd)
    n_inst = len(sd.instances)
    out.l2w = np.ascontiguousarray(l2w, np.float32).reshape(n_inst, 16)
    out.instances = np.zeros(n_inst, INSTANCE_DTYPE)
    out.scene_nodes = np.zeros(2 * n_inst - 1, NODE_DTYPE)
    stats = np.zeros(2, np.int32)
    check(lib.pt_host_build_scene(_f32p(sd.mesh_root_box), _i32p(sd.mesh_siz), _i32p(sd.mesh_depth), len(sd.mesh_siz),
                                  _f32p(out.l2w), _i32p(sd.mesh_index), _i32p(sd.material_ptr), n_inst,
                                  out.instances.ctypes.data, out.scene_nodes.ctypes.data, _i32p(stats)))
    out.bvh_siz, out.bvh_depth = int(stats[0]), int(stats[1])
    return out


def synthetic_camera():
    """Elevated view across the synthetic scene."""
    return transform_matrix((-34.0, 16.0, 38.0), (-20.0, -42.0, 0.0), (1, 1, 1))


# --------------------------------------------------------------------------------------
# per-draw parameters
# --------------------------------------------------------------------------------------
def make_params(w, h, v2w, sky, spp=1, frame=1, jitter=(0.5, 0.5), max_depth=2, fov=None, far_plane=1000.0,
                light_count=0, brdf_sampling=True, skybox_sampling=True, skybox_lighting=True,
                raster_first=False, backproj=None, accum_mode=0):
    """The uniform block of main.cpp:185-205 (+ lightCount, Renderer.cpp:40)."""
    p = PtParams()
    if fov is None:
        fov = float(np.float32(np.float32(math.acos(-1.0)) / np.float32(3)))   # FOV_X = M_PI / 3 with float M_PI
    p.jitter[0], p.jitter[1] = jitter
    p.currentspp = spp
    p.skybox_Light_SUM = sky.light_sum if sky is not None else 1.0
    for i in range(16):
        p.v2wMat[i] = float(v2w[i])
    p.SCREEN_W, p.SCREEN_H, p.WINDOW_W, p.WINDOW_H = w, h, w, h
    p.MAX_DEPTH = max_depth
    p.cameraNear = 1.0 / math.tan(fov / 2)
    p.cameraFar = far_plane
    p.frameCounter = frame
    p.fov = fov
    p.SKY_W = sky.width if sky is not None else 1
    p.SKY_H = sky.height if sky is not None else 1
    if backproj is None:
        backproj = mat4_mul(camera_projection(fov, far_plane, w, h), mat4_inverse(v2w))
    for i in range(16):
        p.backprojMat[i] = float(backproj[i])
    p.BRDFSampling, p.SkyboxSampling, p.SkyboxLighting = int(brdf_sampling), int(skybox_sampling), int(skybox_lighting)
    p.RasterizaionFor1st = int(raster_first)
    p.lightCount = light_count
    p.accum_mode = accum_mode
    return p


def mt19937_jitters(n_spp, static_blender=False):
    """Per-spp jitter as main.cpp:170-172 draws it from std::mt19937 globalRand(0):
    two draws per spp, x first (argument evaluation order is unspecified in C++; this is
    the documented choice), divided by max() in float; the last spp of a frame is forced
    to the pixel centre unless the static blender is on."""
    import random  # noqa: F401  (numpy's MT19937 is the same generator as std::mt19937)
    bg = np.random.MT19937()
    # std::mt19937(0) seeds with init_genrand(0); numpy's legacy seeding does the same
    bg._legacy_seeding(0)
    raw = bg.random_raw(2 * n_spp).astype(np.uint32)
    mx = np.float32(4294967295.0)
    out = []
    for s in range(n_spp):
        x = np.float32(np.float32(raw[2 * s]) * np.float32(1.0) / mx)
        y = np.float32(np.float32(raw[2 * s + 1]) * np.float32(1.0) / mx)
        if not static_blender and s == n_spp - 1:
            x = y = np.float32(0.5)
        out.append((float(x), float(y)))
    return out


def load_glb_capi(path, root_l2w=None):
    """The same scene description as load_glb, read by the library's own reader (pt_host_glb_*):
    dict with tri_count, triangles (n, 24), materials (list of dicts), texture_bytes (encoded
    images), instances (mesh, material, l2w), material stream + pointers."""
    g = C.c_void_p()
    check(lib.pt_host_glb_open(str(path).encode(), C.byref(g)))
    try:
        info = _capi.PtGlbInfo()
        check(lib.pt_host_glb_info(g, C.byref(info)))
        tri_count = np.zeros(info.n_meshes, np.int32)
        tris = np.zeros((info.n_triangles, 24), np.float32)
        check(lib.pt_host_glb_meshes(g, _i32p(tri_count), tris.ctypes.data))
        mats = (_capi.PtGlbMaterial * max(info.n_materials, 1))()
        if info.n_materials:
            check(lib.pt_host_glb_materials(g, mats))
        materials = [dict(type=1, albedo=tuple(np.float32(x) for x in mats[k].albedo), roughness=np.float32(mats[k].roughness),
                          metallic=np.float32(mats[k].metallic), tex=int(mats[k].texture)) for k in range(info.n_materials)]
        tex_bytes, tex_image = [], []
        for k in range(info.n_textures):
            ptr, n, img = C.c_void_p(), C.c_size_t(), C.c_int32()
            check(lib.pt_host_glb_texture(g, k, C.byref(ptr), C.byref(n), C.byref(img)))
            tex_bytes.append(C.string_at(ptr, n.value))
            tex_image.append(img.value)
        mesh_index = np.zeros(info.n_instances, np.int32)
        material_index = np.zeros(info.n_instances, np.int32)
        l2w = np.zeros((info.n_instances, 16), np.float32)
        root = None if root_l2w is None else _f32p(np.ascontiguousarray(root_l2w, np.float32))
        check(lib.pt_host_glb_instances(g, root, _i32p(mesh_index), _i32p(material_index), _f32p(l2w)))
        n = int(lib.pt_host_glb_material_stream(g, None, 0, None))
        check(n)
        stream = np.zeros(n, np.float32)
        material_ptr = np.zeros(info.n_instances, np.int32)
        check(int(lib.pt_host_glb_material_stream(g, _f32p(stream), n, _i32p(material_ptr))))
        return dict(tri_count=tri_count, triangles=tris, materials=materials, texture_bytes=tex_bytes, texture_image=tex_image,
                    mesh_index=mesh_index, material_index=material_index, l2w=l2w, material_stream=stream,
                    material_ptr=material_ptr)
    finally:
        lib.pt_host_glb_close(g)
