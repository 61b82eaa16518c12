"""The library's own .glb reader (pt_host_glb_*, the C-ABI form of what src/tool/loader.cpp:44-158
produces through assimp) against raytracing_amd.scene's Python reader and the arrays casa_scene
builds from it: byte-equal triangles, instance matrices, material stream; malformed files refused."""
import struct

import numpy as np
import pytest

from conftest import CASA


def test_casa_arrays_equal_the_python_reader(casa):
    from raytracing_amd import scene
    root = scene.mat4_mul(scene.transform_matrix(), scene.transform_matrix((0, 0, 0), (-90, 0, 0), (1, 1, 1)))
    g = scene.load_glb_capi(CASA, root)
    meshes, materials, textures, instances = scene.load_glb(CASA, image_decoder=lambda raw, index: raw)
    assert list(g["tri_count"]) == [len(m) for m in meshes] and int(g["tri_count"].sum()) == 34689
    assert g["triangles"].tobytes() == np.concatenate(meshes).tobytes() == casa.triangles.tobytes()
    assert len(g["materials"]) == len(materials) == 13
    for a, b in zip(g["materials"], materials):
        assert a["albedo"] == b["albedo"] and a["roughness"] == b["roughness"] and a["metallic"] == b["metallic"] == 1.0
        assert a["tex"] == b["tex"]
    assert min(m["roughness"] for m in g["materials"]) == np.float32(0.001)      # roughnessFactor 0 -> max(0.001, .)
    assert g["texture_bytes"] == textures and g["texture_bytes"][0][:2] == b"\xff\xd8"   # the embedded JPEG, still encoded
    assert [(int(m), int(k)) for m, k in zip(g["mesh_index"], g["material_index"])] == [(mi, ma) for _, mi, ma, _ in instances]
    assert g["l2w"].tobytes() == casa.l2w.tobytes()
    assert g["material_stream"].tobytes() == np.ascontiguousarray(casa.materials, np.float32).tobytes()
    assert list(g["material_ptr"]) == list(casa.material_ptr)
    # and the whole scene built from the reader's arrays is the scene the tests use
    b = scene.SceneBuilder()
    off = 0
    for n in g["tri_count"]:
        b.add_mesh(g["triangles"][off:off + n])
        off += n
    for k in range(len(g["mesh_index"])):
        b.add_instance(int(g["mesh_index"][k]), g["l2w"][k], g["materials"][int(g["material_index"][k])])
    sd = b.finish()
    for name in ("triangles", "instances", "mesh_nodes", "scene_nodes", "materials"):
        assert getattr(sd, name).tobytes() == getattr(casa, name).tobytes(), name


def _glb(js, blob=b""):
    import json
    j = json.dumps(js).encode()
    j += b" " * (-len(j) % 4)
    blob += b"\0" * (-len(blob) % 4)
    body = struct.pack("<II", len(j), 0x4E4F534A) + j + (struct.pack("<II", len(blob), 0x004E4942) + blob if blob else b"")
    return struct.pack("<III", 0x46546C67, 2, 12 + len(body)) + body


def _open(data):
    import ctypes as C
    from raytracing_amd import _capi
    g = C.c_void_p()
    rc = _capi.lib.pt_host_glb_open_memory(data, len(data), C.byref(g))
    msg = _capi.lib.pt_last_error().decode()
    if rc == 0:
        _capi.lib.pt_host_glb_close(g)
    return rc, msg


def _tiny(**over):
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], np.float32)
    nor = np.tile(np.array([0, 0, 1], np.float32), (4, 1))
    idx = np.array([0, 1, 2, 2, 1, 3], np.uint16)
    blob = pos.tobytes() + nor.tobytes() + idx.tobytes()
    js = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
          "nodes": [{"children": [1], "matrix": [2, 0, 0, 0, 0, 2, 0, 0, 0, 0, 2, 0, 5, 5, 5, 1]},
                    {"mesh": 0, "translation": [1, 2, 3], "scale": [2, 2, 2]}],
          "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1}, "indices": 2, "material": 0}]}],
          "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.5, 0.25, 1, 1], "roughnessFactor": 0.0}}],
          "buffers": [{"byteLength": len(blob)}],
          "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 48}, {"buffer": 0, "byteOffset": 48, "byteLength": 48},
                          {"buffer": 0, "byteOffset": 96, "byteLength": 12}],
          "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"},
                        {"bufferView": 1, "componentType": 5126, "count": 4, "type": "VEC3"},
                        {"bufferView": 2, "componentType": 5123, "count": 6, "type": "SCALAR"}]}
    js.update(over)
    return js, blob


def test_small_file_and_its_mapping(tmp_path):
    from raytracing_amd import scene
    js, blob = _tiny()
    path = tmp_path / "tiny.glb"
    path.write_bytes(_glb(js, blob))
    g = scene.load_glb_capi(path)
    assert list(g["tri_count"]) == [2]
    assert np.array_equal(g["triangles"][1, :9], [0, 1, 0, 1, 0, 0, 1, 1, 0]) and np.all(g["triangles"][:, 18:] == 0)   # no TEXCOORD_0
    # the root's own matrix is dropped (loader.cpp:152-156), the child's TRS kept
    assert np.array_equal(g["l2w"][0], scene.transform_matrix((1, 2, 3), (0, 0, 0), (2, 2, 2)))
    assert list(g["material_stream"]) == [1.0, 0.5, 0.25, 1.0, np.float32(0.001), 1.0, -1.0]


def test_malformed_files_are_refused():
    js, blob = _tiny()
    assert _open(_glb(js, blob))[0] == 0
    assert _open(b"glTF")[0] < 0
    assert _open(_glb(js, blob)[:-20])[0] < 0                                      # truncated binary chunk
    bad, _ = _tiny()
    bad["accessors"][2]["count"] = 600                                             # indices beyond the chunk
    assert "beyond" in _open(_glb(bad, blob))[1]
    bad, _ = _tiny()
    bad["accessors"][0]["count"] = 3                                               # index 3 names no vertex
    assert _open(_glb(bad, blob))[0] < 0
    bad, _ = _tiny()
    bad["nodes"][1]["rotation"] = [0, 0, 0, 1]
    assert "quaternion" in _open(_glb(bad, blob))[1]
    bad, _ = _tiny()
    bad["nodes"][1]["children"] = [0]                                              # a cycle
    assert _open(_glb(bad, blob))[0] < 0
    bad, _ = _tiny()
    bad["meshes"][0]["primitives"][0]["attributes"].pop("NORMAL")
    assert "NORMAL" in _open(_glb(bad, blob))[1]
    assert _open(_glb({"asset": {}}, b""))[0] < 0
    assert _open(struct.pack("<III", 0x46546C67, 2, 40) + struct.pack("<II", 20, 0x4E4F534A) + b'{"a": [1, 2, {"b"')[0] < 0


def test_sizes_that_would_wrap_and_shared_children_are_refused():
    """Offsets, strides and counts are checked as the numbers the JSON holds before any product is formed
    (a stride of 2^62 with count 5 makes 2^64 = 0 in size_t arithmetic), accessors are held inside their
    bufferView, and a node may be reached once: a chain of nodes that each list the next one twice is a
    few hundred bytes that would otherwise be walked 2^depth times."""
    js, blob = _tiny()
    for key, val in (("byteStride", 2.0 ** 62), ("byteStride", 1e300), ("byteStride", 12.5), ("byteOffset", -4),
                     ("byteOffset", 2.0 ** 64), ("byteLength", 2.0 ** 63), ("byteLength", 1e308)):
        bad, _ = _tiny()
        bad["bufferViews"][0][key] = val
        rc, msg = _open(_glb(bad, blob))
        assert rc < 0 and "glb:" in msg, (key, val, msg)
    bad, _ = _tiny()
    bad["accessors"][0]["byteOffset"] = 2.0 ** 63
    assert _open(_glb(bad, blob))[0] < 0
    bad, _ = _tiny()
    bad["accessors"][0]["count"] = 3.5
    assert _open(_glb(bad, blob))[0] < 0
    bad, _ = _tiny()
    bad["bufferViews"][0]["byteLength"] = 36              # the accessor's 4 x 12 bytes no longer fit ITS view (they do fit the chunk)
    assert "bufferView" in _open(_glb(bad, blob))[1]
    bad, _ = _tiny()
    bad["bufferViews"][2]["byteStride"] = 4               # 6 indices 4 bytes apart: 22 bytes from a 12-byte view
    assert "bufferView" in _open(_glb(bad, blob))[1]
    # a legal stride still reads what it should: positions 24 bytes apart = vertices 0 and 2 of the packed array
    ok, _ = _tiny()
    ok["bufferViews"][0]["byteStride"] = 24
    ok["accessors"][0]["count"] = 2
    ok["accessors"][1]["count"] = 2
    ok["accessors"][2] = {"bufferView": 2, "componentType": 5123, "count": 3, "type": "SCALAR"}
    blob2 = blob[:96] + np.array([0, 1, 1, 0, 0, 0], np.uint16).tobytes()
    assert _open(_glb(ok, blob2))[0] == 0
    # 40 levels, every node lists its child twice: 2^40 visits unless the second is refused
    dag, _ = _tiny()
    dag["nodes"] = [{"children": [k + 1, k + 1]} for k in range(40)] + [{"mesh": 0}]
    import time
    t0 = time.time()
    rc, msg = _open(_glb(dag, blob))
    assert rc < 0 and "twice" in msg and time.time() - t0 < 1.0
