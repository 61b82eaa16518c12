"""Experiment, GPU box: what would opening the top of the instances' mesh trees into the scene level win?
Emulated with what exists: every casa mesh is cut into the sub-meshes under a frontier of its own BVH (always
opening the entry of largest world surface area, until `target` entries exist), each becoming an instance with
its parent's transform.  The reference BVHs of such a scene differ from casa's, so results are not comparable bit
for bit -- only the rates are of interest."""
import heapq
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from raytracing_amd import device, scene  # noqa: E402
from flatten_experiment import run  # noqa: E402


def world_area(node, M):
    lo, hi = np.array(node["aa"], np.float64), np.array(node["bb"], np.float64)
    c = np.array([[lo[0] if i & 1 else hi[0], lo[1] if i & 2 else hi[1], lo[2] if i & 4 else hi[2]] for i in range(8)])
    w = c @ M[:3, :3].T + M[:3, 3]
    e = w.max(0) - w.min(0)
    return e[0] * e[1] + e[1] * e[2] + e[2] * e[0]


def split(sd, target):
    nodes = sd.mesh_nodes
    names = nodes.dtype.names
    heap = []
    for i in range(len(sd.instances)):
        M = sd.l2w[i].reshape(4, 4).T.astype(np.float64)
        n = int(sd.mesh_index[i])
        heapq.heappush(heap, (-world_area(nodes[n], M), i, n))
    done = []
    while heap and len(heap) + len(done) < target:
        a, i, n = heapq.heappop(heap)
        if nodes[n]["tri"] >= 0:
            done.append((i, n))
            continue
        M = sd.l2w[i].reshape(4, 4).T.astype(np.float64)
        for c in (int(nodes[n]["ls"]), int(nodes[n]["rs"])):
            heapq.heappush(heap, (-world_area(nodes[c], M), i, c))
    entries = done + [(i, n) for _, i, n in heap]
    b = scene.SceneBuilder()
    for tex in sd.textures:
        b.add_texture(tex)
    for i, n in entries:
        tris = []
        stack = [n]
        while stack:
            k = stack.pop()
            if nodes[k]["tri"] >= 0:
                tris.append(int(nodes[k]["tri"]))
            else:
                stack.append(int(nodes[k]["ls"]))
                stack.append(int(nodes[k]["rs"]))
        m = b.add_mesh(sd.triangles[np.array(sorted(tris))])
        b.add_instance(m, sd.l2w[i].reshape(16), dict(type=1, albedo=(0.8, 0.8, 0.8), roughness=np.float32(0.3), metallic=np.float32(1.0)))
    return b.finish(0), len(entries)


def main():
    sd = scene.casa_scene(os.path.join(ROOT, "assets", "casa_obj.glb"))
    sky = scene.procedural_sky(2048, 1024)
    ctx = device.Context(0)
    run(ctx, sd, sky, "casa")
    for target in (40, 80, 160, 320):
        s2, n = split(sd, target)
        run(ctx, s2, sky, "%d entries" % n)


if __name__ == "__main__":
    main()
