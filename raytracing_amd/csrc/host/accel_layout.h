// Re-layout of the reference-format scene buffers (48-byte nodes, 96-byte triangles,
// 112-byte instance records) into the records the traversal kernels read.  Pure host
// code; also validates every index so that a malformed buffer is refused here instead
// of faulting on the GPU.
#pragma once
#include <string>
#include <vector>
#include "../device/device_types.h"
#include "scene_types.h"

namespace rta {

struct AccelHost {
    std::vector<TNode> mesh_nodes;
    std::vector<TNode> scene_nodes;
    std::vector<TriPos> tri_pos;
    std::vector<TLeaf> tri_leaf;   // same index as tri_pos
    std::vector<InstAccel> instances;
    SceneHeader header;
    int max_stack = 0;  // upper bound of deferred entries one ray can hold (scene + mesh height)
};

// Returns false and sets err when the buffers are inconsistent.
bool build_accel(const BVHNodeInfo *mesh, size_t n_mesh_nodes, const BVHNodeInfo *scene, size_t n_scene_nodes,
                 const TriangleRec *tris, size_t n_tris, const InstanceInfo *inst, size_t n_inst, AccelHost &out,
                 std::string &err);

}  // namespace rta
