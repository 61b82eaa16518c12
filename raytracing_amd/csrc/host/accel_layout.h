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

// Mesh level: depends on the triangle and mesh-BVH buffers only.
struct AccelMeshLevel {
    std::vector<TNode> slots;           // the mesh buffer: one 64-byte slot per reference node, TNode or TLeaf
    // kept on the host for the scene level (instances name their mesh root by node index)
    std::vector<uint32_t> slot_of;      // reference node index -> slot
    std::vector<uint8_t> height;        // height of the subtree under each reference node, capped at 255
    std::vector<uint32_t> wide_of;      // reference node index -> slot of its WNode, REF_NONE if it has none
    std::vector<uint8_t> wide_height;   // per wide node (by (slot - wide_base) / wide_stride): height of the wide subtree
    uint32_t wide_base = 0;             // first wide-node slot
    uint32_t wide_stride = 2;           // slots per wide node: 2 (WNode) or 1 (QNode)
    bool quantised = false;
    // every child box lies inside its parent's (what the reference's builder produces, and what
    // lets the ordered traversal skip or loosen the tests of ancestors)
    bool nested = true;
    size_t n_tris = 0;
    // copies of the nodes that have a wide tree (mesh roots and whatever else instances start at), by
    // ascending node index: all the scene level reads of the mesh BVH, so it can be rebuilt without that array
    std::vector<int32_t> root_ids;
    std::vector<BVHNodeInfo> root_nodes;
};

// Scene level: depends on the instance and scene-BVH buffers (and the mesh roots' boxes).
// Small; rebuilt whenever instances move (the reference does that every frame, DynamicBVH).
struct AccelSceneLevel {
    std::vector<TNode> scene_nodes;     // slots [0, inst_base) of the scene buffer
    std::vector<InstAccel> instances;   // two slots each from inst_base on
    std::vector<TNode> wide_nodes;      // WNodes (two slots each) or QNodes after the instances, as raw slots
    SceneHeader header;
    int max_stack = 0;  // upper bound of deferred entries one ray can hold (scene + mesh height)
    bool nested = true; // as in the mesh level, for the scene tree
};

// Both return false and set err when the buffers are inconsistent.
// Every node without a parent gets a wide tree; extra_roots names further nodes that instances
// use as mesh roots.  The scene level sets missing_root (and fails) when an instance names an
// inner node that has none: the caller rebuilds the mesh level with those roots.  The scene level may be
// given mesh == nullptr: it then works from the root copies the mesh level keeps, and reports an instance
// that starts anywhere else as a missing root.
// quantise: 0 = WNodes, 1 or -1 (the default choice) = QNodes, which measured faster on small and large scenes alike.
bool build_accel_mesh_level(const BVHNodeInfo *mesh, size_t n_mesh_nodes, const TriangleRec *tris, size_t n_tris,
                            const std::vector<int32_t> &extra_roots, int quantise, AccelMeshLevel &out, std::string &err);
bool build_accel_scene_level(const BVHNodeInfo *mesh, size_t n_mesh_nodes, const AccelMeshLevel &ml, const BVHNodeInfo *scene,
                             size_t n_scene_nodes, const InstanceInfo *inst, size_t n_inst, AccelSceneLevel &out,
                             std::string &err, bool *missing_root = nullptr);

// The whole scene as ONE four-wide tree in world space (trace_flat.h): a FlatLeaf per (instance, triangle), two
// slots each, then QNodes.  Built over the world-space boxes of the leaves' reference parent boxes.
struct AccelFlat {
    std::vector<TNode> slots;
    uint32_t root_ref = REF_NONE;
    float pad = 0.0f;    // FastScene::flat_pad
    float reach = 0.0f;  // FastScene::flat_reach
    int max_stack = 0;
    size_t n_leaves = 0;
};
// Returns false when the scene gets no flat tree -- more than max_leaves (instance, triangle) pairs, a singular or
// badly conditioned instance matrix, coordinates that are not finite, or (only_axis_aligned) an instance rotated
// off the axes, whose parent boxes turn into loose world boxes (measured: 20 % slower than the two levels on a
// scene of randomly rotated meshes, 5 % faster on the axis-aligned demo scene) -- and says why; that is no error: the two
// levels serve such a scene.  The buffers must have passed build_accel_mesh_level / build_accel_scene_level.
bool build_accel_flat(const BVHNodeInfo *mesh, size_t n_mesh_nodes, const TriangleRec *tris, size_t n_tris,
                      const BVHNodeInfo *scene, size_t n_scene_nodes, const InstanceInfo *inst, size_t n_inst,
                      size_t max_leaves, bool only_axis_aligned, int n_threads, AccelFlat &out, std::string &why_not);
// Independent check of a flat tree that still holds its slots: it reaches every (instance, triangle) pair of the
// scene exactly once; a leaf carries its instance's matrix, its triangle and the box of the triangle's reference
// parent; every child box as the kernel decodes it contains the world boxes (parent box corners through the
// inverse of the leaf's own matrix, long double) of the leaves below it; max_stack bounds the deepest walk.
// counts: {leaves, wide nodes, child boxes checked, planes that decode to "unbounded"}.
bool verify_accel_flat(const BVHNodeInfo *mesh, size_t n_mesh_nodes, const TriangleRec *tris, const InstanceInfo *inst, size_t n_inst,
                       const AccelFlat &fl, uint64_t counts[4], std::string &err);

// Independent check of the two levels (host only, meant for tests and for debugging a scene
// the kernels disagree on): every wide tree reaches exactly the triangles / instances its
// binary tree reaches, every child box as the kernel decodes it contains the reference's box
// (for a triangle: its parent's), references stay inside the buffers, and max_stack bounds
// the deferred entries of the deepest walk.  counts: {wide mesh nodes, wide scene nodes,
// child boxes checked, planes that decode to "unbounded"}.
bool verify_accel_levels(const BVHNodeInfo *mesh, size_t n_mesh_nodes, const AccelMeshLevel &ml, const BVHNodeInfo *scene,
                         size_t n_scene_nodes, size_t n_inst, const AccelSceneLevel &sl, uint64_t counts[4], std::string &err);

}  // namespace rta
