// Full-sweep SAH binary BVH whose topology, boxes and child order are bit-identical
// to the reference builder's (src/BVH.cpp:11-82), plus the two breadth-first
// flatteners that produce the reference's GPU node arrays
// (src/ResourceManager.cpp:116-140 for meshes, :167-195 for the scene level).
#pragma once
#include <vector>
#include "scene_types.h"

namespace rta {

struct Box {
    float mi[3];
    float mx[3];
    // src/AABB.cpp:16 -- empty box is (+1e18, -1e18)
    Box() { mi[0] = mi[1] = mi[2] = 1e18f; mx[0] = mx[1] = mx[2] = -1e18f; }
    void add_point(const float p[3]) {
        for(int k = 0; k < 3; k++) {
            mi[k] = p[k] < mi[k] ? p[k] : mi[k];
            mx[k] = mx[k] < p[k] ? p[k] : mx[k];
        }
    }
    void combine(const Box &b) {
        for(int k = 0; k < 3; k++) {
            mi[k] = b.mi[k] < mi[k] ? b.mi[k] : mi[k];
            mx[k] = mx[k] < b.mx[k] ? b.mx[k] : mx[k];
        }
    }
    // src/AABB.cpp:32-35 (half area; association (xy + yz) + zx)
    float surface() const {
        float tx = mx[0] - mi[0], ty = mx[1] - mi[1], tz = mx[2] - mi[2];
        return tx * ty + ty * tz + tz * tx;
    }
};

struct SahNode {
    Box box;
    int left = -1, right = -1;  // indices into SahTree::nodes, -1 on leaves
    int prim = -1;              // primitive index on leaves
    int siz = 1;                // subtree node count  (BVH.h:36)
    int depth = 1;              // subtree height      (BVH.h:37)
};

struct SahTree {
    std::vector<SahNode> nodes;
    int root = -1;
};

// Builds the tree over n primitive boxes (n >= 1).  One primitive per leaf.  n_threads > 1 lets
// a large tree use that many threads (axis sweeps of big nodes and sibling subtrees side by side);
// the tree is the same whatever the number.
void build_sah_tree(const Box *boxes, int n, SahTree &out, int n_threads = 1);

// Mesh level.  trees[m] was built over the boxes of mesh m's triangles in array
// order; tri_base[m] is the index of its first triangle in the concatenated
// triangle buffer.  Node k < trees.size() is the root of mesh k.
void flatten_mesh_trees(const std::vector<SahTree> &trees, const std::vector<int> &tri_base,
                        std::vector<BVHNodeInfo> &out);

// Scene level.  tree was built over one world-space box per global instance;
// leaves carry instanceIndex = primitive index and no children.  mesh_siz/mesh_depth
// give each instance's mesh-tree size and height so that the returned totals match
// what the reference prints for the two-level tree (BVH.cpp:23-24): siz / depth of root.
void flatten_scene_tree(const SahTree &tree, std::vector<BVHNodeInfo> &out);

}  // namespace rta
