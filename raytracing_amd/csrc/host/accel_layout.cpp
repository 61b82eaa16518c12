#include "accel_layout.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace rta {
namespace {

void set_box(float lo[3], float hi[3], const BVHNodeInfo &n) {
    for(int k = 0; k < 3; k++) {
        lo[k] = n.aa[k];
        hi[k] = n.bb[k];
    }
}

// Children are numbered after their parent by both flatteners (breadth-first order), so
// "child index > own index" proves the arrays form a forest and every walk terminates.
bool check_children(const BVHNodeInfo *nodes, size_t n, size_t i, const char *what, std::string &err) {
    const BVHNodeInfo &p = nodes[i];
    if(p.lsIndex < 0 || p.rsIndex < 0 || (size_t)p.lsIndex >= n || (size_t)p.rsIndex >= n ||
       (size_t)p.lsIndex <= i || (size_t)p.rsIndex <= i) {
        err = std::string(what) + " BVH node " + std::to_string(i) + " has invalid child indices";
        return false;
    }
    return true;
}

}  // namespace

bool build_accel_mesh_level(const BVHNodeInfo *mesh, size_t n_mesh_nodes, const TriangleRec *tris, size_t n_tris,
                            AccelMeshLevel &out, std::string &err) {
    if(n_mesh_nodes == 0 || n_tris == 0) {
        err = "scene buffers must not be empty (mesh BVH, triangles)";
        return false;
    }
    if(n_mesh_nodes >= REF_INDEX_MASK || n_tris >= REF_INDEX_MASK) {
        err = "scene too large for 30-bit references";
        return false;
    }
    out.n_tris = n_tris;
    std::vector<uint8_t> has_parent(n_mesh_nodes, 0);
    for(size_t i = 0; i < n_mesh_nodes; i++) {
        if(mesh[i].triangleIndex >= 0) {
            if((size_t)mesh[i].triangleIndex >= n_tris) {
                err = "mesh BVH node " + std::to_string(i) + " references a triangle out of range";
                return false;
            }
        } else {
            if(!check_children(mesh, n_mesh_nodes, i, "mesh", err)) return false;
            has_parent[mesh[i].lsIndex] = 1;
            has_parent[mesh[i].rsIndex] = 1;
        }
    }
    // Slots: every reference node gets one 64-byte slot, inner nodes a TNode, leaves a TLeaf, in
    // depth-first order of each tree (node, subtree of ls, subtree of rs): a walk down a tree
    // moves through a shrinking address range, and a node often shares its 128-byte line with the
    // child visited next.  (The reference numbers nodes breadth-first across all meshes.)
    std::vector<uint32_t> &slot_of = out.slot_of;
    slot_of.assign(n_mesh_nodes, 0xFFFFFFFFu);
    uint32_t next = 0;
    {
        std::vector<int32_t> stack;
        for(size_t r = 0; r < n_mesh_nodes; r++) {
            if(has_parent[r]) continue;
            stack.push_back((int32_t)r);
            while(!stack.empty()) {
                const int32_t n = stack.back();
                stack.pop_back();
                if(slot_of[n] != 0xFFFFFFFFu) continue;  // reached twice: the arrays form a DAG, keep the first slot
                slot_of[n] = next++;
                if(mesh[n].triangleIndex < 0) {
                    stack.push_back(mesh[n].rsIndex);
                    stack.push_back(mesh[n].lsIndex);
                }
            }
        }
    }
    auto mesh_ref = [&](int32_t c) { return make_ref(mesh[c].triangleIndex >= 0 ? REF_TRIANGLE : REF_MESH_INNER, slot_of[c]); };
    out.slots.assign(next, TNode());
    memset(out.slots.data(), 0, out.slots.size() * sizeof(TNode));
    out.height.assign(n_mesh_nodes, 1);
    const float inf = std::numeric_limits<float>::infinity();
    auto fill_leaf = [&](int32_t node, const float *plo, const float *phi) {
        TLeaf &l = *reinterpret_cast<TLeaf *>(&out.slots[slot_of[node]]);
        const float *v = tris[mesh[node].triangleIndex].ver;
        for(int k = 0; k < 3; k++) {
            l.parent_lo[k] = plo ? plo[k] : -inf;
            l.parent_hi[k] = phi ? phi[k] : inf;
            l.v0[k] = v[k];
            l.e1[k] = v[3 + k] - v[k];
            l.e2[k] = v[6 + k] - v[k];
        }
        l.triangle = (uint32_t)mesh[node].triangleIndex;
    };
    for(size_t ii = n_mesh_nodes; ii-- > 0;) {
        const BVHNodeInfo &p = mesh[ii];
        if(p.triangleIndex >= 0) {
            if(!has_parent[ii]) fill_leaf((int32_t)ii, nullptr, nullptr);  // a one-triangle mesh
            continue;
        }
        TNode &t = out.slots[slot_of[ii]];
        set_box(t.lo0, t.hi0, mesh[p.lsIndex]);
        set_box(t.lo1, t.hi1, mesh[p.rsIndex]);
        t.c0 = mesh_ref(p.lsIndex);
        t.c1 = mesh_ref(p.rsIndex);
        // a leaf carries the exact box of its parent in the reference tree
        for(int32_t c : {p.lsIndex, p.rsIndex})
            if(mesh[c].triangleIndex >= 0) fill_leaf(c, p.aa, p.bb);
        const int h = std::max<int>(out.height[p.lsIndex], out.height[p.rsIndex]) + 1;
        out.height[ii] = (uint8_t)std::min(h, 255);
    }
    return true;
}

bool build_accel_scene_level(const BVHNodeInfo *mesh, size_t n_mesh_nodes, const AccelMeshLevel &ml, const BVHNodeInfo *scene,
                             size_t n_scene_nodes, const InstanceInfo *inst, size_t n_inst, AccelSceneLevel &out,
                             std::string &err) {
    if(n_scene_nodes == 0 || n_inst == 0) {
        err = "scene buffers must not be empty (scene BVH, instances)";
        return false;
    }
    if(n_inst >= REF_INDEX_MASK || n_scene_nodes >= REF_INDEX_MASK) {
        err = "scene too large for 30-bit references";
        return false;
    }
    if(ml.slot_of.size() != n_mesh_nodes) {
        err = "mesh level was built for another mesh BVH";
        return false;
    }
    auto mesh_ref = [&](int32_t c) { return make_ref(mesh[c].triangleIndex >= 0 ? REF_TRIANGLE : REF_MESH_INNER, ml.slot_of[c]); };
    // ---- instances
    out.instances.resize(n_inst);
    int max_mesh_height = 1;
    for(size_t i = 0; i < n_inst; i++) {
        const InstanceInfo &s = inst[i];
        if(s.meshIndex < 0 || (size_t)s.meshIndex >= n_mesh_nodes) {
            err = "instance " + std::to_string(i) + " has a mesh root index out of range";
            return false;
        }
        if(s.materialPtr < 0) {
            err = "instance " + std::to_string(i) + " has a negative material pointer";
            return false;
        }
        InstAccel &a = out.instances[i];
        for(int r = 0; r < 3; r++)
            for(int c = 0; c < 4; c++) a.row[r][c] = s.world2local[c * 4 + r];
        const BVHNodeInfo &root = mesh[s.meshIndex];
        set_box(a.root_lo, a.root_hi, root);
        a.root_ref = mesh_ref(s.meshIndex);
        a.material_ptr = s.materialPtr;
        for(int k = 0; k < 3; k++) {
            a.world_lo[k] = 0.0f;
            a.world_hi[k] = 0.0f;
        }
        memset(a.pad0, 0, sizeof a.pad0);
        memset(a.pad1, 0, sizeof a.pad1);
        max_mesh_height = std::max<int>(max_mesh_height, ml.height[s.meshIndex]);
    }
    // ---- scene level
    std::vector<uint32_t> sinner_of(n_scene_nodes, 0);
    uint32_t n_sinner = 0;
    for(size_t i = 0; i < n_scene_nodes; i++) {
        if(scene[i].instanceIndex >= 0) {
            if((size_t)scene[i].instanceIndex >= n_inst) {
                err = "scene BVH node " + std::to_string(i) + " references an instance out of range";
                return false;
            }
        } else {
            if(!check_children(scene, n_scene_nodes, i, "scene", err)) return false;
            sinner_of[i] = n_sinner++;
        }
    }
    const uint32_t inst_base = std::max<uint32_t>(n_sinner, 1u);
    if((uint64_t)inst_base + 2 * (uint64_t)n_inst >= REF_INDEX_MASK) {
        err = "scene too large for 30-bit references";
        return false;
    }
    auto scene_ref = [&](int32_t c) {
        return scene[c].instanceIndex >= 0 ? make_ref(REF_INSTANCE, inst_base + 2u * (uint32_t)scene[c].instanceIndex)
                                           : make_ref(REF_SCENE_INNER, sinner_of[c]);
    };
    out.scene_nodes.assign(inst_base, TNode());
    memset(out.scene_nodes.data(), 0, out.scene_nodes.size() * sizeof(TNode));
    std::vector<int> sheight(n_scene_nodes, 1);
    for(size_t ii = n_scene_nodes; ii-- > 0;) {
        const BVHNodeInfo &p = scene[ii];
        if(p.instanceIndex >= 0) continue;
        TNode &t = out.scene_nodes[sinner_of[ii]];
        set_box(t.lo0, t.hi0, scene[p.lsIndex]);
        set_box(t.lo1, t.hi1, scene[p.rsIndex]);
        t.c0 = scene_ref(p.lsIndex);
        t.c1 = scene_ref(p.rsIndex);
        sheight[ii] = std::max(sheight[p.lsIndex], sheight[p.rsIndex]) + 1;
    }
    for(size_t i = 0; i < n_scene_nodes; i++)
        if(scene[i].instanceIndex >= 0) {
            InstAccel &a = out.instances[scene[i].instanceIndex];
            set_box(a.world_lo, a.world_hi, scene[i]);
        }
    set_box(out.header.root_lo, out.header.root_hi, scene[0]);
    out.header.extent = 0.0f;
    for(int k = 0; k < 3; k++) out.header.extent += std::fabs(scene[0].aa[k]) > std::fabs(scene[0].bb[k]) ? std::fabs(scene[0].aa[k]) : std::fabs(scene[0].bb[k]);
    out.header.root_ref = scene_ref(0);
    out.header.n_instances = (uint32_t)n_inst;
    out.header.inst_base = inst_base;
    // each level of a walk defers at most one sibling
    out.max_stack = sheight[0] + max_mesh_height;
    return true;
}

}  // namespace rta
