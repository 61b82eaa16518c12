// Host-side producers of the scene buffers (everything the reference's
// ResourceManager / Transform / Camera / Material classes compute on the CPU before
// the path-tracing pass runs).  No GPU calls in here.
#pragma once
#include <vector>
#include "bvh_builder.h"
#include "scene_types.h"
#include "vecmath.h"

namespace rta {

// src/Transform.cpp:20-53 -- T * (Ry*Rx*Rz) * S with Euler angles in degrees, rotation
// order string "YXZ" (Transform.h:15); the translation column is overwritten last.
Mat4 transform_matrix(const float position[3], const float rotation_deg[3], const float scale[3]);

// src/instance/Camera.cpp:21-43 -- the reference's own projection (x,y to [-1,1] with
// near = 1/tan(fovX/2), y scaled by W/H), used only for motion vectors.
Mat4 camera_projection(float fov_x, float far_plane, int screen_w, int screen_h);

// One mesh = one contiguous run of TriangleRec.  Builds every mesh tree (in parallel
// over meshes, at most n_threads at a time) and flattens them.
// src/instance/Mesh.cpp:10-22 + src/ResourceManager.cpp:95-140.
struct MeshSet {
    std::vector<int> tri_base;       // first triangle of mesh m in the triangle buffer
    std::vector<int> tri_count;
    std::vector<Box> root_box;       // box of each mesh root (local space)
    std::vector<int> siz, depth;     // per mesh tree
    std::vector<BVHNodeInfo> nodes;  // flattened mesh-BVH buffer (slot 4)
};
void build_mesh_set(const TriangleRec *tris, const int *tri_count, int n_mesh, int n_threads, MeshSet &out);

// src/ResourceManager.cpp:200-224 + :146-165 + :167-198 given the already walked
// instance list (local-to-world matrix, mesh index and material pointer per global
// instance, in hierarchy pre-order).
struct SceneLevel {
    std::vector<InstanceInfo> instances;  // slot 3
    std::vector<BVHNodeInfo> nodes;       // slot 5
    int siz = 0, depth = 0;               // of the two-level tree, as the reference prints them
};
void build_scene_level(const MeshSet &meshes, const Mat4 *l2w, const int *mesh_index, const int *material_ptr,
                       int n_inst, SceneLevel &out);

// src/material/RoughConductor.cpp:26-38 / RoughDielectric.cpp:23-36: appends one record
// and returns its float offset (= materialPtr).  tex_index < 0 means "no albedo map".
int material_append_conductor(std::vector<float> &stream, const float albedo[3], float roughness, float metallic,
                              int tex_index);
int material_append_dielectric(std::vector<float> &stream, const float albedo[3], float roughness, float metallic,
                               int tex_index, float ior);

// src/ResourceManager.cpp:62-71
LightInfo make_point_light(const Mat4 &l2w, const float emission[3]);
LightInfo make_directional_light(const Mat4 &l2w, const float emission[3]);

}  // namespace rta
