#include "scene_host.h"
#include <atomic>
#include <cassert>
#include <cmath>
#include <thread>

namespace rta {

Mat4 transform_matrix(const float position[3], const float rotation_deg[3], const float scale[3]) {
    // the reference's M_PI is `const float M_PI = acos(-1)` (include/common.h:11)
    const float pi_f = (float)std::acos(-1.0);
    const float s_rows[16] = {scale[0], 0, 0, 0, 0, scale[1], 0, 0, 0, 0, scale[2], 0, 0, 0, 0, 1};
    // Transform.cpp:21-24 fills s with a column-major initializer list; it is diagonal,
    // so rows and columns coincide.
    Mat4 s = Mat4::by_rows(s_rows);
    float x = rotation_deg[0] / 180 * pi_f;
    float y = rotation_deg[1] / 180 * pi_f;
    float z = rotation_deg[2] / 180 * pi_f;
    const float rx[16] = {1, 0, 0, 0, 0, std::cos(x), -std::sin(x), 0, 0, std::sin(x), std::cos(x), 0, 0, 0, 0, 1};
    const float ry[16] = {std::cos(y), 0, std::sin(y), 0, 0, 1, 0, 0, -std::sin(y), 0, std::cos(y), 0, 0, 0, 0, 1};
    const float rz[16] = {std::cos(z), -std::sin(z), 0, 0, std::sin(z), std::cos(z), 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    Mat4 rot_x = Mat4::by_rows(rx), rot_y = Mat4::by_rows(ry), rot_z = Mat4::by_rows(rz);
    Mat4 res = Mat4::identity();
    res = mul(res, rot_y);
    res = mul(res, rot_x);
    res = mul(res, rot_z);
    res = mul(res, s);
    res.c[3] = Vec4(position[0], position[1], position[2], 1.0f);
    return res;
}

Mat4 camera_projection(float fov_x, float far_plane, int screen_w, int screen_h) {
    float near_plane = (float)(1.0 / std::tan(fov_x / 2));
    float scaleY = (float)(1.0 * screen_h / screen_w);
    float n = -near_plane;
    float f = -far_plane;
    const float rows[16] = {n, 0, 0, 0,
                            0, n / scaleY, 0, 0,
                            0, 0, (n + f) / (f - n), -2 * n * f / (f - n),
                            0, 0, 1, 0};
    Mat4 m = Mat4::by_rows(rows);
    // glm's unary minus on a matrix is 0 - x per component (the vendored glm's type_vec4.inl), so zeros stay +0
    for(int j = 0; j < 4; j++) m.c[j] = Vec4(0.0f - m.c[j].x, 0.0f - m.c[j].y, 0.0f - m.c[j].z, 0.0f - m.c[j].w);
    return m;
}

void build_mesh_set(const TriangleRec *tris, const int *tri_count, int n_mesh, int n_threads, MeshSet &out) {
    out.tri_base.assign(n_mesh, 0);
    out.tri_count.assign(tri_count, tri_count + n_mesh);
    int base = 0;
    for(int m = 0; m < n_mesh; m++) {
        assert(tri_count[m] > 0);
        out.tri_base[m] = base;
        base += tri_count[m];
    }
    std::vector<SahTree> trees(n_mesh);
    // Meshes are built side by side; when there are fewer meshes than threads, each tree also gets a share
    // of the threads for itself (a single huge mesh gets all of them).
    const int per_tree = n_threads > n_mesh ? (n_threads + n_mesh - 1) / n_mesh : 1;
    auto work = [&](int m) {
        int n = tri_count[m];
        std::vector<Box> boxes(n);
        const TriangleRec *t = tris + out.tri_base[m];
        for(int i = 0; i < n; i++)
            for(int v = 0; v < 3; v++) boxes[i].add_point(&t[i].ver[v * 3]);
        build_sah_tree(boxes.data(), n, trees[m], per_tree);
    };
    if(n_threads <= 1 || n_mesh == 1) {
        for(int m = 0; m < n_mesh; m++) work(m);  // (one mesh: its tree uses the threads inside)
    } else {
        std::atomic<int> next(0);
        std::vector<std::thread> pool;
        int nt = n_threads < n_mesh ? n_threads : n_mesh;
        for(int t = 0; t < nt; t++)
            pool.emplace_back([&] {
                for(int m; (m = next.fetch_add(1)) < n_mesh;) work(m);
            });
        for(auto &th : pool) th.join();
    }
    out.root_box.resize(n_mesh);
    out.siz.resize(n_mesh);
    out.depth.resize(n_mesh);
    for(int m = 0; m < n_mesh; m++) {
        const SahNode &r = trees[m].nodes[trees[m].root];
        out.root_box[m] = r.box;
        out.siz[m] = r.siz;
        out.depth[m] = r.depth;
    }
    flatten_mesh_trees(trees, out.tri_base, out.nodes);
}

namespace {
// siz / depth of the two-level tree the way BVHNode::build books them (BVH.cpp:23-24,79-80):
// an instance leaf counts as its mesh tree's size (not +1) and its height + 1.
void two_level_stats(const SahTree &t, int k, const MeshSet &meshes, const int *mesh_index, int &siz, int &depth) {
    const SahNode &p = t.nodes[k];
    if(p.prim >= 0) {
        siz = meshes.siz[mesh_index[p.prim]];
        depth = meshes.depth[mesh_index[p.prim]] + 1;
        return;
    }
    int sl, dl, sr, dr;
    two_level_stats(t, p.left, meshes, mesh_index, sl, dl);
    two_level_stats(t, p.right, meshes, mesh_index, sr, dr);
    siz = sl + sr + 1;
    depth = (dl > dr ? dl : dr) + 1;
}
}  // namespace

void build_scene_level(const MeshSet &meshes, const Mat4 *l2w, const int *mesh_index, const int *material_ptr,
                       int n_inst, SceneLevel &out) {
    assert(n_inst > 0);
    std::vector<Box> world(n_inst);
    out.instances.resize(n_inst);
    for(int i = 0; i < n_inst; i++) {
        const Box &cube = meshes.root_box[mesh_index[i]];
        for(int c = 0; c < 8; c++) {
            Vec4 point((c & 1) ? cube.mi[0] : cube.mx[0], (c & 2) ? cube.mi[1] : cube.mx[1],
                       (c & 4) ? cube.mi[2] : cube.mx[2], 1.0f);
            Vec4 w = mul(l2w[i], point);
            const float p[3] = {w.x, w.y, w.z};
            world[i].add_point(p);
        }
        InstanceInfo &y = out.instances[i];
        Mat4 inv = inverse(l2w[i]);
        for(int k = 0; k < 16; k++) y.world2local[k] = inv.data()[k];
        y.materialPtr = material_ptr[i];
        y.meshIndex = mesh_index[i];
        for(int k = 0; k < 10; k++) y.emptyblock[k] = 0;
    }
    SahTree tree;
    build_sah_tree(world.data(), n_inst, tree);
    flatten_scene_tree(tree, out.nodes);
    two_level_stats(tree, tree.root, meshes, mesh_index, out.siz, out.depth);
}

static int material_append_common(std::vector<float> &stream, float type, const float albedo[3], float roughness,
                                  float metallic, int tex_index) {
    int ptr = (int)stream.size();
    stream.push_back(type);
    stream.push_back(albedo[0]);
    stream.push_back(albedo[1]);
    stream.push_back(albedo[2]);
    stream.push_back(roughness);
    stream.push_back(metallic);
    stream.push_back(tex_index >= 0 ? (float)tex_index : -1.0f);
    return ptr;
}

int material_append_conductor(std::vector<float> &stream, const float albedo[3], float roughness, float metallic,
                              int tex_index) {
    return material_append_common(stream, 1.0f, albedo, roughness, metallic, tex_index);
}

int material_append_dielectric(std::vector<float> &stream, const float albedo[3], float roughness, float metallic,
                               int tex_index, float ior) {
    int ptr = material_append_common(stream, 2.0f, albedo, roughness, metallic, tex_index);
    stream.push_back(ior);
    return ptr;
}

LightInfo make_point_light(const Mat4 &l2w, const float emission[3]) {
    Vec4 p = mul(l2w, Vec4(0, 0, 0, 1.0f));
    LightInfo li;
    li.type = 0;
    li.x = p.x; li.y = p.y; li.z = p.z;
    li.r = emission[0]; li.g = emission[1]; li.b = emission[2];
    li.emptyblock = 0.0f;
    return li;
}

LightInfo make_directional_light(const Mat4 &l2w, const float emission[3]) {
    Vec4 d = mul(l2w, Vec4(0, 0, 1, 0.0f));
    LightInfo li;
    li.type = 1;
    li.x = d.x; li.y = d.y; li.z = d.z;
    li.r = emission[0]; li.g = emission[1]; li.b = emission[2];
    li.emptyblock = 0.0f;
    return li;
}

}  // namespace rta
