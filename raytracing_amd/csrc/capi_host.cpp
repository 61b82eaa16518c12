// Host-only half of the C ABI: scene-buffer producers.  Nothing here calls HIP.
#include <cstring>
#include <random>
#include <thread>
#include <vector>
#include "capi_common.h"
#include "host/bvh_builder.h"
#include "host/scene_host.h"
#include "host/sky_host.h"
#include "host/png_writer.h"
#include "host/accel_layout.h"
#include "host/glb_loader.h"
#include <cstdio>

namespace rta {
static thread_local std::string g_last_error;
int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}
}  // namespace rta

using namespace rta;

extern "C" {

const char *pt_last_error(void) { return g_last_error.c_str(); }
const char *pt_version(void) { return "raytracing_amd 0.1 (gfx950)"; }

int pt_host_build_mesh_bvh(const void *tris, const int32_t *tri_count, int32_t n_mesh, int32_t n_threads,
                           void *nodes_out, float *root_box_out, int32_t *siz_out, int32_t *depth_out) {
    if(!tris || !tri_count || n_mesh <= 0 || !nodes_out) return fail(PT_ERR_INVALID, "pt_host_build_mesh_bvh: null argument or n_mesh <= 0");
    for(int m = 0; m < n_mesh; m++)
        if(tri_count[m] <= 0) return fail(PT_ERR_INVALID, "pt_host_build_mesh_bvh: every mesh needs at least one triangle");
    if(n_threads <= 0) {
        unsigned hc = std::thread::hardware_concurrency();
        n_threads = hc ? (hc > 16 ? 16 : (int)hc) : 1;
    }
    MeshSet ms;
    build_mesh_set(static_cast<const TriangleRec *>(tris), tri_count, n_mesh, n_threads, ms);
    memcpy(nodes_out, ms.nodes.data(), ms.nodes.size() * sizeof(BVHNodeInfo));
    for(int m = 0; m < n_mesh; m++) {
        if(root_box_out) {
            for(int k = 0; k < 3; k++) {
                root_box_out[m * 6 + k] = ms.root_box[m].mi[k];
                root_box_out[m * 6 + 3 + k] = ms.root_box[m].mx[k];
            }
        }
        if(siz_out) siz_out[m] = ms.siz[m];
        if(depth_out) depth_out[m] = ms.depth[m];
    }
    return PT_OK;
}

int pt_host_build_scene(const float *mesh_root_box, const int32_t *mesh_siz, const int32_t *mesh_depth,
                        int32_t n_mesh, const float *l2w, const int32_t *mesh_index, const int32_t *material_ptr,
                        int32_t n_inst, void *instance_info_out, void *scene_nodes_out, int32_t *stats_out) {
    if(!mesh_root_box || !l2w || !mesh_index || !material_ptr || !instance_info_out || !scene_nodes_out)
        return fail(PT_ERR_INVALID, "pt_host_build_scene: null argument");
    if(n_mesh <= 0 || n_inst <= 0) return fail(PT_ERR_INVALID, "pt_host_build_scene: needs at least one mesh and one instance");
    MeshSet ms;
    ms.root_box.resize(n_mesh);
    ms.siz.assign(n_mesh, 1);
    ms.depth.assign(n_mesh, 1);
    for(int m = 0; m < n_mesh; m++) {
        for(int k = 0; k < 3; k++) {
            ms.root_box[m].mi[k] = mesh_root_box[m * 6 + k];
            ms.root_box[m].mx[k] = mesh_root_box[m * 6 + 3 + k];
        }
        if(mesh_siz) ms.siz[m] = mesh_siz[m];
        if(mesh_depth) ms.depth[m] = mesh_depth[m];
    }
    for(int i = 0; i < n_inst; i++)
        if(mesh_index[i] < 0 || mesh_index[i] >= n_mesh) return fail(PT_ERR_INVALID, "pt_host_build_scene: mesh index out of range");
    std::vector<Mat4> mats(n_inst);
    for(int i = 0; i < n_inst; i++) memcpy(mats[i].data(), l2w + i * 16, 64);
    SceneLevel sl;
    build_scene_level(ms, mats.data(), mesh_index, material_ptr, n_inst, sl);
    memcpy(instance_info_out, sl.instances.data(), sl.instances.size() * sizeof(InstanceInfo));
    memcpy(scene_nodes_out, sl.nodes.data(), sl.nodes.size() * sizeof(BVHNodeInfo));
    if(stats_out) { stats_out[0] = sl.siz; stats_out[1] = sl.depth; }
    return PT_OK;
}

int pt_host_check_traversal_layout(const void *mesh_nodes, size_t n_mesh_nodes, const void *tris, size_t n_tris,
                                   const void *scene_nodes, size_t n_scene_nodes, const void *instances, size_t n_inst,
                                   int32_t format, uint64_t counts_out[4]) {
    if(!mesh_nodes || !tris || !scene_nodes || !instances) return fail(PT_ERR_INVALID, "pt_host_check_traversal_layout: null argument");
    if(format < -1 || format > 1) return fail(PT_ERR_INVALID, "pt_host_check_traversal_layout: format must be -1, 0 or 1");
    const BVHNodeInfo *mesh = static_cast<const BVHNodeInfo *>(mesh_nodes), *scene = static_cast<const BVHNodeInfo *>(scene_nodes);
    const InstanceInfo *inst = static_cast<const InstanceInfo *>(instances);
    std::string err;
    AccelMeshLevel ml;
    AccelSceneLevel sl;
    if(!build_accel_mesh_level(mesh, n_mesh_nodes, static_cast<const TriangleRec *>(tris), n_tris, {}, format, ml, err))
        return fail(PT_ERR_INVALID, "scene buffers rejected: " + err);
    bool missing = false;
    bool ok = build_accel_scene_level(mesh, n_mesh_nodes, ml, scene, n_scene_nodes, inst, n_inst, sl, err, &missing);
    if(!ok && missing) {
        std::vector<int32_t> roots;
        for(size_t i = 0; i < n_inst; i++) roots.push_back(inst[i].meshIndex);
        if(!build_accel_mesh_level(mesh, n_mesh_nodes, static_cast<const TriangleRec *>(tris), n_tris, roots, format, ml, err))
            return fail(PT_ERR_INVALID, "scene buffers rejected: " + err);
        ok = build_accel_scene_level(mesh, n_mesh_nodes, ml, scene, n_scene_nodes, inst, n_inst, sl, err);
    }
    if(!ok) return fail(PT_ERR_INVALID, "scene buffers rejected: " + err);
    uint64_t counts[4];
    if(!verify_accel_levels(mesh, n_mesh_nodes, ml, scene, n_scene_nodes, n_inst, sl, counts, err))
        return fail(PT_ERR_STATE, "traversal layout check failed: " + err);
    if(counts_out) memcpy(counts_out, counts, sizeof counts);
    return PT_OK;
}

int pt_host_check_flat_layout(const void *mesh_nodes, size_t n_mesh_nodes, const void *tris, size_t n_tris, const void *scene_nodes,
                              size_t n_scene_nodes, const void *instances, size_t n_inst, int32_t only_axis_aligned,
                              uint64_t counts_out[4], char *why_not, size_t why_cap) {
    if(!mesh_nodes || !tris || !scene_nodes || !instances) return fail(PT_ERR_INVALID, "pt_host_check_flat_layout: null argument");
    const BVHNodeInfo *mesh = static_cast<const BVHNodeInfo *>(mesh_nodes), *scene = static_cast<const BVHNodeInfo *>(scene_nodes);
    const InstanceInfo *inst = static_cast<const InstanceInfo *>(instances);
    const TriangleRec *tr = static_cast<const TriangleRec *>(tris);
    std::string err;
    {
        // (the flat builder relies on what these two have validated)
        AccelMeshLevel ml;
        AccelSceneLevel sl;
        if(!build_accel_mesh_level(mesh, n_mesh_nodes, tr, n_tris, {}, 1, ml, err)) return fail(PT_ERR_INVALID, "scene buffers rejected: " + err);
        bool missing = false;
        bool ok = build_accel_scene_level(mesh, n_mesh_nodes, ml, scene, n_scene_nodes, inst, n_inst, sl, err, &missing);
        if(!ok && !missing) return fail(PT_ERR_INVALID, "scene buffers rejected: " + err);
    }
    if(why_not && why_cap) why_not[0] = 0;
    uint64_t counts[4] = {0, 0, 0, 0};
    AccelFlat fl;
    std::string why;
    if(!build_accel_flat(mesh, n_mesh_nodes, tr, n_tris, scene, n_scene_nodes, inst, n_inst, (size_t)1 << 27, only_axis_aligned != 0, 0, fl, why)) {
        if(why_not && why_cap) snprintf(why_not, why_cap, "%s", why.c_str());
        if(counts_out) memcpy(counts_out, counts, sizeof counts);
        return PT_OK;
    }
    if(!verify_accel_flat(mesh, n_mesh_nodes, tr, inst, n_inst, fl, counts, err)) return fail(PT_ERR_STATE, "flat layout check failed: " + err);
    if(counts_out) memcpy(counts_out, counts, sizeof counts);
    return PT_OK;
}

int pt_host_transform_matrix(const float position[3], const float rotation_deg[3], const float scale[3],
                             float out16[16]) {
    if(!position || !rotation_deg || !scale || !out16) return fail(PT_ERR_INVALID, "pt_host_transform_matrix: null argument");
    Mat4 m = transform_matrix(position, rotation_deg, scale);
    memcpy(out16, m.data(), 64);
    return PT_OK;
}

int pt_host_mat4_mul(const float a[16], const float b[16], float out16[16]) {
    if(!a || !b || !out16) return fail(PT_ERR_INVALID, "pt_host_mat4_mul: null argument");
    Mat4 ma, mb;
    memcpy(ma.data(), a, 64);
    memcpy(mb.data(), b, 64);
    Mat4 r = mul(ma, mb);
    memcpy(out16, r.data(), 64);
    return PT_OK;
}

int pt_host_mat4_inverse(const float m[16], float out16[16]) {
    if(!m || !out16) return fail(PT_ERR_INVALID, "pt_host_mat4_inverse: null argument");
    Mat4 a;
    memcpy(a.data(), m, 64);
    Mat4 r = inverse(a);
    memcpy(out16, r.data(), 64);
    return PT_OK;
}

int pt_host_camera_projection(float fov_x, float far_plane, int32_t screen_w, int32_t screen_h, float out16[16]) {
    if(!out16 || screen_w <= 0 || screen_h <= 0) return fail(PT_ERR_INVALID, "pt_host_camera_projection: bad argument");
    Mat4 m = camera_projection(fov_x, far_plane, screen_w, screen_h);
    memcpy(out16, m.data(), 64);
    return PT_OK;
}

int pt_host_frame_jitters(uint64_t frame, int32_t spp_per_frame, int32_t static_blender, float *xy) {
    if(!xy || spp_per_frame <= 0) return fail(PT_ERR_INVALID, "pt_host_frame_jitters: bad argument");
    std::mt19937 rand(0);
    rand.discard(frame * 2ull * (uint64_t)spp_per_frame);
    for(int32_t s = 0; s < spp_per_frame; s++) {
        const float x = (float)rand() * 1.0f / (float)rand.max();
        const float y = (float)rand() * 1.0f / (float)rand.max();
        const bool centre = !static_blender && s == spp_per_frame - 1;
        xy[2 * s] = centre ? 0.5f : x;
        xy[2 * s + 1] = centre ? 0.5f : y;
    }
    return PT_OK;
}

int pt_host_rgbe_decode(const void *bytes, size_t n, int32_t *w, int32_t *h, float *rgb_out) {
    if(!bytes || !w || !h) return fail(PT_ERR_INVALID, "pt_host_rgbe_decode: null argument");
    std::vector<float> rgb;
    std::string err;
    int ww, hh;
    if(!rgbe_decode(static_cast<const uint8_t *>(bytes), n, ww, hh, rgb, err)) return fail(PT_ERR_INVALID, "pt_host_rgbe_decode: " + err);
    *w = ww;
    *h = hh;
    if(rgb_out) memcpy(rgb_out, rgb.data(), rgb.size() * sizeof(float));
    return PT_OK;
}

int64_t pt_host_rgbe_encode(const float *rgb, int32_t w, int32_t h, void *out, size_t cap) {
    if(!rgb || w <= 0 || h <= 0) return fail(PT_ERR_INVALID, "pt_host_rgbe_encode: bad argument");
    std::vector<uint8_t> bytes;
    rgbe_encode(rgb, w, h, bytes);
    if(out) {
        if(cap < bytes.size()) return fail(PT_ERR_INVALID, "pt_host_rgbe_encode: output buffer too small");
        memcpy(out, bytes.data(), bytes.size());
    }
    return (int64_t)bytes.size();
}

int pt_host_sky_cache(const float *rgb, int32_t w, int32_t h, float *cache_out, double *light_sum_out) {
    if(!rgb || w <= 0 || h <= 0 || !cache_out || !light_sum_out) return fail(PT_ERR_INVALID, "pt_host_sky_cache: bad argument");
    std::vector<float> cache;
    sky_build_cache(rgb, w, h, cache, *light_sum_out);
    memcpy(cache_out, cache.data(), cache.size() * sizeof(float));
    return PT_OK;
}

int pt_host_quantize_image(const float *data, int32_t width, int32_t height, int32_t channel, uint8_t *pixels_out) {
    if(!data || !pixels_out || width <= 0 || height <= 0 || (channel != 1 && channel != 3 && channel != 4))
        return fail(PT_ERR_INVALID, "pt_host_quantize_image: bad argument");
    std::vector<uint8_t> tmp;
    rta::quantize_flip(data, width, height, channel, tmp);
    memcpy(pixels_out, tmp.data(), tmp.size());
    return PT_OK;
}

int pt_host_write_png(const char *path, int32_t width, int32_t height, int32_t channel, const uint8_t *pixels) {
    if(!path) return fail(PT_ERR_INVALID, "pt_host_write_png: null path");
    std::string err;
    if(!rta::write_png(path, width, height, channel, pixels, err)) return fail(PT_ERR_INVALID, err);
    return PT_OK;
}

// ---- glTF binary scenes: the arrays AssimpLoader::load_model would hand to the scene classes ----
struct pt_glb {
    GlbScene scene;
};

int pt_host_glb_open_memory(const void *bytes, size_t n, pt_glb **out) {
    if(!bytes || !out) return fail(PT_ERR_INVALID, "pt_host_glb_open_memory: null argument");
    pt_glb *g = new pt_glb;
    std::string err;
    if(!load_glb(static_cast<const uint8_t *>(bytes), n, g->scene, err)) {
        delete g;
        return fail(PT_ERR_INVALID, err);
    }
    *out = g;
    return PT_OK;
}

int pt_host_glb_open(const char *path, pt_glb **out) {
    if(!path || !out) return fail(PT_ERR_INVALID, "pt_host_glb_open: null argument");
    FILE *f = fopen(path, "rb");
    if(!f) return fail(PT_ERR_INVALID, std::string("pt_host_glb_open: cannot read ") + path);
    std::vector<uint8_t> data;
    uint8_t buf[1 << 16];
    size_t got;
    while((got = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + got);
    fclose(f);
    return pt_host_glb_open_memory(data.data(), data.size(), out);
}

int pt_host_glb_close(pt_glb *glb) {
    delete glb;
    return PT_OK;
}

int pt_host_glb_info(const pt_glb *glb, pt_glb_info *out) {
    if(!glb || !out) return fail(PT_ERR_INVALID, "pt_host_glb_info: null argument");
    out->n_meshes = (int32_t)glb->scene.tri_count.size();
    out->n_materials = (int32_t)glb->scene.materials.size();
    out->n_textures = (int32_t)glb->scene.textures.size();
    out->n_instances = (int32_t)glb->scene.instances.size();
    out->n_triangles = (int64_t)glb->scene.triangles.size();
    return PT_OK;
}

int pt_host_glb_meshes(const pt_glb *glb, int32_t *tri_count_out, void *triangles_out) {
    if(!glb) return fail(PT_ERR_INVALID, "pt_host_glb_meshes: null argument");
    const GlbScene &s = glb->scene;
    if(tri_count_out) memcpy(tri_count_out, s.tri_count.data(), s.tri_count.size() * sizeof(int32_t));
    if(triangles_out) memcpy(triangles_out, s.triangles.data(), s.triangles.size() * sizeof(TriangleRec));
    return PT_OK;
}

int pt_host_glb_materials(const pt_glb *glb, pt_glb_material *out) {
    if(!glb || !out) return fail(PT_ERR_INVALID, "pt_host_glb_materials: null argument");
    static_assert(sizeof(pt_glb_material) == sizeof(GlbMaterial), "material records must match");
    memcpy(out, glb->scene.materials.data(), glb->scene.materials.size() * sizeof(GlbMaterial));
    return PT_OK;
}

int pt_host_glb_texture(const pt_glb *glb, int32_t index, const void **bytes_out, size_t *n_out, int32_t *image_index_out) {
    if(!glb || !bytes_out || !n_out) return fail(PT_ERR_INVALID, "pt_host_glb_texture: null argument");
    if(index < 0 || (size_t)index >= glb->scene.textures.size()) return fail(PT_ERR_INVALID, "pt_host_glb_texture: index out of range");
    const GlbTexture &t = glb->scene.textures[index];
    *bytes_out = glb->scene.file.data() + t.offset;
    *n_out = t.bytes;
    if(image_index_out) *image_index_out = t.image;
    return PT_OK;
}

int pt_host_glb_instances(const pt_glb *glb, const float root_l2w[16], int32_t *mesh_index_out, int32_t *material_index_out,
                          float *l2w_out) {
    if(!glb) return fail(PT_ERR_INVALID, "pt_host_glb_instances: null argument");
    Mat4 root = Mat4::identity();
    if(root_l2w) memcpy(root.data(), root_l2w, 64);
    const GlbScene &s = glb->scene;
    for(size_t i = 0; i < s.instances.size(); i++) {
        if(mesh_index_out) mesh_index_out[i] = s.instances[i].mesh;
        if(material_index_out) material_index_out[i] = s.instances[i].material;
        if(l2w_out) {
            const Mat4 m = root_l2w ? mul(root, s.instances[i].l2w) : s.instances[i].l2w;
            memcpy(l2w_out + i * 16, m.data(), 64);
        }
    }
    return PT_OK;
}

int64_t pt_host_glb_material_stream(const pt_glb *glb, float *stream_out, size_t cap_floats, int32_t *material_ptr_out) {
    if(!glb) return fail(PT_ERR_INVALID, "pt_host_glb_material_stream: null argument");
    const GlbScene &s = glb->scene;
    std::vector<float> stream;
    for(size_t i = 0; i < s.instances.size(); i++) {
        const int m = s.instances[i].material;
        int ptr;
        if(m >= 0) {
            const GlbMaterial &gm = s.materials[m];
            ptr = material_append_conductor(stream, gm.albedo, gm.roughness, gm.metallic, gm.texture);
        } else {
            // an instance without a material keeps the class defaults (src/material/RoughConductor.h)
            const float white[3] = {1.0f, 1.0f, 1.0f};
            ptr = material_append_conductor(stream, white, 1.0f, 0.0f, -1);
        }
        if(material_ptr_out) material_ptr_out[i] = ptr;
    }
    if(stream_out) {
        if(cap_floats < stream.size()) return fail(PT_ERR_INVALID, "pt_host_glb_material_stream: output holds fewer floats than the stream");
        memcpy(stream_out, stream.data(), stream.size() * sizeof(float));
    }
    return (int64_t)stream.size();
}

}  // extern "C"
