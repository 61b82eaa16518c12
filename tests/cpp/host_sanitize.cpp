// Drives the host half of the library (BVH build, scene level, traversal re-layout and its
// checker, image quantisation) on random and degenerate inputs.  Meant to be built with
// -fsanitize=address,undefined (tests/test_host_sanitizers.py): the checks are the sanitizers'.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <random>
#include <string>
#include <vector>
#include "raytracing_amd.h"

static int fails = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if(!(cond)) {                                                       \
            fprintf(stderr, "%s:%d: %s: %s\n", __FILE__, __LINE__, #cond, pt_last_error()); \
            fails++;                                                        \
        }                                                                   \
    } while(0)

int main(int argc, char **argv) {
    std::mt19937 rng(3);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    const int tri_counts[] = {1, 2, 3, 257, 5000, 64};
    const int n_mesh = 6;
    std::vector<int32_t> tri_count(tri_counts, tri_counts + n_mesh);
    size_t n_tris = 0, n_nodes = 0;
    for(int c : tri_count) {
        n_tris += c;
        n_nodes += 2 * c - 1;
    }
    std::vector<float> tris(n_tris * 24, 0.0f);
    for(size_t t = 0; t < n_tris; t++) {
        const float scale = t < 6 ? 1e-6f : (t % 97 == 0 ? 1e5f : 1.0f);
        const float cx = u(rng) * scale * 5, cy = u(rng) * scale * 5, cz = u(rng) * scale * 5;
        for(int k = 0; k < 9; k++) tris[t * 24 + k] = (k % 3 == 0 ? cx : k % 3 == 1 ? cy : cz) + u(rng) * 0.1f * scale;
        if(t % 50 == 7)  // a degenerate triangle and a duplicate of its predecessor
            for(int k = 0; k < 9; k++) tris[t * 24 + k] = tris[(t - 1) * 24 + k];
        tris[t * 24 + 11] = 1.0f;
        tris[t * 24 + 14] = 1.0f;
        tris[t * 24 + 17] = 1.0f;
    }
    std::vector<uint8_t> mesh_nodes(n_nodes * 48);
    std::vector<float> root_box(n_mesh * 6);
    std::vector<int32_t> siz(n_mesh), depth(n_mesh);
    EXPECT(pt_host_build_mesh_bvh(tris.data(), tri_count.data(), n_mesh, 3, mesh_nodes.data(), root_box.data(), siz.data(),
                                  depth.data()) == PT_OK);
    const int n_inst = 9;
    std::vector<float> l2w(n_inst * 16, 0.0f);
    std::vector<int32_t> mesh_index(n_inst), material_ptr(n_inst);
    for(int i = 0; i < n_inst; i++) {
        float pos[3] = {u(rng) * 3, u(rng) * 3, u(rng) * 3}, rot[3] = {u(rng) * 180, u(rng) * 180, u(rng) * 180};
        float scl[3] = {i == 7 ? -1.0f : 1.0f + u(rng) * 0.5f, 1.0f, i == 8 ? 1e-3f : 2.0f};
        EXPECT(pt_host_transform_matrix(pos, rot, scl, &l2w[i * 16]) == PT_OK);
        mesh_index[i] = i % n_mesh;
        material_ptr[i] = i * 8;
    }
    std::vector<uint8_t> inst(n_inst * 112), scene_nodes((2 * n_inst - 1) * 48);
    int32_t stats[2];
    EXPECT(pt_host_build_scene(root_box.data(), siz.data(), depth.data(), n_mesh, l2w.data(), mesh_index.data(), material_ptr.data(),
                               n_inst, inst.data(), scene_nodes.data(), stats) == PT_OK);
    uint64_t counts[4];
    for(int fmt = -1; fmt <= 1; fmt++)
        EXPECT(pt_host_check_traversal_layout(mesh_nodes.data(), n_nodes, tris.data(), n_tris, scene_nodes.data(), 2 * n_inst - 1,
                                              inst.data(), n_inst, fmt, counts) == PT_OK);
    // the flattened tree of the same scene (rotated, mirrored, 1e-3-scaled instances; a one-triangle mesh):
    // refused by the default gate, built and checked when forced
    {
        char why[128];
        EXPECT(pt_host_check_flat_layout(mesh_nodes.data(), n_nodes, tris.data(), n_tris, scene_nodes.data(), 2 * n_inst - 1, inst.data(),
                                         n_inst, 1, counts, why, sizeof why) == PT_OK && counts[0] == 0 && why[0] != 0);
        EXPECT(pt_host_check_flat_layout(mesh_nodes.data(), n_nodes, tris.data(), n_tris, scene_nodes.data(), 2 * n_inst - 1, inst.data(),
                                         n_inst, 0, counts, why, 4) == PT_OK);
        EXPECT(counts[0] > 0 || why[0] != 0);
        // without the 1e-3 scale (whose matrix is too badly conditioned): built for sure, on several threads
        std::vector<float> l2w_b = l2w;
        {
            float pos[3] = {1, 2, 3}, rot[3] = {10, 20, 30}, scl[3] = {1.5f, 1.0f, 2.0f};
            EXPECT(pt_host_transform_matrix(pos, rot, scl, &l2w_b[8 * 16]) == PT_OK);
        }
        std::vector<uint8_t> inst_b(n_inst * 112), scene_b((2 * n_inst - 1) * 48);
        EXPECT(pt_host_build_scene(root_box.data(), siz.data(), depth.data(), n_mesh, l2w_b.data(), mesh_index.data(), material_ptr.data(),
                                   n_inst, inst_b.data(), scene_b.data(), stats) == PT_OK);
        EXPECT(pt_host_check_flat_layout(mesh_nodes.data(), n_nodes, tris.data(), n_tris, scene_b.data(), 2 * n_inst - 1, inst_b.data(),
                                         n_inst, 0, counts, why, sizeof why) == PT_OK);
        EXPECT(counts[0] == 1 + 2 + 3 + 257 + 5000 + 64 + 1 + 2 + 3);
        // (an instance rooted inside a tree and one rooted at a leaf: the flat tree takes any root)
        std::vector<uint8_t> inst3 = inst_b;
        *reinterpret_cast<int32_t *>(&inst3[0 * 112 + 68]) = n_mesh + 3;
        *reinterpret_cast<int32_t *>(&inst3[1 * 112 + 68]) = (int32_t)n_nodes - 1;
        EXPECT(pt_host_check_flat_layout(mesh_nodes.data(), n_nodes, tris.data(), n_tris, scene_b.data(), 2 * n_inst - 1, inst3.data(),
                                         n_inst, 0, counts, nullptr, 0) == PT_OK);
        EXPECT(counts[0] > 5000);
    }
    // an instance rooted inside a tree, one rooted at a leaf; then indices out of range
    {
        std::vector<uint8_t> inst2 = inst;
        auto mesh_index_of = [&](int i) { return reinterpret_cast<int32_t *>(&inst2[i * 112 + 68]); };
        *mesh_index_of(0) = n_mesh + 3;
        *mesh_index_of(1) = (int32_t)n_nodes - 1;
        for(int fmt = 0; fmt <= 1; fmt++)
            EXPECT(pt_host_check_traversal_layout(mesh_nodes.data(), n_nodes, tris.data(), n_tris, scene_nodes.data(), 2 * n_inst - 1,
                                                  inst2.data(), n_inst, fmt, counts) == PT_OK);
        *mesh_index_of(2) = (int32_t)n_nodes;
        EXPECT(pt_host_check_traversal_layout(mesh_nodes.data(), n_nodes, tris.data(), n_tris, scene_nodes.data(), 2 * n_inst - 1,
                                              inst2.data(), n_inst, 1, counts) == PT_ERR_INVALID);
    }
    // corrupted node boxes and child links
    {
        std::vector<uint8_t> bad = mesh_nodes;
        float *f = reinterpret_cast<float *>(&bad[(size_t)(n_mesh + 40) * 48]);
        f[0] = NAN;
        f[5] = INFINITY;
        f = reinterpret_cast<float *>(&bad[(size_t)(n_mesh + 90) * 48]);
        f[1] = -3e38f;
        f[5] = 3e38f;
        for(int fmt = 0; fmt <= 1; fmt++)
            EXPECT(pt_host_check_traversal_layout(bad.data(), n_nodes, tris.data(), n_tris, scene_nodes.data(), 2 * n_inst - 1, inst.data(),
                                                  n_inst, fmt, counts) == PT_OK);
        int32_t *links = reinterpret_cast<int32_t *>(&bad[(size_t)3 * 48 + 32]);  // root of the 257-triangle mesh
        links[0] = 1;  // a child before its parent
        EXPECT(pt_host_check_traversal_layout(bad.data(), n_nodes, tris.data(), n_tris, scene_nodes.data(), 2 * n_inst - 1, inst.data(),
                                              n_inst, 1, counts) == PT_ERR_INVALID);
    }
    // image quantisation with NaN / Inf / negative / huge values
    {
        std::vector<float> img = {NAN, INFINITY, -INFINITY, -1.0f, 0.0f, 0.5f, 1.0f, 1e30f, 0.999f, 2.0f, -0.0f, 1e-30f};
        std::vector<uint8_t> px(img.size());
        EXPECT(pt_host_quantize_image(img.data(), 2, 2, 3, px.data()) == PT_OK);
    }
    // Radiance codec: round trip, then every truncation of the encoded file and random garbage
    {
        const int w = 24, h = 9;
        std::vector<float> rgb(w * h * 3);
        for(float &x : rgb) x = std::fabs(u(rng)) * 40.0f;
        rgb[5] = 0.0f;
        rgb[6] = 1e-30f;
        const int64_t need = pt_host_rgbe_encode(rgb.data(), w, h, nullptr, 0);
        EXPECT(need > 0);
        std::vector<uint8_t> file((size_t)need);
        EXPECT(pt_host_rgbe_encode(rgb.data(), w, h, file.data(), file.size()) == need);
        int32_t dw = 0, dh = 0;
        EXPECT(pt_host_rgbe_decode(file.data(), file.size(), &dw, &dh, nullptr) == PT_OK && dw == w && dh == h);
        std::vector<float> back(w * h * 3);
        EXPECT(pt_host_rgbe_decode(file.data(), file.size(), &dw, &dh, back.data()) == PT_OK);
        for(size_t cut = 0; cut < file.size(); cut += 7) {
            std::vector<uint8_t> part(file.begin(), file.begin() + cut);  // exact-size buffer: overreads are seen
            (void)pt_host_rgbe_decode(part.data(), part.size(), &dw, &dh, dw == w && dh == h ? back.data() : nullptr);
        }
        for(int rep = 0; rep < 200; rep++) {
            std::vector<uint8_t> junk(file.begin(), file.end());
            for(int k = 0; k < 8; k++) junk[rng() % junk.size()] = (uint8_t)rng();
            int32_t jw = 0, jh = 0;
            if(pt_host_rgbe_decode(junk.data(), junk.size(), &jw, &jh, nullptr) == PT_OK && (size_t)jw * jh <= (size_t)w * h)
                (void)pt_host_rgbe_decode(junk.data(), junk.size(), &jw, &jh, back.data());
        }
        // sky cache of that image, of a black one and of one with a NaN
        std::vector<float> cache(w * h * 3);
        double sum = 0;
        EXPECT(pt_host_sky_cache(rgb.data(), w, h, cache.data(), &sum) == PT_OK && sum > 0);
        std::vector<float> black(w * h * 3, 0.0f);
        (void)pt_host_sky_cache(black.data(), w, h, cache.data(), &sum);
        rgb[10] = NAN;
        (void)pt_host_sky_cache(rgb.data(), w, h, cache.data(), &sum);
    }
    // PNG writer
    {
        std::vector<uint8_t> px(5 * 3 * 4);
        for(uint8_t &b : px) b = (uint8_t)rng();
        for(int ch : {1, 3, 4}) EXPECT(pt_host_write_png("/tmp/host_sanitize_out.png", 5, 3, ch, px.data()) == PT_OK);
        EXPECT(pt_host_write_png("/nonexistent-dir/x.png", 5, 3, 3, px.data()) != PT_OK);
        EXPECT(pt_host_write_png("/tmp/host_sanitize_out.png", 5, 3, 2, px.data()) != PT_OK);
    }
    // matrices
    {
        float m[16], inv[16], prod[16], proj[16];
        float pos[3] = {1, 2, 3}, rot[3] = {10, 20, 30}, scl[3] = {1, 2, 0};  // singular
        EXPECT(pt_host_transform_matrix(pos, rot, scl, m) == PT_OK);
        (void)pt_host_mat4_inverse(m, inv);
        EXPECT(pt_host_mat4_mul(m, inv, prod) == PT_OK);
        EXPECT(pt_host_camera_projection(1.0f, 1000.0f, 640, 480, proj) == PT_OK);
    }
    // ONE mesh large enough for the builder and the re-layout to split its tree over threads
    // (axis sweeps of nodes of 2^17 primitives side by side, sibling subtrees in parallel, the
    // wide-node layout cut at a frontier): the threaded parts under ThreadSanitizer
    {
        const int n = 140000;
        std::vector<float> big((size_t)n * 24, 0.0f);
        for(int t = 0; t < n; t++) {
            // coordinates on a coarse grid: many equal box centres
            const float cx = (float)(rng() % 60), cy = (float)(rng() % 60), cz = (float)(rng() % 60);
            for(int k = 0; k < 9; k++) big[(size_t)t * 24 + k] = (k % 3 == 0 ? cx : k % 3 == 1 ? cy : cz) + 0.5f * (float)(rng() % 3);
        }
        int32_t count = n, s1 = 0, d1 = 0;
        float box[6];
        std::vector<uint8_t> nodes1((size_t)(2 * n - 1) * 48), nodes8(nodes1.size());
        EXPECT(pt_host_build_mesh_bvh(big.data(), &count, 1, 1, nodes1.data(), box, &s1, &d1) == PT_OK);
        EXPECT(pt_host_build_mesh_bvh(big.data(), &count, 1, 8, nodes8.data(), box, &s1, &d1) == PT_OK);
        EXPECT(nodes1 == nodes8);
        float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        int32_t mi = 0, mp = 0, st[2];
        std::vector<uint8_t> inst1(112), scene1(48);
        EXPECT(pt_host_build_scene(box, &s1, &d1, 1, ident, &mi, &mp, 1, inst1.data(), scene1.data(), st) == PT_OK);
        EXPECT(pt_host_check_traversal_layout(nodes8.data(), (size_t)(2 * n - 1), big.data(), (size_t)n, scene1.data(), 1, inst1.data(), 1, 1,
                                              counts) == PT_OK);
    }
    // glTF binary reader: the demo model, every truncation class, and random damage to header,
    // JSON and buffer bytes; a damaged file is refused or read, never over-read
    if(argc > 1) {
        FILE *f = fopen(argv[1], "rb");
        EXPECT(f != nullptr);
        if(f) {
            fseek(f, 0, SEEK_END);
            std::vector<uint8_t> file((size_t)ftell(f));
            fseek(f, 0, SEEK_SET);
            EXPECT(fread(file.data(), 1, file.size(), f) == file.size());
            fclose(f);
            auto read_all = [&](const std::vector<uint8_t> &bytes, bool must_work) {
                pt_glb *g = nullptr;
                const int rc = pt_host_glb_open_memory(bytes.data(), bytes.size(), &g);
                if(must_work) EXPECT(rc == PT_OK);
                if(rc != PT_OK) return;
                pt_glb_info gi;
                EXPECT(pt_host_glb_info(g, &gi) == PT_OK);
                std::vector<int32_t> tc(gi.n_meshes), mesh_index(gi.n_instances), mat_index(gi.n_instances), mptr(gi.n_instances);
                std::vector<float> tri((size_t)gi.n_triangles * 24), l2w((size_t)gi.n_instances * 16);
                std::vector<pt_glb_material> mats(gi.n_materials);
                EXPECT(pt_host_glb_meshes(g, tc.data(), tri.data()) == PT_OK);
                EXPECT(pt_host_glb_materials(g, mats.data()) == PT_OK);
                EXPECT(pt_host_glb_instances(g, nullptr, mesh_index.data(), mat_index.data(), l2w.data()) == PT_OK);
                for(int k = 0; k < gi.n_textures; k++) {
                    const void *b = nullptr;
                    size_t nb = 0;
                    int32_t img = 0;
                    if(pt_host_glb_texture(g, k, &b, &nb, &img) == PT_OK && nb > 0) {
                        volatile uint8_t first = static_cast<const uint8_t *>(b)[0], last = static_cast<const uint8_t *>(b)[nb - 1];
                        (void)first;
                        (void)last;
                    }
                }
                const int64_t need = pt_host_glb_material_stream(g, nullptr, 0, nullptr);
                if(need > 0) {
                    std::vector<float> stream((size_t)need);
                    EXPECT(pt_host_glb_material_stream(g, stream.data(), stream.size(), mptr.data()) == need);
                }
                EXPECT(pt_host_glb_close(g) == PT_OK);
            };
            read_all(file, true);
            uint32_t json_len = 0;
            memcpy(&json_len, &file[12], 4);
            for(size_t cut : {(size_t)0, (size_t)11, (size_t)12, (size_t)19, (size_t)20, (size_t)20 + json_len / 2, (size_t)20 + json_len,
                              (size_t)24 + json_len, file.size() / 2, file.size() - 1}) {
                std::vector<uint8_t> part(file.begin(), file.begin() + std::min(cut, file.size()));  // exact-size buffer: over-reads are seen
                read_all(part, false);
            }
            for(int rep = 0; rep < 300; rep++) {
                std::vector<uint8_t> junk = file;
                const size_t span = rep % 3 == 0 ? 28 : (rep % 3 == 1 ? 20 + json_len : junk.size());
                const int hits = 1 + (int)(rng() % 4);
                for(int k = 0; k < hits; k++) {
                    const size_t at = rng() % span;
                    // digits become other digits now and then (indices, counts, offsets stay parseable but wrong)
                    junk[at] = (junk[at] >= '0' && junk[at] <= '9' && rng() % 2) ? (uint8_t)('0' + rng() % 10) : (uint8_t)rng();
                }
                read_all(junk, false);
            }
        }
    }
    // hand-made glTF binaries: sizes whose products wrap in size_t, an accessor that leaves its bufferView
    // while staying inside the chunk, nodes that share children (2^40 visits if walked as a tree)
    {
        auto make = [](const std::string &js_in, size_t blob_bytes) {
            std::string js = js_in;
            while(js.size() % 4) js += ' ';
            std::vector<uint8_t> out(12 + 8 + js.size() + 8 + blob_bytes, 0);
            const uint32_t hdr[3] = {0x46546C67u, 2u, (uint32_t)out.size()};
            memcpy(&out[0], hdr, 12);
            const uint32_t jc[2] = {(uint32_t)js.size(), 0x4E4F534Au}, bc[2] = {(uint32_t)blob_bytes, 0x004E4942u};
            memcpy(&out[12], jc, 8);
            memcpy(&out[20], js.data(), js.size());
            memcpy(&out[20 + js.size()], bc, 8);
            for(size_t k = 0; k + 4 <= blob_bytes; k += 4) {   // floats 0..: positions / normals; the last 12 bytes: indices 0,1,2,2,1,3
                const float f = (float)((k / 4) % 7);
                memcpy(&out[28 + js.size() + k], &f, 4);
            }
            const uint16_t idx[6] = {0, 1, 2, 2, 1, 3};
            memcpy(&out[28 + js.size() + blob_bytes - 12], idx, 12);
            return out;
        };
        auto file_with = [&](const char *view0, const char *acc0, const char *nodes) {
            std::string js = std::string("{\"scene\":0,\"scenes\":[{\"nodes\":[0]}],\"nodes\":") + nodes +
                ",\"meshes\":[{\"primitives\":[{\"attributes\":{\"POSITION\":0,\"NORMAL\":1},\"indices\":2}]}],"
                "\"bufferViews\":[" + view0 + ",{\"buffer\":0,\"byteOffset\":48,\"byteLength\":48},{\"buffer\":0,\"byteOffset\":96,\"byteLength\":12}],"
                "\"accessors\":[" + acc0 + ",{\"bufferView\":1,\"componentType\":5126,\"count\":4,\"type\":\"VEC3\"},"
                "{\"bufferView\":2,\"componentType\":5123,\"count\":6,\"type\":\"SCALAR\"}]}";
            return make(js, 108);
        };
        const char *view_ok = "{\"buffer\":0,\"byteOffset\":0,\"byteLength\":48}";
        const char *acc_ok = "{\"bufferView\":0,\"componentType\":5126,\"count\":4,\"type\":\"VEC3\"}";
        const char *nodes_ok = "[{\"mesh\":0}]";
        pt_glb *g = nullptr;
        std::vector<uint8_t> f = file_with(view_ok, acc_ok, nodes_ok);
        EXPECT(pt_host_glb_open_memory(f.data(), f.size(), &g) == PT_OK);
        if(g) pt_host_glb_close(g);
        for(const char *view : {"{\"buffer\":0,\"byteOffset\":0,\"byteLength\":48,\"byteStride\":4611686018427387904}",
                                "{\"buffer\":0,\"byteOffset\":0,\"byteLength\":48,\"byteStride\":1e300}",
                                "{\"buffer\":0,\"byteOffset\":18446744073709551616,\"byteLength\":48}",
                                "{\"buffer\":0,\"byteOffset\":-8,\"byteLength\":48}",
                                "{\"buffer\":0,\"byteOffset\":0,\"byteLength\":9223372036854775808}",
                                "{\"buffer\":0,\"byteOffset\":0,\"byteLength\":36}",
                                "{\"buffer\":0,\"byteOffset\":0,\"byteLength\":48,\"byteStride\":16}"}) {
            f = file_with(view, acc_ok, nodes_ok);
            g = nullptr;
            EXPECT(pt_host_glb_open_memory(f.data(), f.size(), &g) != PT_OK);
        }
        for(const char *acc : {"{\"bufferView\":0,\"componentType\":5126,\"count\":4,\"type\":\"VEC3\",\"byteOffset\":9223372036854775808}",
                               "{\"bufferView\":0,\"componentType\":5126,\"count\":4.5,\"type\":\"VEC3\"}",
                               "{\"bufferView\":0,\"componentType\":5126,\"count\":5,\"type\":\"VEC3\"}"}) {
            f = file_with(view_ok, acc, nodes_ok);
            g = nullptr;
            EXPECT(pt_host_glb_open_memory(f.data(), f.size(), &g) != PT_OK);
        }
        std::string dag = "[";
        for(int k = 0; k < 40; k++) dag += "{\"children\":[" + std::to_string(k + 1) + "," + std::to_string(k + 1) + "]},";
        dag += "{\"mesh\":0}]";
        f = file_with(view_ok, acc_ok, dag.c_str());
        g = nullptr;
        EXPECT(pt_host_glb_open_memory(f.data(), f.size(), &g) != PT_OK);
    }
    printf(fails ? "FAILED %d\n" : "host sanitize run ok\n", fails);
    return fails ? 1 : 0;
}
