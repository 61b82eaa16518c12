// Drives include/raytracing_amd.hpp the way the reference's update() drives its passes
// (main.cpp:169-315, without the raster pass and the denoisers): per spp use() / bind_texture /
// uniforms / draw(), then merge -> static blend -> tone mapping -> save_as_image.
// Input: a directory of raw arrays written by tests/test_gpu_facade.py; output: the raw output
// buffers and frame.png in the same directory.  Usage: facade_frame <dir> [scene.glb]
// With a .glb the program builds the scene buffers itself, the way init_scene() does
// (main.cpp:376-391: load_model, root rotation (-90, 0, 0), reload_scene), through the library's
// host helpers; only the decoded texture pixels, the sky and the frame parameters still come
// from the directory.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "raytracing_amd.hpp"

using namespace rtamd;

static std::vector<char> slurp(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if(!f) throw std::runtime_error("cannot read " + path);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
template <class T>
static void dump(const std::string &path, const std::vector<T> &v) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}

int main(int argc, char **argv) {
    if(argc < 2) {
        std::cerr << "usage: facade_frame <dir>\n";
        return 2;
    }
    const std::string dir = std::string(argv[1]) + "/";
    try {
        // meta: W H SPP n_textures, then per texture "w h c", then "sky_w sky_h"
        std::istringstream meta(std::string(slurp(dir + "meta.txt").data(), slurp(dir + "meta.txt").size()));
        int W, H, SPP, n_tex;
        meta >> W >> H >> SPP >> n_tex;
        screen_w() = W;
        screen_h() = H;
        Device::init(0);

        ResourceManager rm;
        RTAMD_MANAGER = &rm;
        std::vector<pt_tex *> textures;
        std::vector<uint64_t> handles;
        for(int k = 0; k < n_tex; k++) {
            int w, h, c;
            meta >> w >> h >> c;
            auto px = slurp(dir + "tex" + std::to_string(k) + ".bin");
            textures.push_back(create_texture_u8(w, h, c, reinterpret_cast<const unsigned char *>(px.data())));
            handles.push_back(pt_tex_handle(textures.back()));  // was glGetTextureHandleARB
        }
        int sky_w, sky_h;
        meta >> sky_w >> sky_h;
        auto sky = slurp(dir + "sky.bin"), cache = slurp(dir + "sky_cache.bin");
        pt_tex *skyTex = create_texture_f32(sky_w, sky_h, 3, reinterpret_cast<const float *>(sky.data()));
        pt_tex *cacheTex = create_texture_f32(sky_w, sky_h, 3, reinterpret_cast<const float *>(cache.data()));

        std::vector<char> tris, inst, mesh, scene, mats, lights;
        if(argc >= 3) {
            auto need = [](int rc) {
                if(rc < 0) throw std::runtime_error(pt_last_error());
            };
            pt_glb *glb = nullptr;
            need(pt_host_glb_open(argv[2], &glb));                                   // AssimpLoader::load_model, main.cpp:381
            pt_glb_info gi;
            need(pt_host_glb_info(glb, &gi));
            std::vector<int32_t> tri_count(gi.n_meshes), mesh_index(gi.n_instances), material_ptr(gi.n_instances);
            tris.resize((size_t)gi.n_triangles * sizeof(TriangleRecord));
            need(pt_host_glb_meshes(glb, tri_count.data(), tris.data()));
            // o1->transform.rotation = vec3(-90, 0, 0) under an identity scene root (main.cpp:382, Instance::matrix_to_global)
            const float zero[3] = {0, 0, 0}, one[3] = {1, 1, 1}, rot[3] = {-90, 0, 0};
            float scene_m[16], o1[16], root[16];
            need(pt_host_transform_matrix(zero, zero, one, scene_m));
            need(pt_host_transform_matrix(zero, rot, one, o1));
            need(pt_host_mat4_mul(scene_m, o1, root));
            std::vector<float> l2w((size_t)gi.n_instances * 16);
            need(pt_host_glb_instances(glb, root, mesh_index.data(), nullptr, l2w.data()));
            const int64_t n_mat = pt_host_glb_material_stream(glb, nullptr, 0, nullptr);
            need((int)n_mat);
            mats.resize((size_t)n_mat * sizeof(float));
            need((int)pt_host_glb_material_stream(glb, reinterpret_cast<float *>(mats.data()), (size_t)n_mat, material_ptr.data()));
            // Mesh::build_meshBVH + reload_meshes, then reload_scene (ResourceManager.cpp:95-224)
            size_t n_nodes = 0;
            for(int32_t c : tri_count) n_nodes += 2 * (size_t)c - 1;
            mesh.resize(n_nodes * sizeof(BVHNodeInfo));
            std::vector<float> root_box((size_t)gi.n_meshes * 6);
            std::vector<int32_t> siz(gi.n_meshes), depth(gi.n_meshes);
            need(pt_host_build_mesh_bvh(tris.data(), tri_count.data(), gi.n_meshes, 0, mesh.data(), root_box.data(), siz.data(), depth.data()));
            inst.resize((size_t)gi.n_instances * sizeof(InstanceInfo));
            scene.resize((2 * (size_t)gi.n_instances - 1) * sizeof(BVHNodeInfo));
            int32_t stats[2];
            need(pt_host_build_scene(root_box.data(), siz.data(), depth.data(), gi.n_meshes, l2w.data(), mesh_index.data(), material_ptr.data(),
                                     gi.n_instances, inst.data(), scene.data(), stats));
            need(pt_host_glb_close(glb));
            std::cout << "glb_scene " << gi.n_triangles << " triangles, " << gi.n_instances << " instances, BVH size " << stats[0]
                      << " depth " << stats[1] << std::endl;   // what main.cpp:393-395 prints
            dump(dir + "glb_tris.bin", tris); dump(dir + "glb_inst.bin", inst); dump(dir + "glb_mesh_nodes.bin", mesh);
            dump(dir + "glb_scene_nodes.bin", scene); dump(dir + "glb_materials.bin", mats);
        } else {
            tris = slurp(dir + "tris.bin"); inst = slurp(dir + "inst.bin"); mesh = slurp(dir + "mesh_nodes.bin");
            scene = slurp(dir + "scene_nodes.bin"); mats = slurp(dir + "materials.bin");
        }
        lights = slurp(dir + "lights.bin");
        // the reassignment pattern of ResourceManager.cpp:53-54: x.release(); x = SSBOBuffer<T>(n, data);
        rm.textureHandleSSBO.release();
        rm.textureHandleSSBO = SSBOBuffer<uint64_t>((uint)handles.size(), handles.data());
        rm.materialSSBO = SSBOBuffer<float>((uint)(mats.size() / 4), reinterpret_cast<float *>(mats.data()));
        rm.triangleSSBO = SSBOBuffer<TriangleRecord>((uint)(tris.size() / sizeof(TriangleRecord)), reinterpret_cast<TriangleRecord *>(tris.data()));
        rm.instanceInfoSSBO = SSBOBuffer<InstanceInfo>((uint)(inst.size() / sizeof(InstanceInfo)), reinterpret_cast<InstanceInfo *>(inst.data()));
        rm.meshBVHSSBO = SSBOBuffer<BVHNodeInfo>((uint)(mesh.size() / sizeof(BVHNodeInfo)), reinterpret_cast<BVHNodeInfo *>(mesh.data()));
        rm.sceneBVHSSBO = SSBOBuffer<BVHNodeInfo>((uint)(scene.size() / sizeof(BVHNodeInfo)), reinterpret_cast<BVHNodeInfo *>(scene.data()));
        rm.lightSSBO = SSBOBuffer<LightInfo>((uint)(lights.size() / sizeof(LightInfo)), reinterpret_cast<LightInfo *>(lights.data()));
        rm.lightCount = (int)(lights.size() / sizeof(LightInfo));

        Renderer *renderPass = new Renderer("shader/pathtracing.glsl");
        SVGFMergePass *svgfMergePass = new SVGFMergePass("shader/postprocessing/SVGF_Merge.glsl");
        StaticBlender *staticBlenderPass = new StaticBlender("shader/postprocessing/StaticBlender.glsl");
        ToneMappingGamma *mappingPass = new ToneMappingGamma("shader/postprocessing/ToneMappingGamma.glsl");

        // the denoising branch of update() (main.cpp:213-300), on buffers of its own
        svgf_spatial_filter_level() = 3;
        SVGFTemporalFilter *svgfTemporalFilterPass[2] = {new SVGFTemporalFilter("shader/postprocessing/SVGF_TemporalFilter.glsl"),
                                                         new SVGFTemporalFilter("shader/postprocessing/SVGF_TemporalFilter.glsl")};
        SVGFVarianceFilter *svgfVarianceFilterPass[2] = {new SVGFVarianceFilter("shader/postprocessing/SVGF_VarianceFilter.glsl"),
                                                         new SVGFVarianceFilter("shader/postprocessing/SVGF_VarianceFilter.glsl")};
        SVGFSpatialFilterPass *svgfSpatialFilterPass[2] = {new SVGFSpatialFilterPass("shader/postprocessing/SVGF_SpatialFilter.glsl"),
                                                           new SVGFSpatialFilterPass("shader/postprocessing/SVGF_SpatialFilter.glsl")};
        SVGFMergePass *denoisedMergePass = new SVGFMergePass("shader/postprocessing/SVGF_Merge.glsl");
        ToneMappingGamma *denoisedMappingPass = new ToneMappingGamma("shader/postprocessing/ToneMappingGamma.glsl");
        TAA *taaPass = new TAA("shader/postprocessing/TAA.glsl");
        // the reference leaves its history buffers uninitialised (glBufferData of new T[n]); zero them so that the
        // first frame is defined
        for(int k = 0; k < 2; k++) {
            SVGFTemporalFilter *t = svgfTemporalFilterPass[k];
            for(SSBOBuffer<float> *b : {&t->historyColorGBufferSSBO, &t->historyMomentGBufferSSBO, &t->historyNumSamplesGBufferSSBO,
                                        &t->historynormalGBufferSSBO, &t->historyinstanceIndexGBufferSSBO})
                b->fill(0.0f);
        }

        auto pbytes = slurp(dir + "params.bin");  // SPP records of pt_params, the values update() would compute
        if(pbytes.size() != sizeof(pt_params) * (size_t)SPP) throw std::runtime_error("params.bin has the wrong size");
        const pt_params *P = reinterpret_cast<const pt_params *>(pbytes.data());
        for(int frame = 1; frame <= 2; frame++) {
            // (the second frame hands its samples over as one batch: the loop is the same, the draws are recorded)
            if(frame == 2) renderPass->begin_batch();
            for(int spp = 1; spp <= SPP; spp++) {
                const pt_params &q = P[spp - 1];
                renderPass->use();
                {
                    renderPass->bind_texture("skybox", skyTex, 0);
                    renderPass->bind_texture("skybox_samplecache", cacheTex, 1);
                    renderPass->uniform2f("jitter", q.jitter[0], q.jitter[1]);
                    renderPass->uniform1ui("currentspp", (unsigned)spp);
                    renderPass->uniform1f("skybox_Light_SUM", q.skybox_Light_SUM);
                    renderPass->uniformMatrix4fv("v2wMat", q.v2wMat);
                    renderPass->uniform1i("SCREEN_W", W);
                    renderPass->uniform1i("SCREEN_H", H);
                    renderPass->uniform1i("WINDOW_W", q.WINDOW_W);
                    renderPass->uniform1i("WINDOW_H", q.WINDOW_H);
                    renderPass->uniform1i("MAX_DEPTH", q.MAX_DEPTH);
                    renderPass->uniform1f("cameraNear", q.cameraNear);
                    renderPass->uniform1f("cameraFar", q.cameraFar);
                    renderPass->uniform1ui("frameCounter", (unsigned)frame);
                    renderPass->uniform1f("fov", q.fov);
                    renderPass->uniform1i("SKY_W", sky_w);
                    renderPass->uniform1i("SKY_H", sky_h);
                    renderPass->uniformMatrix4fv("backprojMat", q.backprojMat);
                    renderPass->uniform1i("BRDFSampling", q.BRDFSampling);
                    renderPass->uniform1i("SkyboxSampling", q.SkyboxSampling);
                    renderPass->uniform1i("SkyboxLighting", q.SkyboxLighting);
                    renderPass->uniform1i("RasterizaionFor1st", 0);
                }
                renderPass->draw();
            }
            if(frame == 2) renderPass->end_batch();
            svgfMergePass->use();
            svgfMergePass->uniform1i("SCREEN_W", W);
            svgfMergePass->draw(renderPass->directLumGBufferSSBO, renderPass->indirectLumGBufferSSBO, renderPass->albedoGBufferSSBO);
            SSBOBuffer<float> *targetTracer_rendered = &svgfMergePass->colorGBufferSSBO;
            staticBlenderPass->use();
            staticBlenderPass->draw(*targetTracer_rendered);
            targetTracer_rendered = &staticBlenderPass->historyColorGBufferSSBO;
            mappingPass->use();
            mappingPass->draw(*targetTracer_rendered);
            {
                SSBOBuffer<float> *targetTracer_directLum = &renderPass->directLumGBufferSSBO;
                SSBOBuffer<float> *targetTracer_indirectLum = &renderPass->indirectLumGBufferSSBO;
                SSBOBuffer<float> **targets[2] = {&targetTracer_directLum, &targetTracer_indirectLum};
                for(int k = 0; k < 2; k++) {
                    svgfTemporalFilterPass[k]->use();
                    svgfTemporalFilterPass[k]->draw(**targets[k], renderPass->normalGBufferSSBO, renderPass->instanceIndexGBufferSSBO,
                                                    renderPass->motionGBufferSSBO);
                    svgfVarianceFilterPass[k]->use();
                    svgfVarianceFilterPass[k]->draw(svgfTemporalFilterPass[k]->outputMomentGBufferSSBO, renderPass->normalGBufferSSBO,
                                                    renderPass->depthGBufferSSBO, svgfTemporalFilterPass[k]->outputNumSamplesGBufferSSBO);
                    svgfSpatialFilterPass[k]->use();
                    svgfSpatialFilterPass[k]->draw(svgfTemporalFilterPass[k]->outputColorGBufferSSBO, svgfVarianceFilterPass[k]->varianceGBufferSSBO,
                                                   renderPass->normalGBufferSSBO, renderPass->depthGBufferSSBO);
                    *targets[k] = &svgfSpatialFilterPass[k]->outputColorGBufferSSBO;
                    svgfTemporalFilterPass[k]->update_historycolor(**targets[k]);
                }
                denoisedMergePass->draw(*targetTracer_directLum, *targetTracer_indirectLum, renderPass->albedoGBufferSSBO);
                denoisedMappingPass->draw(denoisedMergePass->colorGBufferSSBO);
                taaPass->draw(denoisedMappingPass->outputColorGBufferSSBO, renderPass->motionGBufferSSBO, renderPass->normalGBufferSSBO,
                              renderPass->instanceIndexGBufferSSBO);
                if(frame == 2) {
                    dump(dir + "out_svgf_di.bin", targetTracer_directLum->download());
                    dump(dir + "out_svgf_idi.bin", targetTracer_indirectLum->download());
                    dump(dir + "out_svgf_num.bin", svgfTemporalFilterPass[0]->outputNumSamplesGBufferSSBO.download());
                    dump(dir + "out_taa.bin", taaPass->outputColorGBufferSSBO.download());
                }
            }
            if(frame == 1) {
                dump(dir + "out_directLum_f1.bin", renderPass->directLumGBufferSSBO.download());
                dump(dir + "out_color_f1.bin", svgfMergePass->colorGBufferSSBO.download());
            }
        }
        dump(dir + "out_directLum.bin", renderPass->directLumGBufferSSBO.download());
        dump(dir + "out_indirectLum.bin", renderPass->indirectLumGBufferSSBO.download());
        dump(dir + "out_albedo.bin", renderPass->albedoGBufferSSBO.download());
        dump(dir + "out_normal.bin", renderPass->normalGBufferSSBO.download());
        dump(dir + "out_depth.bin", renderPass->depthGBufferSSBO.download());
        dump(dir + "out_instanceIndex.bin", renderPass->instanceIndexGBufferSSBO.download());
        dump(dir + "out_motion.bin", renderPass->motionGBufferSSBO.download());
        dump(dir + "out_history.bin", staticBlenderPass->historyColorGBufferSSBO.download());
        dump(dir + "out_ldr.bin", mappingPass->outputColorGBufferSSBO.download());
        mappingPass->outputColorGBufferSSBO.save_as_image(W, H, 3, dir + "frame.png");

        // SSBOBuffer odds and ends: copy(buffer), copy(data), fill, isempty, zero-length buffers
        SSBOBuffer<float> a(16), b(16), empty, zero(0);
        a.fill(2.5f);
        b.copy(&a);
        std::vector<float> host(16, 7.0f);
        a.copy(host.data());
        empty.fill(1.0f);   // no-op on an empty buffer, as in the reference
        empty.copy(host.data());
        bool ok = empty.isempty() && !zero.isempty();
        for(float x : b.download()) ok = ok && x == 2.5f;
        for(float x : a.download()) ok = ok && x == 7.0f;
        a.release(); b.release(); zero.release(); empty.release();
        ok = ok && a.isempty();
        std::cout << (ok ? "ssbo_ok" : "ssbo_FAILED") << std::endl;

        for(int k = 0; k < 2; k++) { delete svgfTemporalFilterPass[k]; delete svgfVarianceFilterPass[k]; delete svgfSpatialFilterPass[k]; }
        delete taaPass; delete denoisedMappingPass; delete denoisedMergePass;
        delete mappingPass; delete staticBlenderPass; delete svgfMergePass; delete renderPass;
        for(pt_tex *t : textures) pt_tex_release(t);
        pt_tex_release(skyTex); pt_tex_release(cacheTex);
        RTAMD_MANAGER = nullptr;
    } catch(const std::exception &e) {
        std::cerr << "facade_frame: " << e.what() << std::endl;
        return 1;
    }
    // rm's buffers are released by its destructor above; the context last
    Device::shutdown();
    return 0;
}
