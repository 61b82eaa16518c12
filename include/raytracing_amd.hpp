// C++ facade over the C ABI (raytracing_amd.h) with the reference's class and member names, so
// that the reference's frame loop (main.cpp:169-315) keeps its shape:
//
//   SSBOBuffer<T>        src/SSBOBuffer.h:19-37          same constructors and methods
//   RenderPass           src/renderpass/RenderPass.h:13-27   use(), bind_texture()
//   Renderer             src/renderpass/Renderer.h:16-31     seven public outputs, draw()
//   SVGFMergePass        src/renderpass/SVGFMergePass.h:11-22
//   StaticBlender        src/renderpass/StaticBlender.h
//   ToneMappingGamma     src/renderpass/ToneMappingGamma.h
//   SVGFTemporalFilter, SVGFVarianceFilter, SVGFSpatialFilterPass, TAA   src/renderpass/*.h
//   ResourceManager      src/ResourceManager.h:80-90     the seven SSBO members + manager pointer
//
// What changes for a caller: GL object names become handles (pt_tex* instead of GLuint), and
// the glUniform*(glGetUniformLocation(pass->shaderProgram, "name"), ...) block of
// main.cpp:185-205 becomes pass->uniform*("name", ...) with the same names and values.
// Header only; link against libraytracing_amd.so.  Errors: the reference has no error
// channel (SSBOBuffer.h, RenderPass.cpp:60-84 print and continue); here every failed call
// throws std::runtime_error carrying pt_last_error().
#pragma once
#include <algorithm>
#include <cassert>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "raytracing_amd.h"

namespace rtamd {

using std::string;
using uint = unsigned int;

inline void check(int rc) {
    if(rc != PT_OK) throw std::runtime_error(pt_last_error());
}

// The process-wide device context (the reference has one GL context, main.cpp:392-411).
class Device {
public:
    static pt_ctx *&slot() {
        static pt_ctx *c = nullptr;
        return c;
    }
    static void init(int device_id = 0) {
        if(!slot()) check(pt_ctx_create(device_id, &slot()));
    }
    static pt_ctx *ctx() {
        init();
        return slot();
    }
    static void synchronize() { check(pt_ctx_synchronize(ctx())); }
    static void shutdown() {
        if(slot()) pt_ctx_destroy(slot());
        slot() = nullptr;
    }
};

// Compile-time constants in the reference (Config.h:14-15); set before constructing passes.
inline int &screen_w() { static int w = 640; return w; }
inline int &screen_h() { static int h = 480; return h; }

class RenderPass;
inline RenderPass *&current_pass() {
    static RenderPass *p = nullptr;
    return p;
}

class RenderPass {
public:
    virtual ~RenderPass() {
        if(current_pass() == this) current_pass() = nullptr;
    }
    void use() { current_pass() = this; }  // RenderPass.cpp:107-109 (glUseProgram)
    // a buffer bound to `index` while this pass is current (SSBOBuffer::bind_current_shader)
    virtual void bind_buffer(int /*index*/, pt_buf * /*buf*/) {}
    virtual void bind_texture(const char * /*name*/, pt_tex * /*tex*/, int /*targetId*/, int /*type*/ = 0) {}
    // SCREEN_W / SCREEN_H uniforms of the post passes (main.cpp:262-263 ...): sizes come from the buffers
    virtual void uniform1i(const char *, int) {}
};

template <class T>
class SSBOBuffer {
    uint m_size;
    pt_buf *m_buf;   // was: GLuint m_ssbo
    bool m_borrowed; // outputs owned by a pass: release() leaves them alone
public:
    SSBOBuffer() : m_size(0), m_buf(nullptr), m_borrowed(false) {}
    SSBOBuffer(uint siz) : m_size(siz), m_buf(nullptr), m_borrowed(false) {
        check(pt_buf_create(Device::ctx(), (size_t)siz * sizeof(T), nullptr, &m_buf));
    }
    SSBOBuffer(uint siz, T *data) : m_size(siz), m_buf(nullptr), m_borrowed(false) {
        check(pt_buf_create(Device::ctx(), (size_t)siz * sizeof(T), siz ? data : nullptr, &m_buf));
    }
    static SSBOBuffer borrow(pt_buf *b) {
        SSBOBuffer s;
        s.m_buf = b;
        s.m_size = (uint)(pt_buf_bytes(b) / sizeof(T));
        s.m_borrowed = true;
        return s;
    }

    void release() {
        if(!isempty() && !m_borrowed) pt_buf_release(m_buf);
        m_buf = nullptr;
    }
    bool isempty() const { return m_buf == nullptr; }
    void bind_current_shader(int index) const {
        if(current_pass()) current_pass()->bind_buffer(index, m_buf);
    }
    void copy(const SSBOBuffer<T> *buffer) {
        assert(m_size == buffer->m_size);
        check(pt_buf_copy(m_buf, buffer->m_buf));
    }
    void copy(const T *data) {
        if(isempty()) return;
        check(pt_buf_upload(m_buf, data, (size_t)m_size * sizeof(T)));
    }
    void fill(const T &val) {
        if(isempty()) return;
        check(pt_buf_fill(m_buf, &val, sizeof(T)));
    }
    void save_as_image(int width, int height, int channel, const string &path) const {
        assert((uint)(width * height * channel) == m_size);
        check(pt_buf_save_as_image(m_buf, width, height, channel, path.c_str()));
    }

    // not in the reference: what glMapBufferRange gave its debugging code
    std::vector<T> download() const {
        std::vector<T> v(m_size);
        if(m_size) check(pt_buf_download(m_buf, v.data(), (size_t)m_size * sizeof(T)));
        return v;
    }
    uint size() const { return m_size; }
    pt_buf *handle() const { return m_buf; }
};

// Texture object names: Texture::load_to_gpu (src/texture/Texture.cpp:7-36) and
// Skybox::load_to_gpu (src/texture/Skybox.cpp:55-84) create these instead of GL textures.
inline pt_tex *create_texture_u8(int w, int h, int channels, const unsigned char *data) {
    pt_tex *t = nullptr;
    check(pt_tex_create(Device::ctx(), w, h, channels, PT_TEX_U8, data, &t));
    return t;
}
inline pt_tex *create_texture_f32(int w, int h, int channels, const float *data) {
    pt_tex *t = nullptr;
    check(pt_tex_create(Device::ctx(), w, h, channels, PT_TEX_F32, data, &t));
    return t;
}

struct TriangleRecord { float ver[9], normal[9], uv[6]; };                                   // pathtracing.glsl:32-36
struct InstanceInfo { float world2local[16]; int materialPtr; int meshIndex; int pad[10]; }; // ResourceManager.h
struct BVHNodeInfo { float aa[4], bb[4]; int ls, rs, instanceIndex, triangleIndex; };
struct LightInfo { int type; float x, y, z, r, g, b, pad; };

class ResourceManager {
public:
    static ResourceManager *&manager_slot() {
        static ResourceManager *m = nullptr;
        return m;
    }
#define RTAMD_MANAGER (::rtamd::ResourceManager::manager_slot())

    SSBOBuffer<uint64_t> textureHandleSSBO;
    SSBOBuffer<float> materialSSBO;
    SSBOBuffer<TriangleRecord> triangleSSBO;
    SSBOBuffer<InstanceInfo> instanceInfoSSBO;
    SSBOBuffer<BVHNodeInfo> meshBVHSSBO;
    SSBOBuffer<BVHNodeInfo> sceneBVHSSBO;
    SSBOBuffer<LightInfo> lightSSBO;
    int lightCount = 0;
    int getLightCount() const { return lightCount; }

    ~ResourceManager() {  // ResourceManager.cpp:231-239
        textureHandleSSBO.release(); materialSSBO.release(); triangleSSBO.release(); instanceInfoSSBO.release();
        meshBVHSSBO.release(); sceneBVHSSBO.release(); lightSSBO.release();
    }
};

class Renderer : public RenderPass {
    pt_pass *m_pass = nullptr;
    pt_params m_params;
    int m_w, m_h;
    bool m_batching = false;
    std::vector<pt_params> m_batch;
    template <class V>
    static bool is(const char *a, const char *b, V) { return !strcmp(a, b); }
public:
    // Outputs (Renderer.h:19-25)
    SSBOBuffer<float> directLumGBufferSSBO, indirectLumGBufferSSBO, motionGBufferSSBO, albedoGBufferSSBO,
        depthGBufferSSBO, normalGBufferSSBO, instanceIndexGBufferSSBO;

    // shaderPath is ignored: the pass is compiled into the library
    Renderer(const string & /*shaderPath*/) : m_w(screen_w()), m_h(screen_h()) {
        memset(&m_params, 0, sizeof m_params);
        m_params.SCREEN_W = m_w;
        m_params.SCREEN_H = m_h;
        check(pt_pass_create(Device::ctx(), m_w, m_h, &m_pass));
        auto out = [&](int which) {
            pt_buf *b = nullptr;
            check(pt_pass_output(m_pass, which, &b));
            return SSBOBuffer<float>::borrow(b);
        };
        directLumGBufferSSBO = out(PT_OUT_DIRECT_LUM);
        indirectLumGBufferSSBO = out(PT_OUT_INDIRECT_LUM);
        motionGBufferSSBO = out(PT_OUT_MOTION);
        albedoGBufferSSBO = out(PT_OUT_ALBEDO);
        depthGBufferSSBO = out(PT_OUT_DEPTH);
        normalGBufferSSBO = out(PT_OUT_NORMAL);
        instanceIndexGBufferSSBO = out(PT_OUT_INSTANCE_INDEX);
    }
    ~Renderer() override { pt_pass_destroy(m_pass); }

    pt_pass *handle() const { return m_pass; }
    pt_params &params() { return m_params; }

    void bind_buffer(int index, pt_buf *buf) override {
        if(index >= 0 && index <= 6) check(pt_pass_bind(m_pass, index, buf));
        // 7..15 are the pass's own outputs (Renderer.cpp:42-50): nothing to bind
    }
    void bind_texture(const char *name, pt_tex *tex, int /*targetId*/, int /*type*/ = 0) override {
        // "skybox", "skybox_samplecache" and the four raster images (main.cpp:179-184)
        check(pt_pass_bind_texture(m_pass, name, tex));
    }

    // the uniform block of main.cpp:185-205, by the shader's own names
    void uniform1i(const char *n, int v) override {
        if(!strcmp(n, "SCREEN_W")) { assert(v == m_w); }
        else if(!strcmp(n, "SCREEN_H")) { assert(v == m_h); }
        else if(!strcmp(n, "WINDOW_W")) m_params.WINDOW_W = v;
        else if(!strcmp(n, "WINDOW_H")) m_params.WINDOW_H = v;
        else if(!strcmp(n, "MAX_DEPTH")) m_params.MAX_DEPTH = v;
        else if(!strcmp(n, "SKY_W")) m_params.SKY_W = v;
        else if(!strcmp(n, "SKY_H")) m_params.SKY_H = v;
        else if(!strcmp(n, "BRDFSampling")) m_params.BRDFSampling = v;
        else if(!strcmp(n, "SkyboxSampling")) m_params.SkyboxSampling = v;
        else if(!strcmp(n, "SkyboxLighting")) m_params.SkyboxLighting = v;
        else if(!strcmp(n, "RasterizaionFor1st")) m_params.RasterizaionFor1st = v;
        else if(!strcmp(n, "lightCount")) m_params.lightCount = v;
        else throw std::runtime_error(string("Renderer::uniform1i: unknown uniform ") + n);
    }
    void uniform1ui(const char *n, unsigned v) {
        if(!strcmp(n, "currentspp")) m_params.currentspp = v;
        else if(!strcmp(n, "frameCounter")) m_params.frameCounter = v;
        else throw std::runtime_error(string("Renderer::uniform1ui: unknown uniform ") + n);
    }
    void uniform1f(const char *n, float v) {
        if(!strcmp(n, "skybox_Light_SUM")) m_params.skybox_Light_SUM = v;
        else if(!strcmp(n, "cameraNear")) m_params.cameraNear = v;
        else if(!strcmp(n, "cameraFar")) m_params.cameraFar = v;
        else if(!strcmp(n, "fov")) m_params.fov = v;
        else throw std::runtime_error(string("Renderer::uniform1f: unknown uniform ") + n);
    }
    void uniform2f(const char *n, float x, float y) {
        if(!strcmp(n, "jitter")) { m_params.jitter[0] = x; m_params.jitter[1] = y; }
        else throw std::runtime_error(string("Renderer::uniform2f: unknown uniform ") + n);
    }
    void uniformMatrix4fv(const char *n, const float *column_major) {  // glm::value_ptr order
        if(!strcmp(n, "v2wMat")) memcpy(m_params.v2wMat, column_major, 64);
        else if(!strcmp(n, "backprojMat")) memcpy(m_params.backprojMat, column_major, 64);
        else throw std::runtime_error(string("Renderer::uniformMatrix4fv: unknown uniform ") + n);
    }

    void draw() {  // Renderer.cpp:30-53
        ResourceManager *rm = RTAMD_MANAGER;
        RenderPass *prev = current_pass();
        use();
        rm->textureHandleSSBO.bind_current_shader(0);
        rm->materialSSBO.bind_current_shader(1);
        rm->triangleSSBO.bind_current_shader(2);
        rm->instanceInfoSSBO.bind_current_shader(3);
        rm->meshBVHSSBO.bind_current_shader(4);
        rm->sceneBVHSSBO.bind_current_shader(5);
        rm->lightSSBO.bind_current_shader(6);
        uniform1i("lightCount", rm->getLightCount());
        current_pass() = prev ? prev : this;
        if(m_batching) {  // (between begin_batch and end_batch a draw only records its sample's uniform block)
            m_batch.push_back(m_params);
            return;
        }
        check(pt_pass_set_params(m_pass, &m_params));
        check(pt_pass_draw(m_pass));  // was: drawcall()
    }

    // No counterpart in the reference.  Its frame is `for(spp = 1 .. Config::SPP) { uniforms; renderPass->draw(); }`
    // (main.cpp:169-208); with begin_batch() before that loop and end_batch() behind it the loop stays as it is, and the
    // samples it drew are handed over together, sixteen at a time (pt_pass_draw_batch: one set of launches per call; same
    // outputs).  The samples of a batch may differ in jitter, currentspp and frameCounter only.
    void begin_batch() {
        m_batching = true;
        m_batch.clear();
    }
    void end_batch() {
        m_batching = false;
        for(size_t k = 0; k < m_batch.size(); k += 16)
            check(pt_pass_draw_batch(m_pass, m_batch.data() + k, (int)std::min<size_t>(16, m_batch.size() - k)));
        m_batch.clear();
    }
};

class SVGFMergePass : public RenderPass {
public:
    SSBOBuffer<float> colorGBufferSSBO;  // Output
    SVGFMergePass(const string &) : colorGBufferSSBO((uint)(screen_w() * screen_h() * 3)) {}
    ~SVGFMergePass() override { colorGBufferSSBO.release(); }
    void draw(const SSBOBuffer<float> &directLumGBufferSSBO, const SSBOBuffer<float> &indirectLumGBufferSSBO,
              const SSBOBuffer<float> &albedoGBufferSSBO) {
        check(pt_post_merge(Device::ctx(), directLumGBufferSSBO.handle(), indirectLumGBufferSSBO.handle(),
                            albedoGBufferSSBO.handle(), colorGBufferSSBO.handle()));
    }
};

class StaticBlender : public RenderPass {
public:
    SSBOBuffer<float> historyColorGBufferSSBO, historyMomentGBufferSSBO;  // Outputs
    int frameCounter;
    StaticBlender(const string &)
        : historyColorGBufferSSBO((uint)(screen_w() * screen_h() * 3)), historyMomentGBufferSSBO((uint)(screen_w() * screen_h() * 2)),
          frameCounter(0) {}
    ~StaticBlender() override {
        historyColorGBufferSSBO.release();
        historyMomentGBufferSSBO.release();  // the reference leaks this one (StaticBlender.cpp:29-31)
    }
    void draw(const SSBOBuffer<float> &colorGBufferSSBO) {  // StaticBlender.cpp:17-27
        ++frameCounter;
        check(pt_post_static_blend(Device::ctx(), (uint32_t)frameCounter, colorGBufferSSBO.handle(), historyColorGBufferSSBO.handle(),
                                   historyMomentGBufferSSBO.handle()));
    }
};

class ToneMappingGamma : public RenderPass {
public:
    SSBOBuffer<float> outputColorGBufferSSBO;
    ToneMappingGamma(const string &) : outputColorGBufferSSBO((uint)(screen_w() * screen_h() * 3)) {}
    ~ToneMappingGamma() override { outputColorGBufferSSBO.release(); }
    void draw(const SSBOBuffer<float> &colorGBufferSSBO) {
        check(pt_post_tonemap(Device::ctx(), colorGBufferSSBO.handle(), outputColorGBufferSSBO.handle()));
    }
};

// Config::SVGFSpatialFilterLevel (src/Config.h:63): a-trous levels SVGFSpatialFilterPass::draw runs
inline int &svgf_spatial_filter_level() { static int level = 0; return level; }

class SVGFTemporalFilter : public RenderPass {  // src/renderpass/SVGFTemporalFilter.{h,cpp}
public:
    bool firstFrame = true;
    SSBOBuffer<float> outputColorGBufferSSBO, outputMomentGBufferSSBO, outputNumSamplesGBufferSSBO;
    SSBOBuffer<float> historyColorGBufferSSBO, historyMomentGBufferSSBO, historyNumSamplesGBufferSSBO,
        historynormalGBufferSSBO, historyinstanceIndexGBufferSSBO;
    SVGFTemporalFilter(const string &)
        : outputColorGBufferSSBO((uint)(screen_w() * screen_h() * 3)), outputMomentGBufferSSBO((uint)(screen_w() * screen_h() * 2)),
          outputNumSamplesGBufferSSBO((uint)(screen_w() * screen_h())), historyColorGBufferSSBO((uint)(screen_w() * screen_h() * 3)),
          historyMomentGBufferSSBO((uint)(screen_w() * screen_h() * 2)), historyNumSamplesGBufferSSBO((uint)(screen_w() * screen_h())),
          historynormalGBufferSSBO((uint)(screen_w() * screen_h() * 3)), historyinstanceIndexGBufferSSBO((uint)(screen_w() * screen_h())) {}
    ~SVGFTemporalFilter() override {
        for(SSBOBuffer<float> *b : {&outputColorGBufferSSBO, &outputMomentGBufferSSBO, &outputNumSamplesGBufferSSBO, &historyColorGBufferSSBO,
                                    &historyMomentGBufferSSBO, &historyNumSamplesGBufferSSBO, &historynormalGBufferSSBO,
                                    &historyinstanceIndexGBufferSSBO})
            b->release();
    }
    void draw(const SSBOBuffer<float> &colorGBufferSSBO, const SSBOBuffer<float> &normalGBufferSSBO,
              const SSBOBuffer<float> &instanceIndexGBufferSSBO, const SSBOBuffer<float> &motionGBufferSSBO) {
        // as SVGFTemporalFilter.cpp:30-31: the flag is cleared BEFORE it is handed to the shader, so the
        // shader's first-frame branch never runs and the first frame blends with whatever the history holds
        if(firstFrame) firstFrame = false;
        check(pt_post_svgf_temporal(Device::ctx(), screen_w(), screen_h(), firstFrame, colorGBufferSSBO.handle(),
                                    normalGBufferSSBO.handle(), instanceIndexGBufferSSBO.handle(), motionGBufferSSBO.handle(),
                                    outputColorGBufferSSBO.handle(), outputMomentGBufferSSBO.handle(), outputNumSamplesGBufferSSBO.handle(),
                                    historyColorGBufferSSBO.handle(), historyMomentGBufferSSBO.handle(),
                                    historyNumSamplesGBufferSSBO.handle(), historynormalGBufferSSBO.handle(),
                                    historyinstanceIndexGBufferSSBO.handle()));
        historyColorGBufferSSBO.copy(&outputColorGBufferSSBO);
        historyMomentGBufferSSBO.copy(&outputMomentGBufferSSBO);
        historyNumSamplesGBufferSSBO.copy(&outputNumSamplesGBufferSSBO);
        historynormalGBufferSSBO.copy(&normalGBufferSSBO);
        historyinstanceIndexGBufferSSBO.copy(&instanceIndexGBufferSSBO);
    }
    void update_historycolor(const SSBOBuffer<float> &colorGBufferSSBO) { historyColorGBufferSSBO.copy(&colorGBufferSSBO); }
};

class SVGFVarianceFilter : public RenderPass {  // src/renderpass/SVGFVarianceFilter.{h,cpp}
public:
    SSBOBuffer<float> varianceGBufferSSBO;
    SVGFVarianceFilter(const string &) : varianceGBufferSSBO((uint)(screen_w() * screen_h())) {}
    ~SVGFVarianceFilter() override { varianceGBufferSSBO.release(); }
    void draw(const SSBOBuffer<float> &momentGBufferSSBO, const SSBOBuffer<float> &normalGBufferSSBO,
              const SSBOBuffer<float> &depthGBufferSSBO, const SSBOBuffer<float> &numSamplesGBufferSSBO) {
        check(pt_post_svgf_variance(Device::ctx(), screen_w(), screen_h(), momentGBufferSSBO.handle(), normalGBufferSSBO.handle(),
                                    depthGBufferSSBO.handle(), numSamplesGBufferSSBO.handle(), varianceGBufferSSBO.handle()));
    }
};

class SVGFSpatialFilterPass : public RenderPass {  // src/renderpass/SVGFSpatialFilterPass.{h,cpp}
public:
    SSBOBuffer<float> outputColorGBufferSSBO, tmpColorGBufferSSBO;
    SVGFSpatialFilterPass(const string &)
        : outputColorGBufferSSBO((uint)(screen_w() * screen_h() * 3)), tmpColorGBufferSSBO((uint)(screen_w() * screen_h() * 3)) {}
    ~SVGFSpatialFilterPass() override {
        outputColorGBufferSSBO.release();
        tmpColorGBufferSSBO.release();
    }
    void draw(const SSBOBuffer<float> &colorGBufferSSBO, const SSBOBuffer<float> &varianceGBufferSSBO,
              const SSBOBuffer<float> &normalGBufferSSBO, const SSBOBuffer<float> &depthGBufferSSBO) {
        outputColorGBufferSSBO.copy(&colorGBufferSSBO);
        for(int i = 0; i < svgf_spatial_filter_level(); i++) {
            check(pt_post_svgf_spatial(Device::ctx(), screen_w(), screen_h(), 1 << i, outputColorGBufferSSBO.handle(),
                                       normalGBufferSSBO.handle(), depthGBufferSSBO.handle(), varianceGBufferSSBO.handle(),
                                       tmpColorGBufferSSBO.handle()));
            std::swap(tmpColorGBufferSSBO, outputColorGBufferSSBO);
        }
    }
};

class TAA : public RenderPass {  // src/renderpass/TAA.{h,cpp}
public:
    bool firstFrame;
    SSBOBuffer<float> historycolorGBufferSSBO, historynormalGBufferSSBO, historyinstanceIndexGBufferSSBO, outputColorGBufferSSBO;
    TAA(const string &)
        : firstFrame(true), historycolorGBufferSSBO((uint)(screen_w() * screen_h() * 3)),
          historynormalGBufferSSBO((uint)(screen_w() * screen_h() * 3)), historyinstanceIndexGBufferSSBO((uint)(screen_w() * screen_h())),
          outputColorGBufferSSBO((uint)(screen_w() * screen_h() * 3)) {}
    ~TAA() override {
        historycolorGBufferSSBO.release(); historynormalGBufferSSBO.release();
        historyinstanceIndexGBufferSSBO.release(); outputColorGBufferSSBO.release();
    }
    void draw(const SSBOBuffer<float> &colorGBufferSSBO, const SSBOBuffer<float> &motionGBufferSSBO,
              const SSBOBuffer<float> &normalGBufferSSBO, const SSBOBuffer<float> &instanceIndexGBufferSSBO) {
        if(firstFrame) {  // TAA.cpp:19-25: the first frame only fills the history (outputColorGBufferSSBO keeps its content)
            firstFrame = false;
            historycolorGBufferSSBO.copy(&colorGBufferSSBO);
            historynormalGBufferSSBO.copy(&normalGBufferSSBO);
            historyinstanceIndexGBufferSSBO.copy(&instanceIndexGBufferSSBO);
            return;
        }
        check(pt_post_taa(Device::ctx(), screen_w(), screen_h(), colorGBufferSSBO.handle(), historycolorGBufferSSBO.handle(),
                          motionGBufferSSBO.handle(), normalGBufferSSBO.handle(), instanceIndexGBufferSSBO.handle(),
                          historynormalGBufferSSBO.handle(), historyinstanceIndexGBufferSSBO.handle(), outputColorGBufferSSBO.handle()));
        historycolorGBufferSSBO.copy(&outputColorGBufferSSBO);
        historynormalGBufferSSBO.copy(&normalGBufferSSBO);
        historyinstanceIndexGBufferSSBO.copy(&instanceIndexGBufferSSBO);
    }
};

}  // namespace rtamd
