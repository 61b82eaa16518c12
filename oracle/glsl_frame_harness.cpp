// TEST INFRASTRUCTURE, build container only: runs the reference's WHOLE fragment shader,
// shader/pathtracing.glsl with everything it includes, as a C++ program -- one call of its main()
// per pixel -- against the glm headers the reference vendors.  oracle/glsl_frame.py prepares the
// translation unit (oracle/_ref/glsl_pathtracing.inc, git-ignored; the shader text is read where it
// lies under /root/reference and is never copied into the repository), feeds this program the
// scene buffers and uniforms, and stores the seven output buffers as the fixture
// tests/golden/glsl_frame_casa.npz, which the oracle is then checked against.
//
// What the script changes in the text, all mechanical (the list in oracle/glsl_frame.py is the
// authority): `#include shader/...` expanded as the reference's read_shader does; `#version` /
// `#extension` lines dropped; `layout(...) buffer X { T name[]; };` -> `static T *name;`;
// `uniform T x;` / `in vec2 screen_uv;` -> statics; parameter qualifiers in/out/inout -> C++ value /
// reference.  Every statement and expression of every function is the reference's (aggregate
// constructors such as Ray(a, b) compile as they stand under -std=gnu++20).
//
// What this file supplies is what the GL driver supplies: buffers, uniforms, sampler2D and
// texture() (GL 4.6 section 8.14: REPEAT wrap, LINEAR or NEAREST filter, base level, fp32),
// and the implicit int -> float / uint -> float conversions GLSL performs.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define GLM_FORCE_SWIZZLE
#define GLM_FORCE_INTRINSICS  // (turns on the language-extension flag that glm wants for swizzle members like v.xyz; vectors stay packed, arithmetic stays scalar)
#include <glm/glm.hpp>
#include "oracle_math.h"

namespace ref_glsl {
using namespace glm;

// ---- sampler2D / texture() -------------------------------------------------------------------
struct sampler2D {
    const void *data = nullptr;
    int w = 0, h = 0, c = 0;
    bool f32 = false, nearest = false;
};
struct Texel {  // what texture() returns: the swizzles the shader uses, as plain members
    float r;
    vec2 xy;
    vec3 xyz, rgb;
};
static inline vec4 fetch(const sampler2D &s, int x, int y) {
    vec4 t(0.0f, 0.0f, 0.0f, 1.0f);
    const size_t base = ((size_t)y * s.w + x) * s.c;
    for(int k = 0; k < s.c && k < 4; k++)
        t[k] = s.f32 ? static_cast<const float *>(s.data)[base + k] : static_cast<const uint8_t *>(s.data)[base + k] / 255.0f;
    return t;
}
static inline int wrap(int i, int n) {
    int m = i % n;
    return m < 0 ? m + n : m;
}
static inline Texel texture(const sampler2D &s, vec2 uv) {
    if(!s.data) {
        fprintf(stderr, "texture() on an unbound sampler\n");
        abort();
    }
    vec4 t;
    const float u = uv.x * (float)s.w, v = uv.y * (float)s.h;
    if(s.nearest) {
        t = fetch(s, wrap((int)std::floor(u), s.w), wrap((int)std::floor(v), s.h));
    } else {
        const float ub = u - 0.5f, vb = v - 0.5f;
        const float fu = std::floor(ub), fv = std::floor(vb);
        const float a = ub - fu, b = vb - fv;
        const int i0 = wrap((int)fu, s.w), j0 = wrap((int)fv, s.h), i1 = wrap(i0 + 1, s.w), j1 = wrap(j0 + 1, s.h);
        const vec4 t00 = fetch(s, i0, j0), t10 = fetch(s, i1, j0), t01 = fetch(s, i0, j1), t11 = fetch(s, i1, j1);
        t = (1.0f - a) * (1.0f - b) * t00 + a * (1.0f - b) * t10 + (1.0f - a) * b * t01 + a * b * t11;
    }
    Texel r;
    r.r = t.x;
    r.xy = vec2(t.x, t.y);
    r.xyz = r.rgb = vec3(t.x, t.y, t.z);
    return r;
}

// ---- implicit conversions of GLSL that C++ / glm do not perform ---------------------------------
static inline float step(float edge, int x) { return glm::step(edge, (float)x); }
static inline float step(float edge, float x) { return glm::step(edge, x); }
static inline vec2 operator+(const uvec2 &a, const vec2 &b) { return vec2(a) + b; }
static inline vec2 operator/(const vec2 &a, int b) { return a / (float)b; }
template <int E0, int E1>
static inline vec2 operator+(const glm::detail::_swizzle<2, float, glm::packed_highp, E0, E1, -1, -2> &a, float b) { return a() + b; }  // v.xy + 1.0
static inline float max(float a, float b) { return glm::max(a, b); }
static inline vec3 max(const vec3 &a, const vec3 &b) { return glm::max(a, b); }
static inline float min(float a, float b) { return glm::min(a, b); }
static inline float max(float a, double b) { return glm::max(a, (float)b); }
static inline float max(double a, float b) { return glm::max((float)a, b); }

#ifdef GLSL_FIXED_TRANSCENDENTALS
// GLSL gives sin, cos, atan, asin and pow an error bound, not a value.  In this build the shader's
// calls resolve to the one recipe the oracle and the HIP path use (oracle_math.h) instead of the C
// library's: whatever still differs from the oracle afterwards is a difference of transcription.
static inline float sin(float x) { return orc_math::m_sin(x); }
static inline float cos(float x) { return orc_math::m_cos(x); }
static inline float atan(float y, float x) { return orc_math::m_atan2(y, x); }
static inline float asin(float x) { return orc_math::m_asin(x); }
static inline float pow(float x, float n) { return orc_math::m_pow(x, n); }
#endif

#include "_ref/glsl_pathtracing.inc"

}  // namespace ref_glsl

using namespace ref_glsl;

// ---- input blob (written by oracle/glsl_frame.py) -----------------------------------------------
static std::vector<char> g_blob;
static size_t g_pos = 0;
template <class T>
static T take() {
    T v;
    memcpy(&v, g_blob.data() + g_pos, sizeof v);
    g_pos += sizeof v;
    return v;
}
static const void *take_bytes(size_t n) {
    const void *p = g_blob.data() + g_pos;
    g_pos += (n + 15) & ~(size_t)15;
    return p;
}
static sampler2D take_tex() {
    sampler2D s;
    s.w = take<int>(); s.h = take<int>(); s.c = take<int>();
    s.f32 = take<int>() != 0;
    s.nearest = take<int>() != 0;
    g_pos = (g_pos + 15) & ~(size_t)15;
    s.data = take_bytes((size_t)s.w * s.h * s.c * (s.f32 ? 4 : 1));
    return s;
}

int main(int argc, char **argv) {
    if(argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if(!f) return 2;
    fseek(f, 0, SEEK_END);
    g_blob.resize((size_t)ftell(f));
    fseek(f, 0, SEEK_SET);
    if(fread(g_blob.data(), 1, g_blob.size(), f) != g_blob.size()) return 2;
    fclose(f);

    static_assert(sizeof(TriangleX) == 96 && sizeof(BVHNode) == 48 && sizeof(InstanceInfo) == 112 && sizeof(Light) == 32,
                  "the shader's structs must have the byte layouts of the buffers");
    // uniforms (main.cpp:185-205)
    SCREEN_W = take<int>(); SCREEN_H = take<int>(); WINDOW_W = take<int>(); WINDOW_H = take<int>();
    MAX_DEPTH = take<int>();
    fov = take<float>(); cameraNear = take<float>(); cameraFar = take<float>();
    skybox_Light_SUM = take<float>();
    SKY_W = take<int>(); SKY_H = take<int>();
    frameCounter = take<uint32_t>();
    const uint32_t n_spp = take<uint32_t>();
    BRDFSampling = take<int>() != 0; SkyboxSampling = take<int>() != 0; SkyboxLighting = take<int>() != 0;
    RasterizaionFor1st = take<int>() != 0;
    lightCount = take<int>();
    for(int k = 0; k < 16; k++) (&v2wMat[0][0])[k] = take<float>();
    for(int k = 0; k < 16; k++) (&backprojMat[0][0])[k] = take<float>();
    std::vector<vec2> jitters(n_spp);
    for(auto &j : jitters) { j.x = take<float>(); j.y = take<float>(); }
    g_pos = (g_pos + 15) & ~(size_t)15;
    // buffers
    const size_t n_mat = take<uint64_t>(), n_tri = take<uint64_t>(), n_inst = take<uint64_t>(), n_mesh = take<uint64_t>(),
                 n_scene = take<uint64_t>(), n_light = take<uint64_t>(), n_tex = take<uint64_t>();
    g_pos = (g_pos + 15) & ~(size_t)15;
    materialBuffer = (float *)take_bytes(n_mat * 4);
    triangleBuffer = (TriangleX *)take_bytes(n_tri * 96);
    instanceInfoBuffer = (InstanceInfo *)take_bytes(n_inst * 112);
    meshBVHBuffer = (BVHNode *)take_bytes(n_mesh * 48);
    sceneBVHBuffer = (BVHNode *)take_bytes(n_scene * 48);
    lightBuffer = (Light *)take_bytes(n_light * 32);
    std::vector<sampler2D> textures(n_tex ? n_tex : 1);
    for(size_t k = 0; k < n_tex; k++) textures[k] = take_tex();
    textureBuffer = textures.data();
    skybox = take_tex();
    skybox_samplecache = take_tex();
    // one set of raster images per sample (RasterPass is drawn per sample with that sample's jitter, main.cpp:174-175)
    std::vector<sampler2D> gbuf(4 * (size_t)n_spp);
    for(auto &g : gbuf) g = take_tex();

    const size_t npix = (size_t)SCREEN_W * SCREEN_H;
    std::vector<float> o_albedo(npix * 3), o_motion(npix * 2), o_depth(npix), o_normal(npix * 3), o_inst(npix), o_di(npix * 3), o_idi(npix * 3);
    albedoGBuffer = o_albedo.data(); motionGBuffer = o_motion.data(); depthGBuffer = o_depth.data(); normalGBuffer = o_normal.data();
    instanceIndexGBuffer = o_inst.data(); directLumGBuffer = o_di.data(); indirectLumGBuffer = o_idi.data();

    // optional row band [y_begin, y_end): a large frame is cut over several PROCESSES (the shader keeps state in globals --
    // its traversal stack, the sampler -- so threads would not do); rows outside the band stay zero in the output
    const int y_begin = argc > 4 ? atoi(argv[3]) : 0, y_end = argc > 4 ? atoi(argv[4]) : SCREEN_H;
    if(y_begin < 0 || y_end > SCREEN_H || y_begin > y_end) return 2;
    // the frame loop of main.cpp:169-208: one full-screen pass per spp, one fragment per pixel, fragment
    // coordinates at pixel centres (what the rasteriser gives a full-screen quad)
    for(uint32_t spp = 1; spp <= n_spp; spp++) {
        currentspp = spp;
        jitter = jitters[spp - 1];
        depthGBufferTexture = gbuf[4 * (spp - 1) + 0];
        normalGBufferTexture = gbuf[4 * (spp - 1) + 1];
        uvGBufferTexture = gbuf[4 * (spp - 1) + 2];
        instanceIndexGBufferTexture = gbuf[4 * (spp - 1) + 3];
        for(int y = y_begin; y < y_end; y++)
            for(int x = 0; x < SCREEN_W; x++) {
                screen_uv = vec2(((float)x + 0.5f) / (float)SCREEN_W, ((float)y + 0.5f) / (float)SCREEN_H);
                stack_h = 0;
                ref_glsl::main();
            }
    }
    f = fopen(argv[2], "wb");
    if(!f) return 2;
    for(auto *v : {&o_albedo, &o_motion, &o_depth, &o_normal, &o_inst, &o_di, &o_idi}) fwrite(v->data(), 4, v->size(), f);
    fclose(f);
    return 0;
}
