// TEST INFRASTRUCTURE ONLY -- see pt_oracle.h for scope, pinning status and the
// conventions chosen where GLSL leaves behaviour open.
//
// Scalar CPU restatement of the reference's path-tracing pass.  S/ = shader/ in the
// reference tree.  Every function names the GLSL lines it follows.
#include "pt_oracle.h"
#include "oracle_math.h"
#include <cmath>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------- S/basic/math.glsl:1-4
const float INF = 1E18f;
const float PI = 3.1415926f;
const float INV_PI = 0.3183098f;
const float EPS = 0.0001f;
const float RAY_MINDIS = 0.001f;  // S/pathtracing.glsl:10

struct vec2 { float x, y; };
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    explicit vec3(float a) : x(a), y(a), z(a) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
};
inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator/(vec3 a, vec3 b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline float sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
inline float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }
inline vec3 mixv(vec3 x, vec3 y, float a) { return x * (1.0f - a) + y * a; }
inline vec3 reflect(vec3 I, vec3 N) { return I - (2.0f * dot(N, I)) * N; }
inline vec3 vmax(vec3 a, vec3 b) { return vec3(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)); }
inline bool any_nan(vec3 v) { return std::isnan(v.x) || std::isnan(v.y) || std::isnan(v.z); }


// sin / cos / atan / asin / pow: the fixed recipes of oracle_math.h (m_sin, m_cos, m_atan2, m_asin, m_pow)
using namespace orc_math;

// S/basic/math.glsl:6-22
inline int roundint(float x) {
    float b = (0.0f < x) ? 0.0f : 1.0f;  // step(x, 0)
    return (int)(x - b + 0.5f);
}
inline float pow2(float x) { return x * x; }
inline float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
inline vec3 powv(vec3 v, float n) { return vec3(m_pow(v.x, n), m_pow(v.y, n), m_pow(v.z, n)); }
// S/basic/math.glsl:25-32 (shadows the builtin)
inline vec3 refract_custom(vec3 I, vec3 N, float eta) {
    float c1 = dot(N, -I);
    float s1 = std::sqrt(1.0f - c1 * c1);
    float s2 = s1 / eta;
    if(s2 >= 1.0f) return vec3(0.0f);
    float c2 = std::sqrt(1.0f - s2 * s2);
    return -N * c2 + (I + N * c1) / eta;
}

// ---------------------------------------------------------------- S/basic/sobol*.glsl
const uint32_t V[640] = {
#include "sobol_table.inc"
};

inline uint32_t wang_hash(uint32_t seed) {  // S/basic/sobol.glsl:21-28
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}

// S/basic/sobol.glsl:4-14.  inv32 = 1.0 / float(0xFFFFFFFFU) = 2^-32 exactly (the
// conversion rounds 2^32 - 1 up to 2^32); float(r) rounds to nearest even, so the result
// can be exactly 1.0.
inline float sobol(uint32_t d, uint32_t i) {
    if(d >= 20u) {  // outside the reference's table: documented extension (pt_oracle.h)
        uint32_t fold = (d - 1u) / 19u;
        d = 1u + (d - 1u) % 19u;
        i = wang_hash(i + 0x9E3779B9u * fold);
    }
    const float inv32 = 1.0f / (float)0xFFFFFFFFU;
    uint32_t r = 0;
    uint32_t offset = d * 32u;
    for(uint32_t j = 0; i != 0; i >>= 1, j++)
        if((i & 1u) != 0) r ^= V[offset + j];
    return (float)r * inv32;
}

// ---------------------------------------------------------------- scene records
struct Node { float aa[4], bb[4]; int32_t ls, rs, inst, tri; };
struct Instance { float w2l[16]; int32_t materialPtr, meshIndex, pad[10]; };
struct Light { int32_t type; float x, y, z, r, g, b, pad; };

struct Ray { vec3 ori, dir; };
struct Intersection {  // S/pathtracing.glsl:196-205
    bool exist = false;
    vec3 position = vec3(0.0f);
    vec3 normal = vec3(0, 0, 1);
    float t = INF;
    vec2 uv = {0, 0};
    int instanceIndex = -1;
    int triangleIndex = -1;
    float bu = 0, bv = 0;  // barycentrics (not in the reference struct; reported to tests)
};

struct BSDFQueryRecord {  // S/materials/materials.glsl:62-71
    vec3 wi, wo;
    int mptr;
    vec2 uv;
    float eta;
};

struct Frame { vec3 s, t, n; };

// Per-invocation state: what the shader keeps in globals (sobolseed, sobolcurdim, the
// traversal stack) plus the counters.
struct Ctx {
    const orc_scene *sc;
    const orc_params *pm;
    uint32_t sobolseed = 0, sobolcurdim = 0;
    int stack_h = 0;
    int stack[256];
    bool counting_extend = true;  // false while a shadow ray is traced
    orc_counters cnt;
    Ctx() { memset(&cnt, 0, sizeof cnt); }

    float rand() { return sobol(sobolcurdim++, sobolseed); }  // S/basic/sobol.glsl:37-40

    const Node *mesh_nodes() const { return (const Node *)sc->mesh_nodes; }
    const Node *scene_nodes() const { return (const Node *)sc->scene_nodes; }
    const Instance *instances() const { return (const Instance *)sc->instances; }
    const Light *lights() const { return (const Light *)sc->lights; }
    float mat(int i) const { return sc->materials[i]; }

    // ------------------------------------------------------------ texture()
    // GL 4.6 core 8.14.2/8.14.3: LINEAR filter, REPEAT wrap, one level.
    void texture(const orc_texture &tx, float s, float t, float out[4]) {
        float u = s * (float)tx.width - 0.5f;
        float v = t * (float)tx.height - 0.5f;
        float fu = std::floor(u), fv = std::floor(v);
        float a = u - fu, b = v - fv;
        auto wrap = [](float f, int n) {
            // REPEAT: texel index modulo size, for negative coordinates too; coordinates
            // beyond +-2^30 texels are clamped first so the integer conversion is defined
            float c = std::fmin(std::fmax(f, -1073741824.0f), 1073741824.0f);
            int m = (int)c % n;
            return m < 0 ? m + n : m;
        };
        int i0 = wrap(fu, tx.width), j0 = wrap(fv, tx.height);
        int i1 = i0 + 1 == tx.width ? 0 : i0 + 1;
        int j1 = j0 + 1 == tx.height ? 0 : j0 + 1;
        float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
        float texel[4][4];
        const int xs[4] = {i0, i1, i0, i1}, ys[4] = {j0, j0, j1, j1};
        for(int k = 0; k < 4; k++) {
            size_t base = ((size_t)ys[k] * tx.width + xs[k]) * tx.channels;
            for(int c = 0; c < 4; c++) {
                float val = c == 3 ? 1.0f : 0.0f;
                if(c < tx.channels) {
                    if(tx.format == 0) val = (float)((const uint8_t *)tx.data)[base + c] / 255.0f;
                    else val = ((const float *)tx.data)[base + c];
                }
                texel[k][c] = val;
            }
        }
        for(int c = 0; c < 4; c++)
            out[c] = ((w00 * texel[0][c] + w10 * texel[1][c]) + w01 * texel[2][c]) + w11 * texel[3][c];
        if(tx.format == 0) cnt.ldr_tex_fetches++;
        else cnt.sky_fetches++;
    }
    // NEAREST filter, REPEAT wrap (GL 4.6 core 8.14.2): red channel of texel floor(s * w), floor(t * h)
    float texture_nearest(const orc_texture &tx, float s, float t) {
        auto wrap = [](float f, int n) {
            float c = std::fmin(std::fmax(f, -1073741824.0f), 1073741824.0f);
            int m = (int)c % n;
            return m < 0 ? m + n : m;
        };
        int i = wrap(std::floor(s * (float)tx.width), tx.width), j = wrap(std::floor(t * (float)tx.height), tx.height);
        size_t base = ((size_t)j * tx.width + i) * tx.channels;
        return tx.format == 0 ? (float)((const uint8_t *)tx.data)[base] / 255.0f : ((const float *)tx.data)[base];
    }
    // int(x) of GLSL is undefined outside the int range and for NaN: saturate, NaN -> 0
    static int float_to_int(float x) {
        if(x != x) return 0;
        if(x >= 2147483520.0f) return 2147483647;
        if(x <= -2147483648.0f) return (int)0x80000000;
        return (int)x;
    }

    vec3 albedo_texture(int albedo_map, vec2 uv) {
        float c[4];
        texture(sc->textures[albedo_map], uv.x, uv.y, c);
        return powv(vec3(c[0], c[1], c[2]), 2.2f);
    }

    // ------------------------------------------------------------ S/materials/materials.glsl:5-31
    vec3 sample_GGX(float alpha, float &pdf) {
        float alpha2 = alpha * alpha;
        float x = rand();
        float cos_theta = std::sqrt((1.0f - x) / (x * (alpha2 - 1.0f) + 1.0f));
        float cos2 = cos_theta * cos_theta;
        float phi = rand() * 2.0f * PI;
        float r = std::sqrt(1.0f - cos2);
        pdf = 2.0f * alpha2 * cos_theta / pow2(cos2 * (alpha2 - 1.0f) + 1.0f) / (2.0f * PI);
        return normalize(vec3(r * m_cos(phi), r * m_sin(phi), cos_theta));
    }
    static float eval_GGX(float alpha, vec3 h) {
        float cos_theta = std::fabs(h.z);
        float alpha2 = alpha * alpha;
        float cos2 = cos_theta * cos_theta;
        return alpha2 * cos_theta / pow2(cos2 * (alpha2 - 1.0f) + 1.0f) / PI;
    }
    static float G_GGX(float alpha, float cosI) {
        float k = pow2(alpha + 1.0f) / 8.0f;
        return cosI / (cosI * (1.0f - k) + k);
    }
    // S/materials/materials.glsl:43-59
    static vec3 SchlickFresnel(vec3 f0, float c) { return f0 + (vec3(1.0f) - f0) * pow5(1.0f - c); }
    static float fresnel(float cosI, float eta) {
        cosI = std::fabs(cosI);
        float sinT = 1.0f / eta * std::sqrt(1.0f - cosI * cosI);
        if(sinT > 1.0f) return 1.0f;
        float cosT = std::sqrt(1.0f - sinT * sinT);
        float Rs = pow2((cosI - eta * cosT) / (cosI + eta * cosT));
        float Rp = pow2((cosT - eta * cosI) / (cosT + eta * cosI));
        return (Rs + Rp) / 2.0f;
    }

    vec3 base_albedo(const BSDFQueryRecord &bRec) {
        vec3 albedo(mat(bRec.mptr + 1), mat(bRec.mptr + 2), mat(bRec.mptr + 3));
        int albedo_map = roundint(mat(bRec.mptr + 6));
        if(albedo_map >= 0) albedo = albedo_texture(albedo_map, bRec.uv);
        return albedo;
    }

    // ------------------------------------------------------------ S/materials/RoughConductor.glsl
    vec3 eval_RoughConductor(const BSDFQueryRecord &bRec) {  // :3-32
        if(sign(bRec.wi.z) != sign(bRec.wo.z)) return vec3(0.0f);
        vec3 albedo(mat(bRec.mptr + 1), mat(bRec.mptr + 2), mat(bRec.mptr + 3));
        float alpha = mat(bRec.mptr + 4);
        float metallic = mat(bRec.mptr + 5);
        int albedo_map = roundint(mat(bRec.mptr + 6));
        if(albedo_map >= 0) albedo = albedo_texture(albedo_map, bRec.uv);

        vec3 F0 = mixv(vec3(0.04f), albedo, metallic);
        vec3 H = normalize(bRec.wi + bRec.wo);
        H = H * sign(H.z);
        float OdotH = std::fabs(dot(bRec.wo, H));
        float OdotN = std::fabs(bRec.wo.z), IdotN = std::fabs(bRec.wi.z);
        float D = eval_GGX(alpha, H);
        vec3 F = SchlickFresnel(F0, OdotH);

        float k = pow2(alpha + 1.0f) / 8.0f;
        float Gi4io = 0.25f / ((IdotN * (1.0f - k) + k) * (OdotN * (1.0f - k) + k));

        float FL = pow5(1.0f - IdotN), FV = pow5(1.0f - OdotN);
        float Fd90 = 0.5f + 2.0f * pow2(OdotH) * alpha;
        float Fd = mixf(1.0f, Fd90, FL) * mixf(1.0f, Fd90, FV);

        return D * F * Gi4io + (1.0f - metallic) * albedo * Fd * INV_PI;
    }
    float pdf_RoughConductor(const BSDFQueryRecord &bRec) {  // :34-43
        if(sign(bRec.wi.z) != sign(bRec.wo.z)) return 0.0f;
        float alpha = mat(bRec.mptr + 4);
        vec3 H = normalize(bRec.wi + bRec.wo);
        H = H * sign(H.z);
        float OdotH = std::fabs(dot(bRec.wo, H));
        float pdf = eval_GGX(alpha, H);
        return pdf / (4.0f * OdotH);
    }
    vec3 sample_RoughConductor(BSDFQueryRecord &bRec, float &pdf) {  // :45-59
        float alpha = mat(bRec.mptr + 4);
        vec3 H = sample_GGX(alpha, pdf);
        bRec.wo = reflect(-bRec.wi, H);
        bRec.eta = 1.0f;
        if(sign(bRec.wi.z) != sign(bRec.wo.z)) {
            pdf = -1.0f;
            return vec3(0.0f);
        }
        float OdotH = std::fabs(dot(bRec.wo, H));
        pdf /= (4.0f * OdotH);
        return eval_RoughConductor(bRec);
    }

    // ------------------------------------------------------------ S/materials/RoughDielectric.glsl
    vec3 eval_RoughDielectric(const BSDFQueryRecord &bRec) {  // :7-48
        vec3 albedo = base_albedo(bRec);
        float alpha = mat(bRec.mptr + 4);
        float eta = mat(bRec.mptr + 7);
        if(bRec.wi.z < 0.0f) eta = 1.0f / eta;
        bool isreflect = sign(bRec.wi.z) == sign(bRec.wo.z);
        vec3 H;
        if(isreflect) H = normalize(bRec.wi + bRec.wo);
        else H = -normalize(bRec.wi + eta * bRec.wo);

        float D = eval_GGX(alpha, H);
        float F = fresnel(dot(bRec.wi, H), eta);
        float G = G_GGX(alpha, std::fabs(bRec.wi.z)) * G_GGX(alpha, std::fabs(bRec.wo.z));
        if(sign(dot(bRec.wi, H)) != sign(bRec.wi.z) || sign(dot(bRec.wo, H)) != sign(bRec.wo.z)) G = 0.0f;

        if(isreflect) {
            return albedo * F * D * G / std::fabs(4.0f * bRec.wi.z * bRec.wo.z);
        } else {
            float cosIH = dot(bRec.wi, H);
            float cosOH = dot(bRec.wo, H);
            float deno = pow2(cosIH + eta * cosOH);
            float value = ((1.0f - F) * D * G * cosIH * cosOH) / (bRec.wi.z * bRec.wo.z * deno);
            return albedo * std::fabs(value);
        }
    }
    float pdf_RoughDielectric(const BSDFQueryRecord &bRec) {  // :50-76
        float alpha = mat(bRec.mptr + 4);
        float eta = mat(bRec.mptr + 7);
        if(bRec.wi.z < 0.0f) eta = 1.0f / eta;
        bool isreflect = sign(bRec.wi.z) == sign(bRec.wo.z);
        vec3 H;
        float dwh_dwo;
        if(isreflect) {
            H = normalize(bRec.wi + bRec.wo);
            dwh_dwo = 1.0f / std::fabs(4.0f * dot(bRec.wi, H));
        } else {
            H = -normalize(bRec.wi + eta * bRec.wo);
            float deno = pow2(dot(bRec.wi, H) + eta * dot(bRec.wo, H));
            dwh_dwo = eta * eta * std::fabs(dot(bRec.wo, H)) / deno;
        }
        float F = fresnel(dot(bRec.wi, H), eta);
        float pdf = eval_GGX(alpha, H);
        if(isreflect) pdf *= F;
        else pdf *= 1.0f - F;
        return pdf * dwh_dwo;
    }
    vec3 sample_RoughDielectric(BSDFQueryRecord &bRec, float &pdf) {  // :78-133
        vec3 albedo = base_albedo(bRec);
        float alpha = mat(bRec.mptr + 4);
        float eta = mat(bRec.mptr + 7);
        if(bRec.wi.z < 0.0f) eta = 1.0f / eta;

        pdf = -1.0f;
        float pdfH;
        vec3 H = sample_GGX(alpha, pdfH);
        float F = fresnel(dot(bRec.wi, H), eta);
        float D = eval_GGX(alpha, H);

        if(rand() < F) {
            bRec.eta = 1.0f;
            bRec.wo = reflect(-bRec.wi, H);
            if(sign(bRec.wi.z) != sign(bRec.wo.z)) return vec3(0.0f);
            float G = G_GGX(alpha, std::fabs(bRec.wi.z)) * G_GGX(alpha, std::fabs(bRec.wo.z));
            if(sign(dot(bRec.wi, H)) != sign(bRec.wi.z) || sign(dot(bRec.wo, H)) != sign(bRec.wo.z)) G = 0.0f;
            pdf = F * pdfH / std::fabs(4.0f * dot(bRec.wi, H));
            vec3 r = albedo * F * D * G / (4.0f * bRec.wi.z * bRec.wo.z);
            return vec3(std::fabs(r.x), std::fabs(r.y), std::fabs(r.z));
        } else {
            bRec.eta = eta;
            bRec.wo = refract_custom(-bRec.wi, H * sign(dot(H, bRec.wi)), eta);
            if(sign(bRec.wi.z) == sign(bRec.wo.z)) return vec3(0.0f);
            float G = G_GGX(alpha, std::fabs(bRec.wi.z)) * G_GGX(alpha, std::fabs(bRec.wo.z));
            if(sign(dot(bRec.wi, H)) != sign(bRec.wi.z) || sign(dot(bRec.wo, H)) != sign(bRec.wo.z)) G = 0.0f;
            float deno = pow2(dot(bRec.wi, H) + eta * dot(bRec.wo, H));
            float value = ((1.0f - F) * D * G * dot(bRec.wi, H) * dot(bRec.wo, H)) / (bRec.wi.z * bRec.wo.z * deno);
            pdf = (1.0f - F) * pdfH * eta * eta * std::fabs(dot(bRec.wo, H)) / deno;
            return albedo * std::fabs(value);
        }
    }

    // ------------------------------------------------------------ S/materials/materials.glsl:82-111
    vec3 eval_material(const BSDFQueryRecord &bRec) {
        cnt.material_evals++;
        int type = roundint(mat(bRec.mptr));
        if(type == 1) return eval_RoughConductor(bRec);
        else if(type == 2) return eval_RoughDielectric(bRec);
        return vec3(0.0f);
    }
    float pdf_material(const BSDFQueryRecord &bRec) {
        cnt.material_evals++;
        int type = roundint(mat(bRec.mptr));
        if(type == 1) return pdf_RoughConductor(bRec);
        else if(type == 2) return pdf_RoughDielectric(bRec);
        return 0.0f;
    }
    vec3 sample_material(BSDFQueryRecord &bRec, float &pdf) {
        cnt.material_evals++;
        int type = roundint(mat(bRec.mptr));
        if(type == 1) return sample_RoughConductor(bRec, pdf);
        else if(type == 2) return sample_RoughDielectric(bRec, pdf);
        pdf = -1.0f;
        return vec3(0.0f);
    }
    vec3 albedo_material(const BSDFQueryRecord &bRec) {
        cnt.material_evals++;
        int type = roundint(mat(bRec.mptr));
        if(type == 1 || type == 2) return base_albedo(bRec);  // RoughConductor.glsl:61-70, RoughDielectric.glsl:135-144
        return vec3(0.0f);
    }

    // ------------------------------------------------------------ S/pathtracing.glsl:148-166
    static Frame create_coord(vec3 n) {
        Frame f;
        if(std::fabs(n.z) > 1.0f - EPS) f.s = normalize(cross(n, vec3(1, 0, 0)));
        else f.s = normalize(cross(n, vec3(0, 0, 1)));
        f.t = cross(n, f.s);
        f.n = n;
        return f;
    }
    static vec3 to_world(const Frame &f, vec3 v) { return v.x * f.s + v.y * f.t + v.z * f.n; }
    static vec3 to_local(const Frame &f, vec3 v) { return vec3(dot(v, f.s), dot(v, f.t), dot(v, f.n)); }

    // ------------------------------------------------------------ S/pathtracing.glsl:208-221
    static bool intersect_aabb(const Ray &ray, const float aa[4], const float bb[4], float &nearT) {
        float tmi = 0.0f, tmx = INF;
        vec3 t1 = (vec3(aa[0], aa[1], aa[2]) - ray.ori) / ray.dir;
        vec3 t2 = (vec3(bb[0], bb[1], bb[2]) - ray.ori) / ray.dir;
        tmi = std::fmax(tmi, std::fmin(t1.x, t2.x));
        tmi = std::fmax(tmi, std::fmin(t1.y, t2.y));
        tmi = std::fmax(tmi, std::fmin(t1.z, t2.z));
        tmx = std::fmin(tmx, std::fmax(t1.x, t2.x));
        tmx = std::fmin(tmx, std::fmax(t1.y, t2.y));
        tmx = std::fmin(tmx, std::fmax(t1.z, t2.z));
        nearT = tmi;
        return tmx >= tmi && tmx >= RAY_MINDIS;
    }

    // ------------------------------------------------------------ S/pathtracing.glsl:224-249 (+ :40-49, :183-186)
    Intersection intersect_triangle(const Ray &ray, int triangleIndex) {
        const float *tr = sc->triangles + (size_t)triangleIndex * 24;
        vec3 v0(tr[0], tr[1], tr[2]), v1(tr[3], tr[4], tr[5]), v2(tr[6], tr[7], tr[8]);
        vec3 n0(tr[9], tr[10], tr[11]), n1(tr[12], tr[13], tr[14]), n2(tr[15], tr[16], tr[17]);
        vec3 E1 = v1 - v0;
        vec3 E2 = v2 - v0;
        vec3 S = ray.ori - v0;
        vec3 S1 = cross(ray.dir, E2);
        vec3 S2 = cross(S, E1);
        float k = 1.0f / dot(S1, E1);
        float t = dot(S2, E2) * k;
        float u = dot(S1, S) * k;
        float v = dot(S2, ray.dir) * k;

        Intersection r;
        r.exist = (RAY_MINDIS <= t && 0.0f <= u && 0.0f <= v && u + v <= 1.0f);
        r.position = ray.ori + t * ray.dir;
        float w = 1.0f - u - v;
        r.normal = normalize(n0 * w + n1 * u + n2 * v);
        r.t = t;
        r.uv.x = tr[18] * w + tr[20] * u + tr[22] * v;
        r.uv.y = tr[19] * w + tr[21] * u + tr[23] * v;
        r.instanceIndex = -1;
        r.triangleIndex = triangleIndex;
        r.bu = u;
        r.bv = v;
        return r;
    }

    // ------------------------------------------------------------ S/pathtracing.glsl:254-289
    void intersect_meshBVH(int instanceIndex, const Ray &_ray, Intersection &result) {
        const Instance &ins = instances()[instanceIndex];
        int root = ins.meshIndex;
        int starth = stack_h;
        stack[++stack_h] = root;
        if((uint64_t)stack_h > cnt.max_stack) cnt.max_stack = stack_h;

        const float *m = ins.w2l;  // column major: m[c*4 + r]
        Ray ray;
        // (w2l * vec4(ori, 1)).xyz with glm's association
        for(int r = 0; r < 3; r++) {
            float a = m[0 + r] * _ray.ori.x + m[4 + r] * _ray.ori.y;
            float b = m[8 + r] * _ray.ori.z + m[12 + r] * 1.0f;
            (&ray.ori.x)[r] = a + b;
        }
        // w2l[0].xyz * d[0] + w2l[1].xyz * d[1] + w2l[2].xyz * d[2]   (scale kept)
        for(int r = 0; r < 3; r++)
            (&ray.dir.x)[r] = (m[0 + r] * _ray.dir.x + m[4 + r] * _ray.dir.y) + m[8 + r] * _ray.dir.z;

        const Node *nodes = mesh_nodes();
        while(stack_h > starth) {
            int id = stack[stack_h--];
            const Node &cur = nodes[id];
            if(counting_extend) cnt.ext_nodes++;
            if(cur.tri >= 0) {
                cnt.leaf_tests++;
                if(counting_extend) cnt.ext_leaf_tests++;
                Intersection isect = intersect_triangle(ray, cur.tri);
                if(isect.exist && isect.t < result.t) {
                    // normal * mat3(w2l): row vector times matrix = dot with each column
                    vec3 n = isect.normal;
                    vec3 nn(n.x * m[0] + n.y * m[1] + n.z * m[2], n.x * m[4] + n.y * m[5] + n.z * m[6],
                            n.x * m[8] + n.y * m[9] + n.z * m[10]);
                    isect.normal = normalize(nn);
                    isect.position = _ray.ori + isect.t * _ray.dir;
                    result = isect;
                    result.instanceIndex = instanceIndex;
                }
            } else {
                cnt.mesh_inner_nodes++;
                float tnear;
                if(!intersect_aabb(ray, cur.aa, cur.bb, tnear)) continue;
                if(tnear > result.t) continue;
                stack[++stack_h] = cur.ls;
                stack[++stack_h] = cur.rs;
                if((uint64_t)stack_h > cnt.max_stack) cnt.max_stack = stack_h;
            }
        }
    }

    // ------------------------------------------------------------ S/pathtracing.glsl:292-316
    Intersection intersect_sceneBVH(const Ray &ray) {
        cnt.rays++;
        Intersection result;
        // A ray with a NaN in its origin or direction (the reference's refract() and GGX
        // sampling produce them: sqrt of a rounding-negative, math.glsl:27) cannot hit anything:
        // every triangle test compares NaN and fails.  But NaN-dropping min/max make every box
        // pass, so the walk below would visit the whole scene for nothing.  Same answer, no walk.
        if(std::isnan(ray.ori.x) || std::isnan(ray.ori.y) || std::isnan(ray.ori.z) || std::isnan(ray.dir.x) ||
           std::isnan(ray.dir.y) || std::isnan(ray.dir.z))
            return result;
        stack_h = 0;
        stack[++stack_h] = 0;
        const Node *nodes = scene_nodes();
        while(stack_h > 0) {
            int id = stack[stack_h--];
            const Node &cur = nodes[id];
            cnt.scene_nodes++;
            if(counting_extend) cnt.ext_nodes++;
            float tnear;
            if(!intersect_aabb(ray, cur.aa, cur.bb, tnear)) continue;
            if(tnear > result.t) continue;
            if(cur.inst >= 0) {
                intersect_meshBVH(cur.inst, ray, result);
                continue;
            }
            stack[++stack_h] = cur.ls;
            stack[++stack_h] = cur.rs;
            if((uint64_t)stack_h > cnt.max_stack) cnt.max_stack = stack_h;
        }
        if(result.exist) cnt.hits++;
        return result;
    }

    // ------------------------------------------------------------ S/pathtracing.glsl:358-399
    vec3 spherical_sample(float &pdf) {
        float z = rand() * 2.0f - 1.0f;
        float r = std::fmax(0.0f, std::sqrt(1.0f - z * z));
        float phi = 2.0f * PI * rand();
        pdf = 0.25f * INV_PI;
        return vec3(r * m_cos(phi), r * m_sin(phi), z);
    }
    vec3 get_background_color(vec3 v) {
        float ux = m_atan2(v.z, v.x), uy = m_asin(v.y);
        ux /= 2.0f * PI;
        uy /= PI;
        ux += 0.5f;
        uy += 0.5f;
        uy = 1.0f - uy;
        float c[4];
        texture(sc->skybox, ux, uy, c);
        return vec3(c[0], c[1], c[2]);
    }
    vec3 skybox_sample(float &pdf) {
        float x = rand();
        float y = rand();
        float c[4];
        texture(sc->skybox_samplecache, x, y, c);
        float theta = (c[0] - 0.5f) * 2.0f * PI;
        float phi = (1.0f - c[1] - 0.5f) * PI;
        float r = m_cos(phi);
        vec3 wi(r * m_cos(theta), m_sin(phi), r * m_sin(theta));
        float w2a = (2.0f * PI * PI * std::sqrt(1.0f - wi.y * wi.y)) / (float)(pm->SKY_H * pm->SKY_W);
        if(w2a == 0.0f) w2a = (float)pm->SKY_W * PI / (float)pm->SKY_H;
        pdf = c[2] / w2a;
        return wi;
    }
    float skybox_sample_pdf(vec3 v) {
        vec3 l = get_background_color(v);
        float lw = (l.x * 0.2f + l.y * 0.7f + l.z * 0.1f) / pm->skybox_Light_SUM;
        float w2a = (2.0f * PI * PI * std::sqrt(1.0f - v.y * v.y)) / (float)(pm->SKY_H * pm->SKY_W);
        if(w2a == 0.0f) w2a = (float)pm->SKY_W * PI / (float)pm->SKY_H;
        return lw / w2a;
    }

    // ------------------------------------------------------------ S/pathtracing.glsl:489-582
    void shade_mis_advanced(Ray ray, const Intersection &first_isect, vec3 &resultDI, vec3 &resultGI) {
        resultDI = vec3(0.0f);
        resultGI = vec3(0.0f);
        vec3 history(1.0f);
        Intersection isect = first_isect;
        if(!isect.exist) {
            resultDI = resultGI = get_background_color(ray.dir);
            return;
        }
        const bool skyL = pm->SkyboxLighting != 0, skyS = pm->SkyboxSampling != 0;
        for(int dep = 0; dep < pm->MAX_DEPTH; dep++) {
            cnt.shaded_vertices++;
            int materialPtr = instances()[isect.instanceIndex].materialPtr;
            Frame coord = create_coord(isect.normal);
            vec3 wi = to_local(coord, -ray.dir);
            vec3 global_wo;
            float pdf;
            BSDFQueryRecord bRec{wi, vec3(0.0f), materialPtr, isect.uv, 1.0f};

            if(skyL && skyS) {  // skybox direct light, :516-528
                global_wo = skybox_sample(pdf);
                if(pdf > 0.0f) {
                    cnt.rays_shadow++;
                    counting_extend = false;
                    Intersection tsect = intersect_sceneBVH(Ray{isect.position, global_wo});
                    counting_extend = true;
                    if(!tsect.exist) {
                        bRec.wo = to_local(coord, global_wo);
                        vec3 fr = eval_material(bRec);
                        vec3 recv_col = get_background_color(global_wo) * fr * std::fabs(bRec.wo.z) / (pdf + pdf_material(bRec));
                        resultGI = resultGI + history * recv_col;
                        if(dep == 0) resultDI = recv_col;
                    }
                }
            }

            for(int li = 0; li < pm->lightCount; li++) {  // :531-554
                const Light &light = lights()[li];
                vec3 dir, radiance;
                float dis2;
                if(light.type == 0) {
                    dir = vec3(light.x, light.y, light.z) - isect.position;
                    dis2 = dot(dir, dir);
                    radiance = vec3(light.r, light.g, light.b) / dis2;
                    dir = normalize(dir);
                } else {
                    dir = -normalize(vec3(light.x, light.y, light.z));
                    dis2 = INF;
                    radiance = vec3(light.r, light.g, light.b);
                }
                cnt.rays_shadow++;
                counting_extend = false;
                Intersection tsect = intersect_sceneBVH(Ray{isect.position, dir});
                counting_extend = true;
                if(!tsect.exist || tsect.t * tsect.t > dis2) {
                    bRec.wo = to_local(coord, dir);
                    vec3 fr = eval_material(bRec);
                    vec3 recv_col = radiance * fr * std::fabs(bRec.wo.z);
                    resultGI = resultGI + history * recv_col;
                    if(dep == 0) resultDI = recv_col;
                }
            }

            vec3 fr = sample_material(bRec, pdf);  // :558-562
            if(!pm->BRDFSampling) {
                bRec.wo = spherical_sample(pdf);
                fr = eval_material(bRec);
            }
            global_wo = to_world(coord, bRec.wo);
            if(pdf <= 0.0f) break;

            ray = Ray{isect.position, global_wo};
            cnt.rays_extend++;
            isect = intersect_sceneBVH(ray);
            if(!isect.exist) {
                if(skyL && skyS) {
                    vec3 recv_col = get_background_color(global_wo) * fr * std::fabs(bRec.wo.z) / (pdf + skybox_sample_pdf(global_wo));
                    resultGI = resultGI + history * recv_col;
                } else if(skyL) {
                    vec3 recv_col = get_background_color(global_wo) * fr * std::fabs(bRec.wo.z) / pdf;
                    resultGI = resultGI + history * recv_col;
                }
                break;
            } else {
                history = history * (fr * std::fabs(bRec.wo.z) / pdf);
            }
        }
    }

    // ------------------------------------------------------------ S/pathtracing.glsl:585-691
    void pixel(uint32_t px, uint32_t py, const orc_outputs &out) {
        const int W = pm->SCREEN_W, H = pm->SCREEN_H;
        uint32_t pixelPtr = py * (uint32_t)W + px;
        uint32_t seed = (px * 1973u + py * 9277u + pm->frameCounter * pm->currentspp * 26699u) | 1u;
        sobolseed = seed;
        sobolcurdim = 1u;
        cnt.samples++;

        float p_x = (float)px + pm->jitter[0], p_y = (float)py + pm->jitter[1];
        float disz = (float)W * 0.5f / std::tan(pm->fov / 2.0f);
        const float *m = pm->v2wMat;
        vec3 ori(m[12], m[13], m[14]);  // v2wMat * (0,0,0,1)
        float cx = p_x - (float)(W / 2), cy = p_y - (float)(H / 2), cz = -disz;
        vec3 d;
        for(int r = 0; r < 3; r++) {
            float a = m[0 + r] * cx + m[4 + r] * cy;
            float b = m[8 + r] * cz + m[12 + r] * 0.0f;
            (&d.x)[r] = a + b;
        }
        Ray ray{ori, normalize(d)};

        Intersection first_isect;
        float depth;
        int instanceIndex;
        if(pm->RasterizaionFor1st) {
            // :606-624: the first hit comes from the raster pass's images
            float suv_x0 = ((float)px + 0.5f) / (float)W, suv_y0 = ((float)py + 0.5f) / (float)H;
            float wu = suv_x0 * (float)W / (float)pm->WINDOW_W;
            float wv = (suv_y0 - 1.0f) * (float)H / (float)pm->WINDOW_H + 1.0f;
            float c[4];
            texture(sc->depthGBufferTexture, wu, wv, c);
            depth = c[0];
            instanceIndex = float_to_int(texture_nearest(sc->instanceIndexGBufferTexture, wu, wv) + 0.01f);
            first_isect.t = depth;
            first_isect.exist = depth <= 100000.0f;
            texture(sc->normalGBufferTexture, wu, wv, c);
            first_isect.normal = normalize(vec3(c[0], c[1], c[2]));
            texture(sc->uvGBufferTexture, wu, wv, c);
            first_isect.uv = {c[0], c[1]};
            first_isect.position = ray.ori + ray.dir * first_isect.t;
            first_isect.instanceIndex = instanceIndex;
            // an index that names no instance would be an out-of-bounds read in the reference
            // (:641); such a pixel is treated as not covered
            if(instanceIndex < 0 || instanceIndex >= sc->n_instances) first_isect.exist = false;
        } else {
            cnt.rays_extend++;
            first_isect = intersect_sceneBVH(ray);  // :626-628
            // The reference keeps the RASTER depth and instance index here (:615-616): the depth output is the
            // raster depth (:684) and the albedo is looked up through the raster instance index (:641), while
            // everything else comes from the traced hit.  With the two raster images bound that is reproduced;
            // without a raster pass both are those of the traced hit (1e6 on a miss = the raster clear value).
            depth = first_isect.exist ? first_isect.t : 1000000.0f;
            instanceIndex = first_isect.instanceIndex;
            if(sc->depthGBufferTexture.data && sc->instanceIndexGBufferTexture.data) {
                float suv_x0 = ((float)px + 0.5f) / (float)W, suv_y0 = ((float)py + 0.5f) / (float)H;
                float wu = suv_x0 * (float)W / (float)pm->WINDOW_W;
                float wv = (suv_y0 - 1.0f) * (float)H / (float)pm->WINDOW_H + 1.0f;
                float c[4];
                texture(sc->depthGBufferTexture, wu, wv, c);
                depth = c[0];
                const int raster_index = float_to_int(texture_nearest(sc->instanceIndexGBufferTexture, wu, wv) + 0.01f);
                // (an index that names no instance is an out-of-bounds read in the reference: the traced one stands in)
                if(raster_index >= 0 && raster_index < sc->n_instances) instanceIndex = raster_index;
            }
        }

        vec3 resultDI, resultGI, albedo;
        shade_mis_advanced(ray, first_isect, resultDI, resultGI);
        if(any_nan(resultDI)) resultDI = vec3(0.0f);
        if(any_nan(resultGI)) resultGI = vec3(0.0f);
        vec3 resultIDI = resultGI - resultDI;

        if(first_isect.exist) {
            int materialPtr = instances()[instanceIndex].materialPtr;
            BSDFQueryRecord bRec{vec3(0.0f), vec3(0.0f), materialPtr, first_isect.uv, 1.0f};
            albedo = albedo_material(bRec);
        } else {
            albedo = resultGI;
        }
        resultDI = resultDI / vmax(vec3(EPS), albedo);
        resultIDI = resultIDI / vmax(vec3(EPS), albedo);

        const float *bp = pm->backprojMat;
        vec3 P = first_isect.position;
        float ndc[4];
        for(int r = 0; r < 4; r++) {
            float a = bp[0 + r] * P.x + bp[4 + r] * P.y;
            float b = bp[8 + r] * P.z + bp[12 + r] * 1.0f;
            ndc[r] = a + b;
        }
        float ndx = ndc[0] / ndc[3], ndy = ndc[1] / ndc[3];
        float suv_x = ((float)px + 0.5f) / (float)W, suv_y = ((float)py + 0.5f) / (float)H;
        float motion_x = suv_x - (ndx + 1.0f) / 2.0f, motion_y = suv_y - (ndy + 1.0f) / 2.0f;

        size_t o3 = (size_t)pixelPtr * 3;
        if(pm->currentspp != 1u) {
            vec3 lastdi(out.directLum[o3], out.directLum[o3 + 1], out.directLum[o3 + 2]);
            vec3 lastidi(out.indirectLum[o3], out.indirectLum[o3 + 1], out.indirectLum[o3 + 2]);
            vec3 lastalbedo(out.albedo[o3], out.albedo[o3 + 1], out.albedo[o3 + 2]);
            if(pm->accum_mode == 1) {  // extension: plain sums
                resultDI = lastdi + resultDI;
                resultIDI = lastidi + resultIDI;
                albedo = lastalbedo + albedo;
            } else {
                float a = 1.0f / (float)pm->currentspp;
                resultDI = mixv(lastdi, resultDI, a);
                resultIDI = mixv(lastidi, resultIDI, a);
                albedo = mixv(lastalbedo, albedo, a);
            }
        }
        if(out.directLum) { out.directLum[o3] = resultDI.x; out.directLum[o3 + 1] = resultDI.y; out.directLum[o3 + 2] = resultDI.z; }
        if(out.indirectLum) { out.indirectLum[o3] = resultIDI.x; out.indirectLum[o3 + 1] = resultIDI.y; out.indirectLum[o3 + 2] = resultIDI.z; }
        if(out.albedo) { out.albedo[o3] = albedo.x; out.albedo[o3 + 1] = albedo.y; out.albedo[o3 + 2] = albedo.z; }
        if(out.motion) { out.motion[(size_t)pixelPtr * 2] = motion_x; out.motion[(size_t)pixelPtr * 2 + 1] = motion_y; }
        if(out.depth) out.depth[pixelPtr] = depth;
        if(out.normal) { out.normal[o3] = first_isect.normal.x; out.normal[o3 + 1] = first_isect.normal.y; out.normal[o3 + 2] = first_isect.normal.z; }
        if(out.instanceIndex) out.instanceIndex[pixelPtr] = (float)first_isect.instanceIndex;
    }
};

void add_counters(orc_counters &a, const orc_counters &b) {
    const uint64_t *s = (const uint64_t *)&b;
    uint64_t *d = (uint64_t *)&a;
    const size_t n = sizeof(orc_counters) / 8;
    const size_t max_idx = offsetof(orc_counters, max_stack) / 8;
    for(size_t i = 0; i < n; i++) {
        if(i == max_idx) d[i] = d[i] > s[i] ? d[i] : s[i];
        else d[i] += s[i];
    }
}

int pick_threads(int n_threads) {
#ifdef _OPENMP
    return n_threads > 0 ? n_threads : omp_get_max_threads();
#else
    (void)n_threads;
    return 1;
#endif
}

}  // namespace

extern "C" {

void orc_intersect(const orc_scene *scene, const float *rays, size_t n, orc_hit *hits, orc_counters *counters,
                   int n_threads) {
    orc_counters total;
    memset(&total, 0, sizeof total);
    int nt = pick_threads(n_threads);
#pragma omp parallel num_threads(nt)
    {
        Ctx c;
        c.sc = scene;
        c.pm = nullptr;
#pragma omp for schedule(dynamic, 4096)
        for(long long i = 0; i < (long long)n; i++) {
            Ray r{vec3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), vec3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5])};
            Intersection x = c.intersect_sceneBVH(r);
            if(hits) {
                hits[i].instance = x.instanceIndex;
                hits[i].triangle = x.exist ? x.triangleIndex : -1;
                hits[i].t = x.t;
                hits[i].u = x.exist ? x.bu : 0.0f;
                hits[i].v = x.exist ? x.bv : 0.0f;
            }
        }
#pragma omp critical
        add_counters(total, c.cnt);
    }
    if(counters) *counters = total;
}

void orc_render(const orc_scene *scene, const orc_params *params, const orc_outputs *out, orc_counters *counters,
                int n_threads) {
    orc_counters total;
    memset(&total, 0, sizeof total);
    int nt = pick_threads(n_threads);
    const int W = params->SCREEN_W, H = params->SCREEN_H;
#pragma omp parallel num_threads(nt)
    {
        Ctx c;
        c.sc = scene;
        c.pm = params;
#pragma omp for schedule(dynamic, 1)
        for(int y = 0; y < H; y++)
            for(int x = 0; x < W; x++) c.pixel((uint32_t)x, (uint32_t)y, *out);
#pragma omp critical
        add_counters(total, c.cnt);
    }
    if(counters) *counters = total;
}

float orc_sobol(uint32_t d, uint32_t i) { return sobol(d, i); }

void orc_material(const orc_scene *scene, int which, int mptr, const float wi[3], const float wo[3],
                  const float uv[2], uint32_t seed, uint32_t dim, float *out) {
    Ctx c;
    c.sc = scene;
    c.pm = nullptr;
    c.sobolseed = seed;
    c.sobolcurdim = dim;
    BSDFQueryRecord b{vec3(wi[0], wi[1], wi[2]), vec3(wo[0], wo[1], wo[2]), mptr, vec2{uv[0], uv[1]}, 1.0f};
    if(which == 0) {
        vec3 r = c.eval_material(b);
        out[0] = r.x; out[1] = r.y; out[2] = r.z;
    } else if(which == 1) {
        out[0] = c.pdf_material(b);
    } else if(which == 2) {
        float pdf;
        vec3 r = c.sample_material(b, pdf);
        out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = pdf;
        out[4] = b.wo.x; out[5] = b.wo.y; out[6] = b.wo.z; out[7] = b.eta;
    } else {
        vec3 r = c.albedo_material(b);
        out[0] = r.x; out[1] = r.y; out[2] = r.z;
    }
}

void orc_texture_fetch(const orc_texture *tex, float s, float t, float out[4]) {
    Ctx c;
    c.sc = nullptr;
    c.pm = nullptr;
    c.texture(*tex, s, t, out);
}

void orc_background(const orc_scene *scene, const float dir[3], float out[3]) {
    Ctx c;
    c.sc = scene;
    c.pm = nullptr;
    vec3 r = c.get_background_color(vec3(dir[0], dir[1], dir[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

void orc_skybox_sample(const orc_scene *scene, const orc_params *params, uint32_t seed, uint32_t dim, float out[4]) {
    Ctx c;
    c.sc = scene;
    c.pm = params;
    c.sobolseed = seed;
    c.sobolcurdim = dim;
    float pdf;
    vec3 w = c.skybox_sample(pdf);
    out[0] = w.x; out[1] = w.y; out[2] = w.z; out[3] = pdf;
}

float orc_skybox_sample_pdf(const orc_scene *scene, const orc_params *params, const float dir[3]) {
    Ctx c;
    c.sc = scene;
    c.pm = params;
    return c.skybox_sample_pdf(vec3(dir[0], dir[1], dir[2]));
}

// SVGF_Merge.glsl:24-52
void orc_post_merge(size_t npix, const float *di, const float *idi, const float *albedo, float *color) {
    for(size_t i = 0; i < npix; i++) {
        vec3 d(di[i * 3], di[i * 3 + 1], di[i * 3 + 2]), id(idi[i * 3], idi[i * 3 + 1], idi[i * 3 + 2]);
        vec3 al(albedo[i * 3], albedo[i * 3 + 1], albedo[i * 3 + 2]);
        vec3 result = (d + id) * al;
        if(std::isnan(result.x) || std::isnan(result.y) || std::isnan(result.z)) result = vec3(10000.0f);
        if(std::isinf(result.x) || std::isinf(result.y) || std::isinf(result.z)) result = vec3(10000.0f);
        color[i * 3] = result.x; color[i * 3 + 1] = result.y; color[i * 3 + 2] = result.z;
    }
}

// StaticBlender.glsl:22-61
void orc_post_static_blend(size_t npix, uint32_t frame, const float *colorbuf, float *hist_color, float *hist_moment) {
    for(size_t i = 0; i < npix; i++) {
        vec3 color(colorbuf[i * 3], colorbuf[i * 3 + 1], colorbuf[i * 3 + 2]);
        float lum = color.x * 0.212671f + color.y * 0.715160f + color.z * 0.072169f;
        float mx = lum, my = lum * lum;
        vec3 lastcolor(hist_color[i * 3], hist_color[i * 3 + 1], hist_color[i * 3 + 2]);
        float lmx = hist_moment[i * 2], lmy = hist_moment[i * 2 + 1];
        float sigma = std::sqrt(std::fmax(1.0f, lmy - lmx * lmx));
        float a = 1.0f / (float)frame;
        if(frame != 1u && std::fabs(mx - lmx) > sigma * 3.0f) color = lastcolor;
        else color = mixv(lastcolor, color, a);
        hist_color[i * 3] = color.x; hist_color[i * 3 + 1] = color.y; hist_color[i * 3 + 2] = color.z;
        hist_moment[i * 2] = mixf(lmx, mx, a);
        hist_moment[i * 2 + 1] = mixf(lmy, my, a);
    }
}

// ToneMappingGamma.glsl:17-54 (ACES, then pow 0.45 per channel)
void orc_post_tonemap(size_t n, const float *in, float *out) {
    const float A = 2.51f, B = 0.03f, C = 2.43f, D = 0.59f, E = 0.14f;
    for(size_t i = 0; i < n; i++) {
        float c = in[i];
        float t = (c * (A * c + B)) / (c * (C * c + D) + E);
        out[i] = m_pow(t, 0.45f);
    }
}

// src/SSBOBuffer.h:47-56: datau = min(255, int(x * 255)) as a byte, then cv::flip(img, img, 0)
void orc_quantize_image(const float *data, int w, int h, int c, uint8_t *out) {
    const size_t row = (size_t)w * c;
    for(int y = 0; y < h; y++)
        for(size_t k = 0; k < row; k++) {
            float x = data[row * (size_t)(h - 1 - y) + k] * 255.0f;
            long long v = std::isnan(x) ? 0 : (x >= 2147483520.0f ? 2147483647ll : (x <= -2147483648.0f ? -2147483648ll : (long long)(int)x));
            if(v > 255) v = 255;
            out[row * (size_t)y + k] = (uint8_t)(unsigned long long)v;
        }
}

// ---------------------------------------------------------------- denoising chain
namespace {
inline float m_exp(float x) { return m_exp2(x * 1.44269504088896341f); }
// pow(x, 128), x >= 0: seven squarings (well inside what GLSL allows for pow)
inline float pow128(float x) {
    for(int k = 0; k < 7; k++) x = x * x;
    return x;
}
inline vec3 rd3(const float *b, long long ptr) { return vec3(b[ptr * 3 + 0], b[ptr * 3 + 1], b[ptr * 3 + 2]); }
inline void wr3(float *b, long long ptr, vec3 v) { b[ptr * 3 + 0] = v.x; b[ptr * 3 + 1] = v.y; b[ptr * 3 + 2] = v.z; }
inline float lumf(vec3 c) { return c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f; }
// int(x): undefined in GLSL outside the int range; saturating, NaN -> 0
inline int to_int(float x) {
    if(x != x) return 0;
    if(x >= 2147483520.0f) return 2147483647;
    if(x <= -2147483648.0f) return (int)0x80000000;
    return (int)x;
}
// geometry_test, SVGF_TemporalFilter.glsl:50-65 and TAA.glsl:64-79
inline bool same_surface(const float *normal, const float *inst, const float *h_normal, const float *h_inst, long long p1, long long p2) {
    return dot(rd3(normal, p1), rd3(h_normal, p2)) > 0.9f && std::fabs(inst[p1] - h_inst[p2]) < 0.1f;
}
const float k_atrous[4] = {3.0f / 8, 1.0f / 4.0f, 1.0f / 16, 1.0f / 32};
}  // namespace

void orc_svgf_temporal(int W, int H, int first_frame, const float *colorB, const float *normal, const float *inst,
                       const float *motionB, float *o_color, float *o_moment, float *o_num, const float *h_color,
                       const float *h_moment, const float *h_num, const float *h_normal, const float *h_inst) {
    for(int py = 0; py < H; py++)
        for(int px = 0; px < W; px++) {
            const long long pixelPtr = (long long)py * W + px;
            const float su = ((float)px + 0.5f) / (float)W, sv = ((float)py + 0.5f) / (float)H;
            vec3 color = rd3(colorB, pixelPtr);
            float lum = lumf(color);
            float mx = lum, my = lum * lum;
            float motx = motionB[pixelPtr * 2], moty = motionB[pixelPtr * 2 + 1];
            if(first_frame) {
                wr3(o_color, pixelPtr, color);
                o_moment[pixelPtr * 2] = mx; o_moment[pixelPtr * 2 + 1] = my;
                o_num[pixelPtr] = 1.0f;
                continue;
            }
            vec3 mu(0.0f), var(0.0f);
            int nv = 0;
            for(int i = 0; i < 9; i++) {
                int ix = px + (i / 3 - 1), iy = py + (i % 3 - 1);
                if(ix < 0 || ix >= W || iy < 0 || iy >= H) continue;
                vec3 r = rd3(colorB, (long long)iy * W + ix);
                nv++;
                mu = mu + r;
                var = var + r * r;
            }
            mu = mu / (float)nv;
            var = vmax(vec3(0.0f), var / (float)nv - mu * mu);
            vec3 sigma = vec3(std::sqrt(var.x), std::sqrt(var.y), std::sqrt(var.z)) * 5.0f;

            float lu = su - motx, lv = sv - moty;
            if(lu >= 1.0f || lv >= 1.0f || lu < 0.0f || lv < 0.0f) {
                o_num[pixelPtr] = 1.0f;
            } else {
                int lix = to_int(lu * (float)W), liy = to_int(lv * (float)H);
                long long lastPixelPtr = (long long)liy * W + lix;
                // interpolate(last_uv, pixelPtr, ...)
                int lbx = to_int(std::floor(lu * (float)W - 0.5f)), lby = to_int(std::floor(lv * (float)H - 0.5f));
                float pointx = lu * (float)W, pointy = lv * (float)H;
                vec3 outcolor(0.0f);
                float omx = 0.0f, omy = 0.0f, tot = 0.0f;
                for(int i = 0; i < 4; i++) {
                    long long ix = (long long)lbx + i / 2, iy = (long long)lby + i % 2;
                    if(ix < 0 || ix >= W || iy < 0 || iy >= H) continue;
                    long long ptr = iy * W + ix;
                    if(!same_surface(normal, inst, h_normal, h_inst, pixelPtr, ptr)) continue;
                    float dx = std::fabs(pointx - (float)ix - 0.5f), dy = std::fabs(pointy - (float)iy - 0.5f);
                    float weight = std::fmax(0.0f, (1.0f - dx) * (1.0f - dy));
                    outcolor = outcolor + rd3(h_color, ptr) * weight;
                    omx += h_moment[ptr * 2] * weight;
                    omy += h_moment[ptr * 2 + 1] * weight;
                    tot += weight;
                }
                vec3 lastcolor = outcolor / tot;
                float lmx = omx / tot, lmy = omy / tot;
                bool valid = tot > 0.00001f;
                vec3 dir = lastcolor - mu;
                if(!valid || std::fabs(dir.x) > sigma.x || std::fabs(dir.y) > sigma.y || std::fabs(dir.z) > sigma.z) {
                    o_num[pixelPtr] = 1.0f;
                } else {
                    if(lix < 0 || lix >= W || liy < 0 || liy >= H) lastPixelPtr = pixelPtr;
                    float historyLen = h_num[lastPixelPtr];
                    o_num[pixelPtr] = historyLen + 1.0f;
                    color = mixv(lastcolor, color, 0.2f);
                    mx = mixf(lmx, mx, 0.2f);
                    my = mixf(lmy, my, 0.2f);
                }
            }
            wr3(o_color, pixelPtr, color);
            o_moment[pixelPtr * 2] = mx; o_moment[pixelPtr * 2 + 1] = my;
        }
}

void orc_svgf_variance(int W, int H, const float *momentB, const float *normalB, const float *depthB, const float *numB,
                       float *variance) {
    for(int py = 0; py < H; py++)
        for(int px = 0; px < W; px++) {
            const long long pixelPtr = (long long)py * W + px;
            vec3 normal = rd3(normalB, pixelPtr);
            float depth = depthB[pixelPtr];
            float totalWeight = 0.0f, rx = 0.0f, ry = 0.0f;
            const int reach = numB[pixelPtr] <= 4.0f ? 3 : 1;
            for(int i = -reach; i <= reach; i++)
                for(int j = -reach; j <= reach; j++) {
                    int ix = px + i, iy = py + j;
                    if(ix >= W || ix < 0 || iy >= H || iy < 0) continue;
                    long long ptr = (long long)iy * W + ix;
                    float w = k_atrous[std::abs(i)] * k_atrous[std::abs(j)];
                    if(reach == 3)
                        w = w * pow128(std::fmax(0.0f, dot(normal, rd3(normalB, ptr)))) * m_exp(-std::fabs(depth - depthB[ptr]) / 2.0f);
                    totalWeight += w;
                    rx += w * momentB[ptr * 2];
                    ry += w * momentB[ptr * 2 + 1];
                }
            float mx = rx / totalWeight, my = ry / totalWeight;
            variance[pixelPtr] = std::fmax(0.0f, my - mx * mx);
        }
}

void orc_svgf_spatial(int W, int H, int step, const float *colorB, const float *normalB, const float *depthB,
                      const float *varianceB, float *out) {
    for(int py = 0; py < H; py++)
        for(int px = 0; px < W; px++) {
            const long long pixelPtr = (long long)py * W + px;
            vec3 color = rd3(colorB, pixelPtr), normal = rd3(normalB, pixelPtr);
            float depth = depthB[pixelPtr];
            float sigma = std::sqrt(varianceB[pixelPtr]);
            float totalWeight = 0.0f;
            vec3 result(0.0f);
            for(int i = -2; i <= 2; i++)
                for(int j = -2; j <= 2; j++) {
                    long long ix = px + (long long)i * step, iy = py + (long long)j * step;
                    if(ix >= W || ix < 0 || iy >= H || iy < 0) continue;
                    long long ptr = iy * W + ix;
                    vec3 c2 = rd3(colorB, ptr);
                    vec3 diff = color - c2;
                    float len = std::sqrt(dot(diff, diff));
                    float w = k_atrous[std::abs(i)] * k_atrous[std::abs(j)] *
                              pow128(std::fmax(0.0f, dot(normal, rd3(normalB, ptr)))) *
                              m_exp(-std::fabs(depth - depthB[ptr]) / 2.0f) *
                              m_exp(-std::fabs(len) / (sigma * 4.0f + 0.00001f));
                    totalWeight += w;
                    result = result + w * c2;
                }
            result = result / totalWeight;
            if(any_nan(result) || std::isinf(result.x) || std::isinf(result.y) || std::isinf(result.z)) result = vec3(10000.0f, 0.0f, 0.0f);
            wr3(out, pixelPtr, result);
        }
}

namespace {
vec3 RGB2YCoCgR(vec3 c) {  // TAA.glsl:40-50
    vec3 o;
    o.y = c.x - c.z;
    float temp = c.z + o.y / 2.0f;
    o.z = c.y - temp;
    o.x = temp + o.z / 2.0f;
    return o;
}
vec3 YCoCgR2RGB(vec3 c) {  // :52-62
    vec3 o;
    float temp = c.x - c.z / 2.0f;
    o.y = c.z + temp;
    o.z = temp - c.y / 2.0f;
    o.x = o.z + c.y;
    return o;
}
}  // namespace

void orc_taa(int W, int H, const float *colorB, const float *h_color, const float *motionB, const float *normal,
             const float *inst, const float *h_normal, const float *h_inst, float *out) {
    for(int py = 0; py < H; py++)
        for(int px = 0; px < W; px++) {
            const long long pixelPtr = (long long)py * W + px;
            const float su = ((float)px + 0.5f) / (float)W, sv = ((float)py + 0.5f) / (float)H;
            vec3 cur = rd3(colorB, pixelPtr);
            vec3 mu(0.0f), var(0.0f);
            int nv = 0;
            for(int i = 0; i < 9; i++) {
                int ix = px + (i / 3 - 1), iy = py + (i % 3 - 1);
                if(ix < 0 || ix >= W || iy < 0 || iy >= H) continue;
                vec3 r = RGB2YCoCgR(rd3(colorB, (long long)iy * W + ix));
                nv++;
                mu = mu + r;
                var = var + r * r;
            }
            mu = mu / (float)nv;
            var = vmax(vec3(0.0f), var / (float)nv - mu * mu);
            vec3 sigma(std::sqrt(var.x), std::sqrt(var.y), std::sqrt(var.z));
            float lu = su - motionB[pixelPtr * 2], lv = sv - motionB[pixelPtr * 2 + 1];
            // interpolate(last_uv, pixelPtr, mu, sigma, lastcolor)
            int lbx = to_int(lu * (float)W - 0.5f), lby = to_int(lv * (float)H - 0.5f);
            float pointx = lu * (float)W, pointy = lv * (float)H;
            vec3 outcolor(0.0f);
            float tot = 0.0f;
            for(int i = 0; i < 4; i++) {
                long long ix = (long long)lbx + i / 2, iy = (long long)lby + i % 2;
                if(ix < 0 || ix >= W || iy < 0 || iy >= H) continue;
                long long ptr = iy * W + ix;
                if(!same_surface(normal, inst, h_normal, h_inst, pixelPtr, ptr)) continue;
                float dx = std::fabs(pointx - (float)ix - 0.5f), dy = std::fabs(pointy - (float)iy - 0.5f);
                float weight = std::fmax(0.0f, (1.0f - dx) * (1.0f - dy));
                vec3 lastcolor = RGB2YCoCgR(rd3(h_color, ptr));
                vec3 dir = lastcolor - mu;
                float milen = 1.0f;
                milen = std::fmin(milen, sigma.x / std::fabs(dir.x));
                milen = std::fmin(milen, sigma.y / std::fabs(dir.y));
                milen = std::fmin(milen, sigma.z / std::fabs(dir.z));
                lastcolor = YCoCgR2RGB(mu + dir * milen);
                outcolor = outcolor + lastcolor * weight;
                tot += weight;
            }
            vec3 lastcolor = outcolor / tot;
            vec3 result = tot > 0.00001f ? mixv(lastcolor, cur, 0.05f) : cur;
            if(result.x < 0.0f || result.y < 0.0f || result.z < 0.0f) result = vec3(0.0f);
            if(any_nan(result)) result = vec3(0.0f);
            wr3(out, pixelPtr, result);
        }
}

}  // extern "C"
