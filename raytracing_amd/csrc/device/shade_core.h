// Device-side restatement of everything the reference's fragment shader does at a path
// vertex: Sobol sampler, texture filtering, GGX conductor / dielectric BSDFs, sky lookup
// and sky importance sampling.  S/ = shader/ in the reference tree.
//
// Expressions are written with the operand order and association of the GLSL source and
// must stay unfused (-ffp-contract=off); the CPU checker evaluates the same expressions,
// and sin/cos/atan/asin/pow come from the fixed recipes of shade_math.h, so the two agree bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.h"
#include "shade_math.h"

namespace rta {

constexpr float S_INF = 1E18f;       // S/basic/math.glsl:1
constexpr float S_PI = 3.1415926f;   // :2 (truncated on purpose: it is the reference's value)
constexpr float S_INV_PI = 0.3183098f;
constexpr float S_EPS = 0.0001f;

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 v3(float a) { return v3(a, a, a); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator/(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
__device__ __forceinline__ V3 normalize(V3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
__device__ __forceinline__ float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
__device__ __forceinline__ float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }
__device__ __forceinline__ V3 mixv(V3 x, V3 y, float a) { return x * (1.0f - a) + y * a; }
__device__ __forceinline__ V3 reflect(V3 I, V3 N) { return I - (2.0f * dot(N, I)) * N; }
__device__ __forceinline__ V3 vmax(V3 a, V3 b) { return v3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
__device__ __forceinline__ bool any_nan(V3 v) { return isnan(v.x) || isnan(v.y) || isnan(v.z); }

// S/basic/math.glsl:6-32
__device__ __forceinline__ int roundint(float x) {
    float b = (0.0f < x) ? 0.0f : 1.0f;
    return (int)(x - b + 0.5f);
}
__device__ __forceinline__ float pow2(float x) { return x * x; }
__device__ __forceinline__ float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
__device__ __forceinline__ V3 powv(V3 v, float n) { return v3(rt_pow(v.x, n), rt_pow(v.y, n), rt_pow(v.z, n)); }
__device__ __forceinline__ V3 refract_custom(V3 I, V3 N, float eta) {
    float c1 = dot(N, -I);
    float s1 = sqrtf(1.0f - c1 * c1);
    float s2 = s1 / eta;
    if(s2 >= 1.0f) return v3(0.0f);
    float c2 = sqrtf(1.0f - s2 * s2);
    return -N * c2 + (I + N * c1) / eta;
}

// ---- Sobol sampler, S/basic/sobol.glsl:4-40 ---------------------------------------------
__device__ __forceinline__ unsigned wang_hash(unsigned seed) {
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}

// The sampler's value for dimension d and index i is the XOR of the direction numbers V[d][j]
// over the set bits j of i (S/basic/sobol.glsl:6-14).  XOR is associative, so the same value
// comes out of eight table reads, one per 4-bit digit of i, from a table that holds the XOR of
// every subset of four consecutive direction numbers: T[d][k][v] = XOR of V[d][4k + b] over the
// bits b of v.  No loop over the bits, no divergence between lanes with different indices.
constexpr int SOBOL_DIMS = 20;
constexpr int SOBOL_NIBBLE_WORDS = SOBOL_DIMS * 8 * 16;  // 2 560 words = 10 KiB
struct SobolNibbleTable {
    unsigned t[SOBOL_NIBBLE_WORDS];
};
constexpr SobolNibbleTable make_sobol_nibble_table() {
    constexpr unsigned V[SOBOL_DIMS * 32] = {
#include "sobol_table.inc"
    };
    SobolNibbleTable r{};
    for(int d = 0; d < SOBOL_DIMS; d++)
        for(int k = 0; k < 8; k++)
            for(int v = 0; v < 16; v++) {
                unsigned x = 0;
                for(int b = 0; b < 4; b++)
                    if((v >> b) & 1) x ^= V[d * 32 + 4 * k + b];
                r.t[(d * 8 + k) * 16 + v] = x;
            }
    return r;
}
static __device__ const SobolNibbleTable g_sobol_nibbles = make_sobol_nibble_table();

// every thread of the block calls this once, before the first rand()
__device__ __forceinline__ void load_sobol_to_lds(unsigned *s_T) {
    const uint4 *src = reinterpret_cast<const uint4 *>(g_sobol_nibbles.t);
    uint4 *dst = reinterpret_cast<uint4 *>(s_T);
    for(int k = threadIdx.x; k < SOBOL_NIBBLE_WORDS / 4; k += blockDim.x) dst[k] = src[k];
    __syncthreads();
}

struct Sampler {
    const unsigned *T;  // the nibble table (in LDS)
    unsigned seed, dim;
    // Dimensions 0..19 are the reference's.  Dimension d >= 20 (which the reference reads
    // out of bounds) reuses table row 1 + (d-1) % 19 with the index replaced by
    // wang_hash(index + 0x9E3779B9 * ((d-1) / 19)).
    __device__ float rand() {
        unsigned d = dim++, i = seed;
        if(d >= 20u) {
            unsigned fold = (d - 1u) / 19u;
            d = 1u + (d - 1u) % 19u;
            i = wang_hash(i + 0x9E3779B9u * fold);
        }
        const unsigned *row = T + d * 128u;
        const unsigned r = (row[i & 15u] ^ row[16u + ((i >> 4) & 15u)]) ^ (row[32u + ((i >> 8) & 15u)] ^ row[48u + ((i >> 12) & 15u)]) ^
                           (row[64u + ((i >> 16) & 15u)] ^ row[80u + ((i >> 20) & 15u)]) ^
                           (row[96u + ((i >> 24) & 15u)] ^ row[112u + (i >> 28)]);
        return (float)r * 2.3283064365386963e-10f;  // 1.0 / float(0xFFFFFFFFu) = 2^-32
    }
};

// ---- texture(), GL 4.6 core 8.14: LINEAR filter, REPEAT wrap, base level only -----------
__device__ __forceinline__ int wrap_texel(float f, int n) {
    float c = fminf(fmaxf(f, -1073741824.0f), 1073741824.0f);
    int m = (int)c % n;
    return m < 0 ? m + n : m;
}

// One texel = ONE load instruction wherever the format allows: a gather costs the memory pipeline a pass per distinct
// cache line among the 64 lanes whatever its width, so three dword loads of one 12-byte texel cost three times one
// 12-byte load.  (Global loads wider than a dword need dword alignment only; a dword load needs none.)
struct __attribute__((packed, aligned(4))) PackedF3 { float x, y, z; };
struct __attribute__((packed, aligned(4))) PackedF4 { float x, y, z, w; };

__device__ __forceinline__ V3 fetch_texel_rgb(const TexDesc &tx, int x, int y) {
    size_t base = ((size_t)y * tx.width + x) * tx.channels;
    float c[3] = {0.0f, 0.0f, 0.0f};
    if(tx.format == 0 && tx.channels == 4) {
        // RGBA8: the texel is one aligned word
        const unsigned w = static_cast<const unsigned *>(tx.data)[(size_t)y * tx.width + x];
        c[0] = (float)(w & 0xFFu) / 255.0f;
        c[1] = (float)((w >> 8) & 0xFFu) / 255.0f;
        c[2] = (float)((w >> 16) & 0xFFu) / 255.0f;
#ifndef RT_PACKED_LOADS
#define RT_PACKED_LOADS 1
#endif
    } else if(RT_PACKED_LOADS && tx.format == 0 && tx.channels == 3) {
        // RGB8: three bytes at any alignment, read as one (unaligned) word; the fourth byte is the next texel's, or
        // the padding pt_tex_create leaves behind the last one
        unsigned w;
        __builtin_memcpy(&w, static_cast<const unsigned char *>(tx.data) + base, 4);
        c[0] = (float)(w & 0xFFu) / 255.0f;
        c[1] = (float)((w >> 8) & 0xFFu) / 255.0f;
        c[2] = (float)((w >> 16) & 0xFFu) / 255.0f;
    } else if(RT_PACKED_LOADS && tx.format == 1 && tx.channels == 3) {
        const PackedF3 t = *reinterpret_cast<const PackedF3 *>(static_cast<const float *>(tx.data) + base);
        c[0] = t.x; c[1] = t.y; c[2] = t.z;
    } else if(tx.format == 0) {
        const unsigned char *p = static_cast<const unsigned char *>(tx.data) + base;
        for(int k = 0; k < 3; k++)
            if(k < tx.channels) c[k] = (float)p[k] / 255.0f;
    } else {
        const float *p = static_cast<const float *>(tx.data) + base;
        for(int k = 0; k < 3; k++)
            if(k < tx.channels) c[k] = p[k];
    }
    return v3(c[0], c[1], c[2]);
}

__device__ __forceinline__ V3 texture_rgb(const TexDesc &tx, float s, float t) {
    float u = s * (float)tx.width - 0.5f;
    float v = t * (float)tx.height - 0.5f;
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    int i0 = wrap_texel(fu, tx.width), j0 = wrap_texel(fv, tx.height);
    int i1 = i0 + 1 == tx.width ? 0 : i0 + 1;
    int j1 = j0 + 1 == tx.height ? 0 : j0 + 1;
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    V3 t00 = fetch_texel_rgb(tx, i0, j0), t10 = fetch_texel_rgb(tx, i1, j0);
    V3 t01 = fetch_texel_rgb(tx, i0, j1), t11 = fetch_texel_rgb(tx, i1, j1);
    return ((w00 * t00 + w10 * t10) + w01 * t01) + w11 * t11;
}

// NEAREST filter, REPEAT wrap: red channel of texel floor(s * w), floor(t * h)
__device__ __forceinline__ float texture_nearest_r(const TexDesc &tx, float s, float t) {
    const int i = wrap_texel(floorf(s * (float)tx.width), tx.width), j = wrap_texel(floorf(t * (float)tx.height), tx.height);
    return fetch_texel_rgb(tx, i, j).x;
}
// int(x) of GLSL is undefined outside the int range and for NaN: saturate, NaN -> 0
__device__ __forceinline__ int float_to_int(float x) {
    if(x != x) return 0;
    if(x >= 2147483520.0f) return 2147483647;
    if(x <= -2147483648.0f) return (int)0x80000000;
    return (int)x;
}

// ---- materials ----------------------------------------------------------------------------
struct BSDFQueryRecord {  // S/materials/materials.glsl:62-71
    V3 wi, wo;
    int mptr;
    float uvx, uvy;
    float eta;
};

struct ShadeCtx {
    const float *materials;
    const TexDesc *textures;
    int n_textures;
    TexDesc sky, sky_cache;
    int sky_w, sky_h;
    float sky_light_sum;
    Sampler rng;

    // The material record of the vertex being shaded, read once (bind_material): every
    // query of a vertex names the same record and the same uv, and the reference's per-call
    // texture fetch + pow (RoughConductor.glsl:3-32) is a pure function of them.
    float m_type = 0.0f, m_alpha = 0.0f, m_metallic = 0.0f, m_eta = 0.0f;
    V3 m_albedo;

    __device__ void bind_material(int mptr, float uvx, float uvy) {
        // seven floats in two loads (the record sits at any multiple of four bytes)
        const PackedF4 ma = *reinterpret_cast<const PackedF4 *>(materials + mptr);
        const PackedF3 mb = *reinterpret_cast<const PackedF3 *>(materials + mptr + 4);
        m_type = ma.x;
        m_albedo = v3(ma.y, ma.z, ma.w);
        m_alpha = mb.x;
        m_metallic = mb.y;
        const int albedo_map = roundint(mb.z);
        m_eta = roundint(m_type) == 2 ? materials[mptr + 7] : 0.0f;  // (conductor records end after 7 floats)
        if(albedo_map >= 0 && albedo_map < n_textures) m_albedo = powv(texture_rgb(textures[albedo_map], uvx, uvy), 2.2f);
    }
    __device__ float mat(int i) const { return i == 0 ? m_type : i == 4 ? m_alpha : i == 5 ? m_metallic : m_eta; }
    __device__ V3 base_albedo(const BSDFQueryRecord &) const { return m_albedo; }

    // S/materials/materials.glsl:5-31
    __device__ V3 sample_GGX(float alpha, float &pdf) {
        float alpha2 = alpha * alpha;
        float x = rng.rand();
        float cos_theta = sqrtf((1.0f - x) / (x * (alpha2 - 1.0f) + 1.0f));
        float cos2 = cos_theta * cos_theta;
        float phi = rng.rand() * 2.0f * S_PI;
        float r = sqrtf(1.0f - cos2);
        pdf = 2.0f * alpha2 * cos_theta / pow2(cos2 * (alpha2 - 1.0f) + 1.0f) / (2.0f * S_PI);
        return normalize(v3(r * rt_cos(phi), r * rt_sin(phi), cos_theta));
    }
    __device__ static float eval_GGX(float alpha, V3 h) {
        float cos_theta = fabsf(h.z);
        float alpha2 = alpha * alpha;
        float cos2 = cos_theta * cos_theta;
        return alpha2 * cos_theta / pow2(cos2 * (alpha2 - 1.0f) + 1.0f) / S_PI;
    }
    __device__ static float G_GGX(float alpha, float cosI) {
        float k = pow2(alpha + 1.0f) / 8.0f;
        return cosI / (cosI * (1.0f - k) + k);
    }
    // S/materials/materials.glsl:43-59
    __device__ static V3 SchlickFresnel(V3 f0, float c) { return f0 + (v3(1.0f) - f0) * pow5(1.0f - c); }
    __device__ static float fresnel(float cosI, float eta) {
        cosI = fabsf(cosI);
        float sinT = 1.0f / eta * sqrtf(1.0f - cosI * cosI);
        if(sinT > 1.0f) return 1.0f;
        float cosT = sqrtf(1.0f - sinT * sinT);
        float Rs = pow2((cosI - eta * cosT) / (cosI + eta * cosT));
        float Rp = pow2((cosT - eta * cosI) / (cosT + eta * cosI));
        return (Rs + Rp) / 2.0f;
    }

    // S/materials/RoughConductor.glsl:3-59
    __device__ V3 eval_RoughConductor(const BSDFQueryRecord &b) const {
        if(signf(b.wi.z) != signf(b.wo.z)) return v3(0.0f);
        V3 albedo = base_albedo(b);
        float alpha = mat(4);
        float metallic = mat(5);
        V3 F0 = mixv(v3(0.04f), albedo, metallic);
        V3 H = normalize(b.wi + b.wo);
        H = H * signf(H.z);
        float OdotH = fabsf(dot(b.wo, H));
        float OdotN = fabsf(b.wo.z), IdotN = fabsf(b.wi.z);
        float D = eval_GGX(alpha, H);
        V3 F = SchlickFresnel(F0, OdotH);
        float k = pow2(alpha + 1.0f) / 8.0f;
        float Gi4io = 0.25f / ((IdotN * (1.0f - k) + k) * (OdotN * (1.0f - k) + k));
        float FL = pow5(1.0f - IdotN), FV = pow5(1.0f - OdotN);
        float Fd90 = 0.5f + 2.0f * pow2(OdotH) * alpha;
        float Fd = mixf(1.0f, Fd90, FL) * mixf(1.0f, Fd90, FV);
        return D * F * Gi4io + (1.0f - metallic) * albedo * Fd * S_INV_PI;
    }
    __device__ float pdf_RoughConductor(const BSDFQueryRecord &b) const {
        if(signf(b.wi.z) != signf(b.wo.z)) return 0.0f;
        float alpha = mat(4);
        V3 H = normalize(b.wi + b.wo);
        H = H * signf(H.z);
        float OdotH = fabsf(dot(b.wo, H));
        float pdf = eval_GGX(alpha, H);
        return pdf / (4.0f * OdotH);
    }
    __device__ V3 sample_RoughConductor(BSDFQueryRecord &b, float &pdf) {
        float alpha = mat(4);
        V3 H = sample_GGX(alpha, pdf);
        b.wo = reflect(-b.wi, H);
        b.eta = 1.0f;
        if(signf(b.wi.z) != signf(b.wo.z)) {
            pdf = -1.0f;
            return v3(0.0f);
        }
        float OdotH = fabsf(dot(b.wo, H));
        pdf /= (4.0f * OdotH);
        return eval_RoughConductor(b);
    }

    // S/materials/RoughDielectric.glsl:7-133
    __device__ V3 eval_RoughDielectric(const BSDFQueryRecord &b) const {
        V3 albedo = base_albedo(b);
        float alpha = mat(4);
        float eta = mat(7);
        if(b.wi.z < 0.0f) eta = 1.0f / eta;
        bool isreflect = signf(b.wi.z) == signf(b.wo.z);
        V3 H;
        if(isreflect) H = normalize(b.wi + b.wo);
        else H = -normalize(b.wi + eta * b.wo);
        float D = eval_GGX(alpha, H);
        float F = fresnel(dot(b.wi, H), eta);
        float G = G_GGX(alpha, fabsf(b.wi.z)) * G_GGX(alpha, fabsf(b.wo.z));
        if(signf(dot(b.wi, H)) != signf(b.wi.z) || signf(dot(b.wo, H)) != signf(b.wo.z)) G = 0.0f;
        if(isreflect) {
            return albedo * F * D * G / fabsf(4.0f * b.wi.z * b.wo.z);
        } else {
            float cosIH = dot(b.wi, H);
            float cosOH = dot(b.wo, H);
            float deno = pow2(cosIH + eta * cosOH);
            float value = ((1.0f - F) * D * G * cosIH * cosOH) / (b.wi.z * b.wo.z * deno);
            return albedo * fabsf(value);
        }
    }
    __device__ float pdf_RoughDielectric(const BSDFQueryRecord &b) const {
        float alpha = mat(4);
        float eta = mat(7);
        if(b.wi.z < 0.0f) eta = 1.0f / eta;
        bool isreflect = signf(b.wi.z) == signf(b.wo.z);
        V3 H;
        float dwh_dwo;
        if(isreflect) {
            H = normalize(b.wi + b.wo);
            dwh_dwo = 1.0f / fabsf(4.0f * dot(b.wi, H));
        } else {
            H = -normalize(b.wi + eta * b.wo);
            float deno = pow2(dot(b.wi, H) + eta * dot(b.wo, H));
            dwh_dwo = eta * eta * fabsf(dot(b.wo, H)) / deno;
        }
        float F = fresnel(dot(b.wi, H), eta);
        float pdf = eval_GGX(alpha, H);
        if(isreflect) pdf *= F;
        else pdf *= 1.0f - F;
        return pdf * dwh_dwo;
    }
    __device__ V3 sample_RoughDielectric(BSDFQueryRecord &b, float &pdf) {
        V3 albedo = base_albedo(b);
        float alpha = mat(4);
        float eta = mat(7);
        if(b.wi.z < 0.0f) eta = 1.0f / eta;
        pdf = -1.0f;
        float pdfH;
        V3 H = sample_GGX(alpha, pdfH);
        float F = fresnel(dot(b.wi, H), eta);
        float D = eval_GGX(alpha, H);
        if(rng.rand() < F) {
            b.eta = 1.0f;
            b.wo = reflect(-b.wi, H);
            if(signf(b.wi.z) != signf(b.wo.z)) return v3(0.0f);
            float G = G_GGX(alpha, fabsf(b.wi.z)) * G_GGX(alpha, fabsf(b.wo.z));
            if(signf(dot(b.wi, H)) != signf(b.wi.z) || signf(dot(b.wo, H)) != signf(b.wo.z)) G = 0.0f;
            pdf = F * pdfH / fabsf(4.0f * dot(b.wi, H));
            V3 r = albedo * F * D * G / (4.0f * b.wi.z * b.wo.z);
            return v3(fabsf(r.x), fabsf(r.y), fabsf(r.z));
        } else {
            b.eta = eta;
            b.wo = refract_custom(-b.wi, H * signf(dot(H, b.wi)), eta);
            if(signf(b.wi.z) == signf(b.wo.z)) return v3(0.0f);
            float G = G_GGX(alpha, fabsf(b.wi.z)) * G_GGX(alpha, fabsf(b.wo.z));
            if(signf(dot(b.wi, H)) != signf(b.wi.z) || signf(dot(b.wo, H)) != signf(b.wo.z)) G = 0.0f;
            float deno = pow2(dot(b.wi, H) + eta * dot(b.wo, H));
            float value = ((1.0f - F) * D * G * dot(b.wi, H) * dot(b.wo, H)) / (b.wi.z * b.wo.z * deno);
            pdf = (1.0f - F) * pdfH * eta * eta * fabsf(dot(b.wo, H)) / deno;
            return albedo * fabsf(value);
        }
    }

    // S/materials/materials.glsl:82-111
    __device__ V3 eval_material(const BSDFQueryRecord &b) const {
        int type = roundint(mat(0));
        if(type == 1) return eval_RoughConductor(b);
        else if(type == 2) return eval_RoughDielectric(b);
        return v3(0.0f);
    }
    __device__ float pdf_material(const BSDFQueryRecord &b) const {
        int type = roundint(mat(0));
        if(type == 1) return pdf_RoughConductor(b);
        else if(type == 2) return pdf_RoughDielectric(b);
        return 0.0f;
    }
    __device__ V3 sample_material(BSDFQueryRecord &b, float &pdf) {
        int type = roundint(mat(0));
        if(type == 1) return sample_RoughConductor(b, pdf);
        else if(type == 2) return sample_RoughDielectric(b, pdf);
        pdf = -1.0f;
        return v3(0.0f);
    }
    __device__ V3 albedo_material(const BSDFQueryRecord &b) const {
        int type = roundint(mat(0));
        if(type == 1 || type == 2) return base_albedo(b);
        return v3(0.0f);
    }

    // ---- sky, S/pathtracing.glsl:358-399
    __device__ V3 spherical_sample(float &pdf) {
        float z = rng.rand() * 2.0f - 1.0f;
        float r = fmaxf(0.0f, sqrtf(1.0f - z * z));
        float phi = 2.0f * S_PI * rng.rand();
        pdf = 0.25f * S_INV_PI;
        return v3(r * rt_cos(phi), r * rt_sin(phi), z);
    }
    __device__ V3 get_background_color(V3 v) const {
        float ux = rt_atan2(v.z, v.x), uy = rt_asin(v.y);
        ux /= 2.0f * S_PI;
        uy /= S_PI;
        ux += 0.5f;
        uy += 0.5f;
        uy = 1.0f - uy;
        return texture_rgb(sky, ux, uy);
    }
    __device__ float w2a_of(float y) const {
        float w2a = (2.0f * S_PI * S_PI * sqrtf(1.0f - y * y)) / (float)(sky_h * sky_w);
        if(w2a == 0.0f) w2a = (float)sky_w * S_PI / (float)sky_h;
        return w2a;
    }
    __device__ V3 skybox_sample(float &pdf) {
        float x = rng.rand();
        float y = rng.rand();
        V3 samp = texture_rgb(sky_cache, x, y);
        float theta = (samp.x - 0.5f) * 2.0f * S_PI;
        float phi = (1.0f - samp.y - 0.5f) * S_PI;
        float r = rt_cos(phi);
        V3 wi = v3(r * rt_cos(theta), rt_sin(phi), r * rt_sin(theta));
        pdf = samp.z / w2a_of(wi.y);
        return wi;
    }
    __device__ float skybox_sample_pdf(V3 v) const {
        V3 l = get_background_color(v);
        float lw = (l.x * 0.2f + l.y * 0.7f + l.z * 0.1f) / sky_light_sum;
        return lw / w2a_of(v.y);
    }
};

// S/pathtracing.glsl:148-166
struct Frame {
    V3 s, t, n;
};
__device__ __forceinline__ Frame create_coord(V3 n) {
    Frame f;
    if(fabsf(n.z) > 1.0f - S_EPS) f.s = normalize(cross(n, v3(1.0f, 0.0f, 0.0f)));
    else f.s = normalize(cross(n, v3(0.0f, 0.0f, 1.0f)));
    f.t = cross(n, f.s);
    f.n = n;
    return f;
}
__device__ __forceinline__ V3 to_world(const Frame &f, V3 v) { return v.x * f.s + v.y * f.t + v.z * f.n; }
__device__ __forceinline__ V3 to_local(const Frame &f, V3 v) { return v3(dot(v, f.s), dot(v, f.t), dot(v, f.n)); }

}  // namespace rta
