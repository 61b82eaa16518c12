// TEST INFRASTRUCTURE ONLY.  The one fixed recipe for sin / cos / atan / asin / pow that the CPU
// restatement (pt_oracle.cpp) shares with the HIP path (raytracing_amd/csrc/device/shade_math.h),
// kept in a header of its own so that oracle/glsl_frame_harness.cpp can also run the reference's
// shader text with these very functions in place of the C library's: with the transcendentals
// equal on both sides, every remaining difference would be a difference of transcription.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc_math {

// ---------------------------------------------------------------- sin / cos / atan / asin / pow
// GLSL bounds the error of these functions but does not fix their values, so one recipe
// is fixed here (pt_oracle.h, conventions) from IEEE +, -, *, /, sqrt, floor and integer
// bit operations; the HIP path evaluates the same recipe and therefore the same bits.
inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline void sincos_reduce(float x, float &r, int &q) {
    float kf = std::floor(x * 0.636619772367581343f + 0.5f);
    r = ((x - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.549789948768648e-8f;
    q = (int)kf & 3;
}
inline float sin_poly(float r) {
    float z = r * r;
    return r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
}
inline float cos_poly(float r) {
    float z = r * r;
    return (1.0f - 0.5f * z) + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
}
inline float m_sin(float x) {
    float r; int q;
    sincos_reduce(x, r, q);
    float s = sin_poly(r), c = cos_poly(r);
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
inline float m_cos(float x) {
    float r; int q;
    sincos_reduce(x, r, q);
    float s = sin_poly(r), c = cos_poly(r);
    float v = (q & 1) ? s : c;
    return ((q + 1) & 2) ? -v : v;
}
inline float atan_pos(float t) {
    float y0, z;
    if(t > 2.414213562373095f) { y0 = 1.5707963267948966f; z = -1.0f / t; }
    else if(t > 0.4142135623730950f) { y0 = 0.7853981633974483f; z = (t - 1.0f) / (t + 1.0f); }
    else { y0 = 0.0f; z = t; }
    float zz = z * z;
    float p = (((8.05374449538e-2f * zz - 1.38776856032e-1f) * zz + 1.99777106478e-1f) * zz - 3.33329491539e-1f) * zz * z + z;
    return y0 + p;
}
inline float m_atan2(float y, float x) {
    const float pi = 3.14159265358979323846f;
    if(x == 0.0f) {
        if(y > 0.0f) return 1.5707963267948966f;
        if(y < 0.0f) return -1.5707963267948966f;
        return 0.0f;
    }
    float a = atan_pos(std::fabs(y / x));
    if(x < 0.0f) a = pi - a;
    return y < 0.0f ? -a : a;
}
inline float m_asin(float x) {
    float a = std::fabs(x);
    if(a > 1.0f) a = 1.0f;
    bool big = a > 0.5f;
    float z, s;
    if(big) { z = 0.5f * (1.0f - a); s = std::sqrt(z); }
    else { z = a * a; s = a; }
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * s + s;
    if(big) p = 1.5707963267948966f - (p + p);
    return x < 0.0f ? -p : p;
}
inline float m_log2(float x) {
    uint32_t ix = f2u(x);
    int e = (int)(ix >> 23) - 127;
    uint32_t m = (ix & 0x007FFFFFu) | 0x3F800000u;
    float f = u2f(m);
    if(f > 1.41421356237f) { f = f * 0.5f; e += 1; }
    float s = (f - 1.0f) / (f + 1.0f);
    float z = s * s;
    float p = s * (2.0f + z * (0.666666666666667f + z * (0.4f + z * (0.285714285714286f + z * 0.222222222222222f))));
    return (float)e + p * 1.44269504088896341f;
}
inline float m_exp2(float x) {
    if(x < -126.0f) return 0.0f;
    if(x > 127.0f) x = 127.0f;
    float kf = std::floor(x + 0.5f);
    float f = x - kf;
    float t = f * 0.693147180559945309f;
    float p = 1.0f + t * (1.0f + t * (0.5f + t * (0.166666666666667f + t * (4.16666666666667e-2f + t * (8.33333333333333e-3f + t * 1.38888888888889e-3f)))));
    int k = (int)kf;
    return p * u2f((uint32_t)(k + 127) << 23);
}
inline float m_pow(float x, float n) {
    if(!(x > 0.0f)) return 0.0f;
    return m_exp2(n * m_log2(x));
}

}  // namespace orc_math
