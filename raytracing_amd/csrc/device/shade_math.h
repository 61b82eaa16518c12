// sin, cos, atan, asin and pow for the shading code, built from IEEE fp32 +, -, *, /,
// sqrt, floor and integer bit operations only.
//
// GLSL gives these functions an error bound, not a value (and real drivers use much
// coarser approximations than a C math library), so the reference's result is
// vendor-defined at the last bits.  Fixing ONE recipe here and in the CPU checker makes
// the two agree bit for bit; this matters because the reference's GGX lobe at its minimum
// roughness 0.001 is so ill-conditioned in fp32 (1 - cos^2 with cos one ulp below 1) that a
// one-ulp difference in sin/cos moves a pixel by 10-30 %.
// Absolute error of each recipe is about 1e-7 (sin, cos on |x| <= 8), 2e-7 (atan, asin) and
// 2e-6 relative (pow on (0, 1] bases), far inside what GLSL allows.
#pragma once
#include <hip/hip_runtime.h>

namespace rta {

// Argument reduction to [-pi/4, pi/4] by multiples of pi/2 in three pieces
// (1.5703125 is exact in 8 bits, so k * hi is exact for every k that occurs).
__device__ __forceinline__ void sincos_reduce(float x, float &r, int &q) {
    float kf = floorf(x * 0.636619772367581343f + 0.5f);
    r = ((x - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.549789948768648e-8f;
    q = (int)kf & 3;
}
__device__ __forceinline__ float sin_poly(float r) {
    float z = r * r;
    return r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
}
__device__ __forceinline__ float cos_poly(float r) {
    float z = r * r;
    return (1.0f - 0.5f * z) + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
}
__device__ __forceinline__ float rt_sin(float x) {
    float r;
    int q;
    sincos_reduce(x, r, q);
    float s = sin_poly(r), c = cos_poly(r);
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
__device__ __forceinline__ float rt_cos(float x) {
    float r;
    int q;
    sincos_reduce(x, r, q);
    float s = sin_poly(r), c = cos_poly(r);
    float v = (q & 1) ? s : c;
    return ((q + 1) & 2) ? -v : v;
}

// atan on [0, inf) by the classic three-interval reduction, then atan2 by quadrant.
__device__ __forceinline__ float atan_pos(float t) {
    float y0, z;
    if(t > 2.414213562373095f) {
        y0 = 1.5707963267948966f;
        z = -1.0f / t;
    } else if(t > 0.4142135623730950f) {
        y0 = 0.7853981633974483f;
        z = (t - 1.0f) / (t + 1.0f);
    } else {
        y0 = 0.0f;
        z = t;
    }
    float zz = z * z;
    float p = (((8.05374449538e-2f * zz - 1.38776856032e-1f) * zz + 1.99777106478e-1f) * zz - 3.33329491539e-1f) * zz * z + z;
    return y0 + p;
}
__device__ __forceinline__ float rt_atan2(float y, float x) {
    const float pi = 3.14159265358979323846f;
    if(x == 0.0f) {
        if(y > 0.0f) return 1.5707963267948966f;
        if(y < 0.0f) return -1.5707963267948966f;
        return 0.0f;
    }
    float a = atan_pos(fabsf(y / x));
    if(x < 0.0f) a = pi - a;
    return y < 0.0f ? -a : a;
}

__device__ __forceinline__ float rt_asin(float x) {
    float a = fabsf(x);
    if(a > 1.0f) a = 1.0f;  // the only callers pass components of unit vectors
    bool big = a > 0.5f;
    float z, s;
    if(big) {
        z = 0.5f * (1.0f - a);
        s = sqrtf(z);
    } else {
        z = a * a;
        s = a;
    }
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * s + s;
    if(big) p = 1.5707963267948966f - (p + p);
    return x < 0.0f ? -p : p;
}

// pow(x, n) = 2^(n * log2 x) for x > 0 (0 for x <= 0: the callers raise texel values in
// [0, 1] to 2.2).  log2: mantissa in [sqrt(1/2), sqrt(2)), log((1+s)/(1-s)) series;
// exp2: integer part into the exponent field, degree-6 polynomial on [-1/2, 1/2].
__device__ __forceinline__ float rt_log2(float x) {
    unsigned ix = __float_as_uint(x);
    int e = (int)(ix >> 23) - 127;
    unsigned m = (ix & 0x007FFFFFu) | 0x3F800000u;
    float f = __uint_as_float(m);  // [1, 2)
    if(f > 1.41421356237f) {
        f = f * 0.5f;
        e += 1;
    }
    float s = (f - 1.0f) / (f + 1.0f);
    float z = s * s;
    float p = s * (2.0f + z * (0.666666666666667f + z * (0.4f + z * (0.285714285714286f + z * 0.222222222222222f))));
    return (float)e + p * 1.44269504088896341f;
}
__device__ __forceinline__ float rt_exp2(float x) {
    if(x < -126.0f) return 0.0f;
    if(x > 127.0f) x = 127.0f;
    float kf = floorf(x + 0.5f);
    float f = x - kf;  // [-1/2, 1/2]
    float t = f * 0.693147180559945309f;
    float p = 1.0f + t * (1.0f + t * (0.5f + t * (0.166666666666667f + t * (4.16666666666667e-2f + t * (8.33333333333333e-3f + t * 1.38888888888889e-3f)))));
    int k = (int)kf;
    return p * __uint_as_float((unsigned)(k + 127) << 23);
}
__device__ __forceinline__ float rt_pow(float x, float n) {
    if(!(x > 0.0f)) return 0.0f;
    return rt_exp2(n * rt_log2(x));
}

}  // namespace rta
