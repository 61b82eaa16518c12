// The reference's denoising chain behind the path tracer: SVGF temporal accumulation
// (shader/postprocessing/SVGF_TemporalFilter.glsl:124-200), variance estimation
// (SVGF_VarianceFilter.glsl:30-103), a-trous spatial filter (SVGF_SpatialFilter.glsl:32-90) and
// TAA with variance clipping in YCoCg-R (TAA.glsl:40-190).  One thread per pixel; the stencils
// (3x3 .. 7x7, dilated 5x5) read neighbours straight from global memory, which the L2 serves:
// 12-20 B per tap, all taps of a wave within a few rows.
//
// Conventions where GLSL leaves the result open (the oracle uses the same): min/max drop a NaN
// operand, int(x) saturates (NaN -> 0), exp follows the recipe of shade_math.h, pow(x, 128) is seven squarings.
#include "kernels.h"
#include "shade_core.h"
#include "shade_math.h"

namespace rta {
namespace {

constexpr int DN_BLOCK = 256;

__device__ __forceinline__ float rt_exp(float x) { return rt_exp2(x * 1.44269504088896341f); }
// pow(x, 128) for x >= 0 as seven squarings (GLSL bounds pow's error through exp2(y * log2(x)); this is tighter and
// a tenth of the instructions; the oracle squares the same way)
__device__ __forceinline__ float pow128(float x) {
    x = x * x; x = x * x; x = x * x; x = x * x; x = x * x; x = x * x;
    return x * x;
}
__device__ __forceinline__ V3 load3(const float *p, int ptr) { return v3(p[(size_t)ptr * 3], p[(size_t)ptr * 3 + 1], p[(size_t)ptr * 3 + 2]); }
__device__ __forceinline__ float lum709(V3 c) { return (c.x * 0.212671f + c.y * 0.715160f) + c.z * 0.072169f; }
__device__ __forceinline__ float mix1(float x, float y, float a) { return x * (1.0f - a) + y * a; }

// geometry_test of both shaders: same surface in this frame and in the history
__device__ __forceinline__ bool geometry_test(const float *normal, const float *inst, const float *h_normal,
                                              const float *h_inst, int ptr1, int ptr2) {
    const V3 n1 = load3(normal, ptr1), n2 = load3(h_normal, ptr2);
    return dot(n1, n2) > 0.9f && fabsf(inst[ptr1] - h_inst[ptr2]) < 0.1f;
}

struct TemporalArgs {
    int W, H, first_frame;
    const float *color, *normal, *inst, *motion;
    float *o_color, *o_moment, *o_num;
    const float *h_color, *h_moment, *h_num, *h_normal, *h_inst;
};

__global__ __launch_bounds__(DN_BLOCK) void svgf_temporal_kernel(TemporalArgs a) {
    const int pixelPtr = blockIdx.x * blockDim.x + threadIdx.x;
    if(pixelPtr >= a.W * a.H) return;
    const int px = pixelPtr % a.W, py = pixelPtr / a.W;
    const float fW = (float)a.W, fH = (float)a.H;
    V3 color = load3(a.color, pixelPtr);
    const float lum = lum709(color);
    float mom_x = lum, mom_y = lum * lum;
    if(a.first_frame) {
        a.o_color[(size_t)pixelPtr * 3] = color.x; a.o_color[(size_t)pixelPtr * 3 + 1] = color.y; a.o_color[(size_t)pixelPtr * 3 + 2] = color.z;
        a.o_moment[(size_t)pixelPtr * 2] = mom_x; a.o_moment[(size_t)pixelPtr * 2 + 1] = mom_y;
        a.o_num[pixelPtr] = 1.0f;
        return;
    }
    // variance guide (RGB, HDR), :139-156
    V3 mu = v3(0.0f), var = v3(0.0f);
    int nv = 0;
    for(int i = 0; i < 9; i++) {
        const int x = px + (i / 3 - 1), y = py + (i % 3 - 1);
        if(x < 0 || x >= a.W || y < 0 || y >= a.H) continue;
        const V3 r = load3(a.color, y * a.W + x);
        nv++;
        mu = mu + r;
        var = var + r * r;
    }
    mu = mu / (float)nv;
    const V3 vv = var / (float)nv - mu * mu;
    var = v3(fmaxf(0.0f, vv.x), fmaxf(0.0f, vv.y), fmaxf(0.0f, vv.z));
    const V3 sigma = v3(sqrtf(var.x) * 5.0f, sqrtf(var.y) * 5.0f, sqrtf(var.z) * 5.0f);

    const float suv_x = ((float)px + 0.5f) / fW, suv_y = ((float)py + 0.5f) / fH;
    const float lu = suv_x - a.motion[(size_t)pixelPtr * 2], lv = suv_y - a.motion[(size_t)pixelPtr * 2 + 1];
    if(lu >= 1.0f || lv >= 1.0f || lu < 0.0f || lv < 0.0f) {
        a.o_num[pixelPtr] = 1.0f;
    } else {
        const int lpx = float_to_int(lu * fW), lpy = float_to_int(lv * fH);
        // interpolate(), :68-104
        const int lbx = float_to_int(floorf(lu * fW - 0.5f)), lby = float_to_int(floorf(lv * fH - 0.5f));
        const float ptx = lu * fW, pty = lv * fH;
        V3 outcolor = v3(0.0f);
        float om_x = 0.0f, om_y = 0.0f, totweight = 0.0f;
        for(int i = 0; i < 4; i++) {
            const int ix = lbx + i / 2, iy = lby + i % 2;
            if(ix < 0 || ix >= a.W || iy < 0 || iy >= a.H) continue;
            const int ptr = iy * a.W + ix;
            if(!geometry_test(a.normal, a.inst, a.h_normal, a.h_inst, pixelPtr, ptr)) continue;
            const float dx = fabsf(ptx - (float)ix - 0.5f), dy = fabsf(pty - (float)iy - 0.5f);
            const float weight = fmaxf(0.0f, (1.0f - dx) * (1.0f - dy));
            const V3 lc = load3(a.h_color, ptr);
            outcolor = outcolor + lc * weight;
            om_x += a.h_moment[(size_t)ptr * 2] * weight;
            om_y += a.h_moment[(size_t)ptr * 2 + 1] * weight;
            totweight += weight;
        }
        const V3 lastcolor = outcolor / totweight;
        const float lm_x = om_x / totweight, lm_y = om_y / totweight;
        const bool valid = totweight > 0.00001f;
        const V3 dir = lastcolor - mu;
        // NaN in lu/lv passes the range test above; its pixel index (0 by the saturating conversion) stays in range
        if(!valid || fabsf(dir.x) > sigma.x || fabsf(dir.y) > sigma.y || fabsf(dir.z) > sigma.z) {
            a.o_num[pixelPtr] = 1.0f;
        } else {
            int lptr = lpy * a.W + lpx;
            if(lpx < 0 || lpx >= a.W || lpy < 0 || lpy >= a.H) lptr = pixelPtr;  // cannot happen for finite uv in [0,1)
            a.o_num[pixelPtr] = a.h_num[lptr] + 1.0f;
            color = v3(mix1(lastcolor.x, color.x, 0.2f), mix1(lastcolor.y, color.y, 0.2f), mix1(lastcolor.z, color.z, 0.2f));
            mom_x = mix1(lm_x, mom_x, 0.2f);
            mom_y = mix1(lm_y, mom_y, 0.2f);
        }
    }
    a.o_color[(size_t)pixelPtr * 3] = color.x; a.o_color[(size_t)pixelPtr * 3 + 1] = color.y; a.o_color[(size_t)pixelPtr * 3 + 2] = color.z;
    a.o_moment[(size_t)pixelPtr * 2] = mom_x; a.o_moment[(size_t)pixelPtr * 2 + 1] = mom_y;
}

__device__ __forceinline__ float tap_kernel(int k) {  // {3/8, 1/4, 1/16, 1/32}
    k = k < 0 ? -k : k;
    return k == 0 ? 0.375f : (k == 1 ? 0.25f : (k == 2 ? 0.0625f : 0.03125f));
}

__global__ __launch_bounds__(DN_BLOCK) void svgf_variance_kernel(int W, int H, const float *moment, const float *normal,
                                                                 const float *depth, const float *num, float *variance) {
    const int pixelPtr = blockIdx.x * blockDim.x + threadIdx.x;
    if(pixelPtr >= W * H) return;
    const int px = pixelPtr % W, py = pixelPtr / W;
    const V3 n = load3(normal, pixelPtr);
    float mx = moment[(size_t)pixelPtr * 2], my = moment[(size_t)pixelPtr * 2 + 1];
    const float ns = num[pixelPtr], d = depth[pixelPtr];
    float total = 0.0f, rx = 0.0f, ry = 0.0f;
    if(ns <= 4.0f) {  // short history: 7x7 edge-stopping estimate, :45-73
        for(int i = -3; i <= 3; i++)
            for(int j = -3; j <= 3; j++) {
                const int x = px + i, y = py + j;
                if(x >= W || x < 0 || y >= H || y < 0) continue;
                const int ptr = y * W + x;
                const V3 n2 = load3(normal, ptr);
                const float w = tap_kernel(i) * tap_kernel(j) * pow128(fmaxf(0.0f, dot(n, n2))) *
                                rt_exp(-fabsf(d - depth[ptr]) / 2.0f);
                total += w;
                rx += w * moment[(size_t)ptr * 2];
                ry += w * moment[(size_t)ptr * 2 + 1];
            }
    } else {  // 3x3 blur, :75-94
        for(int i = -1; i <= 1; i++)
            for(int j = -1; j <= 1; j++) {
                const int x = px + i, y = py + j;
                if(x >= W || x < 0 || y >= H || y < 0) continue;
                const int ptr = y * W + x;
                const float w = tap_kernel(i) * tap_kernel(j);
                total += w;
                rx += w * moment[(size_t)ptr * 2];
                ry += w * moment[(size_t)ptr * 2 + 1];
            }
    }
    mx = rx / total;
    my = ry / total;
    variance[pixelPtr] = fmaxf(0.0f, my - mx * mx);
}

__global__ __launch_bounds__(DN_BLOCK) void svgf_spatial_kernel(int W, int H, int step, const float *color, const float *normal,
                                                                const float *depth, const float *variance, float *out) {
    const int pixelPtr = blockIdx.x * blockDim.x + threadIdx.x;
    if(pixelPtr >= W * H) return;
    const int px = pixelPtr % W, py = pixelPtr / W;
    const V3 c = load3(color, pixelPtr), n = load3(normal, pixelPtr);
    const float d = depth[pixelPtr];
    const float sigma = sqrtf(variance[pixelPtr]);
    float total = 0.0f;
    V3 result = v3(0.0f);
    for(int i = -2; i <= 2; i++)
        for(int j = -2; j <= 2; j++) {
            const long long x = (long long)px + (long long)i * step, y = (long long)py + (long long)j * step;
            if(x >= W || x < 0 || y >= H || y < 0) continue;
            const int ptr = (int)(y * W + x);
            const V3 c2 = load3(color, ptr), n2 = load3(normal, ptr);
            const V3 dc = c - c2;
            const float len = sqrtf((dc.x * dc.x + dc.y * dc.y) + dc.z * dc.z);
            const float w = tap_kernel(i) * tap_kernel(j) * pow128(fmaxf(0.0f, dot(n, n2))) *
                            rt_exp(-fabsf(d - depth[ptr]) / 2.0f) * rt_exp(-fabsf(len) / (sigma * 4.0f + 0.00001f));
            total += w;
            result = result + w * c2;
        }
    result = result / total;
    const float inf = __uint_as_float(0x7F800000u);
    if(result.x != result.x || result.y != result.y || result.z != result.z || fabsf(result.x) == inf || fabsf(result.y) == inf ||
       fabsf(result.z) == inf)
        result = v3(10000.0f, 0.0f, 0.0f);
    out[(size_t)pixelPtr * 3] = result.x; out[(size_t)pixelPtr * 3 + 1] = result.y; out[(size_t)pixelPtr * 3 + 2] = result.z;
}

__device__ __forceinline__ V3 rgb2ycocgr(V3 c) {  // TAA.glsl:40-50
    V3 o;
    o.y = c.x - c.z;
    const float temp = c.z + o.y / 2.0f;
    o.z = c.y - temp;
    o.x = temp + o.z / 2.0f;
    return o;
}
__device__ __forceinline__ V3 ycocgr2rgb(V3 c) {  // :52-62
    V3 o;
    const float temp = c.x - c.z / 2.0f;
    o.y = c.z + temp;
    o.z = temp - c.y / 2.0f;
    o.x = o.z + c.y;
    return o;
}

struct TaaArgs {
    int W, H;
    const float *color, *h_color, *motion, *normal, *inst, *h_normal, *h_inst;
    float *out;
};

__global__ __launch_bounds__(DN_BLOCK) void taa_kernel(TaaArgs a) {
    const int pixelPtr = blockIdx.x * blockDim.x + threadIdx.x;
    if(pixelPtr >= a.W * a.H) return;
    const int px = pixelPtr % a.W, py = pixelPtr / a.W;
    const float fW = (float)a.W, fH = (float)a.H;
    const V3 cur = load3(a.color, pixelPtr);
    V3 mu = v3(0.0f), var = v3(0.0f);
    int nv = 0;
    for(int i = 0; i < 9; i++) {
        const int x = px + (i / 3 - 1), y = py + (i % 3 - 1);
        if(x < 0 || x >= a.W || y < 0 || y >= a.H) continue;
        const V3 r = rgb2ycocgr(load3(a.color, y * a.W + x));
        nv++;
        mu = mu + r;
        var = var + r * r;
    }
    mu = mu / (float)nv;
    const V3 vv = var / (float)nv - mu * mu;
    const V3 sigma = v3(sqrtf(fmaxf(0.0f, vv.x)), sqrtf(fmaxf(0.0f, vv.y)), sqrtf(fmaxf(0.0f, vv.z)));

    const float suv_x = ((float)px + 0.5f) / fW, suv_y = ((float)py + 0.5f) / fH;
    const float lu = suv_x - a.motion[(size_t)pixelPtr * 2], lv = suv_y - a.motion[(size_t)pixelPtr * 2 + 1];
    // interpolate(), :82-122 (int() truncates here, where the SVGF filter floors)
    const int lbx = float_to_int(lu * fW - 0.5f), lby = float_to_int(lv * fH - 0.5f);
    const float ptx = lu * fW, pty = lv * fH;
    V3 outcolor = v3(0.0f);
    float totweight = 0.0f;
    for(int i = 0; i < 4; i++) {
        const long long ix = (long long)lbx + i / 2, iy = (long long)lby + i % 2;
        if(ix < 0 || ix >= a.W || iy < 0 || iy >= a.H) continue;
        const int ptr = (int)(iy * a.W + ix);
        if(!geometry_test(a.normal, a.inst, a.h_normal, a.h_inst, pixelPtr, ptr)) continue;
        const float dx = fabsf(ptx - (float)ix - 0.5f), dy = fabsf(pty - (float)iy - 0.5f);
        const float weight = fmaxf(0.0f, (1.0f - dx) * (1.0f - dy));
        V3 lastcolor = rgb2ycocgr(load3(a.h_color, ptr));
        const V3 dir = lastcolor - mu;
        float milen = 1.0f;
        milen = fminf(milen, sigma.x / fabsf(dir.x));
        milen = fminf(milen, sigma.y / fabsf(dir.y));
        milen = fminf(milen, sigma.z / fabsf(dir.z));
        lastcolor = ycocgr2rgb(mu + dir * milen);
        outcolor = outcolor + lastcolor * weight;
        totweight += weight;
    }
    const V3 lastcolor = outcolor / totweight;
    V3 result = cur;
    if(totweight > 0.00001f)
        result = v3(mix1(lastcolor.x, cur.x, 0.05f), mix1(lastcolor.y, cur.y, 0.05f), mix1(lastcolor.z, cur.z, 0.05f));
    if(result.x < 0.0f || result.y < 0.0f || result.z < 0.0f) result = v3(0.0f);
    if(result.x != result.x || result.y != result.y || result.z != result.z) result = v3(0.0f);
    a.out[(size_t)pixelPtr * 3] = result.x; a.out[(size_t)pixelPtr * 3 + 1] = result.y; a.out[(size_t)pixelPtr * 3 + 2] = result.z;
}

inline dim3 grid_for(int npix) { return dim3((unsigned)((npix + DN_BLOCK - 1) / DN_BLOCK)); }

}  // namespace

void launch_svgf_temporal(hipStream_t s, int W, int H, int first_frame, const float *color, const float *normal, const float *inst,
                          const float *motion, float *o_color, float *o_moment, float *o_num, const float *h_color,
                          const float *h_moment, const float *h_num, const float *h_normal, const float *h_inst) {
    TemporalArgs a{W, H, first_frame, color, normal, inst, motion, o_color, o_moment, o_num, h_color, h_moment, h_num, h_normal, h_inst};
    hipLaunchKernelGGL(svgf_temporal_kernel, grid_for(W * H), dim3(DN_BLOCK), 0, s, a);
}
void launch_svgf_variance(hipStream_t s, int W, int H, const float *moment, const float *normal, const float *depth,
                          const float *num, float *variance) {
    hipLaunchKernelGGL(svgf_variance_kernel, grid_for(W * H), dim3(DN_BLOCK), 0, s, W, H, moment, normal, depth, num, variance);
}
void launch_svgf_spatial(hipStream_t s, int W, int H, int step, const float *color, const float *normal, const float *depth,
                         const float *variance, float *out) {
    hipLaunchKernelGGL(svgf_spatial_kernel, grid_for(W * H), dim3(DN_BLOCK), 0, s, W, H, step, color, normal, depth, variance, out);
}
void launch_taa(hipStream_t s, int W, int H, const float *color, const float *h_color, const float *motion, const float *normal,
                const float *inst, const float *h_normal, const float *h_inst, float *out) {
    TaaArgs a{W, H, color, h_color, motion, normal, inst, h_normal, h_inst, out};
    hipLaunchKernelGGL(taa_kernel, grid_for(W * H), dim3(DN_BLOCK), 0, s, a);
}

}  // namespace rta
