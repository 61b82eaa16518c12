// The three element-wise passes that turn the path tracer's outputs into a displayable
// image: merge (shader/postprocessing/SVGF_Merge.glsl:24-52), static blending across frames
// (StaticBlender.glsl:26-61) and ACES tone mapping + gamma (ToneMappingGamma.glsl:17-54).
// Streaming kernels, one thread per pixel, bound by HBM: 36 B in + 12 B out, 32 B in + 20 B
// out (+12 B in), 12 B in + 12 B out per pixel.
#include "kernels.h"
#include "shade_math.h"

namespace rta {
namespace {

constexpr int POST_BLOCK = 256;

__global__ __launch_bounds__(POST_BLOCK) void merge_kernel(int npix, const float *di, const float *idi,
                                                           const float *albedo, float *color) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= npix) return;
    const size_t o = (size_t)i * 3;
    float r[3];
    bool bad_nan = false, bad_inf = false;
    for(int k = 0; k < 3; k++) {
        r[k] = (di[o + k] + idi[o + k]) * albedo[o + k];
        bad_nan = bad_nan || r[k] != r[k];
    }
    if(bad_nan) r[0] = r[1] = r[2] = 10000.0f;
    for(int k = 0; k < 3; k++) bad_inf = bad_inf || fabsf(r[k]) == __uint_as_float(0x7F800000u);
    if(bad_inf) r[0] = r[1] = r[2] = 10000.0f;
    for(int k = 0; k < 3; k++) color[o + k] = r[k];
}

__global__ __launch_bounds__(POST_BLOCK) void tonemap_kernel(size_t n, const float *in, float *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    const float c = in[i];
    const float A = 2.51f, B = 0.03f, C = 2.43f, D = 0.59f, E = 0.14f;
    const float aces = (c * (A * c + B)) / (c * (C * c + D) + E);
    out[i] = rt_pow(aces, 0.45f);
}

__global__ __launch_bounds__(POST_BLOCK) void static_blend_kernel(int npix, unsigned frame, const float *color,
                                                                  float *hist_color, float *hist_moment) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= npix) return;
    const size_t o = (size_t)i * 3, m = (size_t)i * 2;
    const float cx = color[o], cy = color[o + 1], cz = color[o + 2];
    const float lum = (cx * 0.212671f + cy * 0.715160f) + cz * 0.072169f;
    const float mom_x = lum, mom_y = lum * lum;
    const float lx = hist_color[o], ly = hist_color[o + 1], lz = hist_color[o + 2];
    const float lmx = hist_moment[m], lmy = hist_moment[m + 1];
    const float sigma = sqrtf(fmaxf(1.0f, lmy - lmx * lmx));
    const float a = 1.0f / (float)frame;
    float ox, oy, oz;
    if(frame != 1u && fabsf(mom_x - lmx) > sigma * 3.0f) {
        ox = lx; oy = ly; oz = lz;
    } else {  // mix(last, cur, a) = last * (1 - a) + cur * a
        ox = lx * (1.0f - a) + cx * a;
        oy = ly * (1.0f - a) + cy * a;
        oz = lz * (1.0f - a) + cz * a;
    }
    hist_color[o] = ox; hist_color[o + 1] = oy; hist_color[o + 2] = oz;
    hist_moment[m] = lmx * (1.0f - a) + mom_x * a;
    hist_moment[m + 1] = lmy * (1.0f - a) + mom_y * a;
}

}  // namespace

void launch_post_merge(hipStream_t s, int npix, const float *di, const float *idi, const float *albedo, float *color) {
    hipLaunchKernelGGL(merge_kernel, dim3((npix + POST_BLOCK - 1) / POST_BLOCK), dim3(POST_BLOCK), 0, s, npix, di, idi, albedo, color);
}
void launch_post_tonemap(hipStream_t s, size_t n, const float *in, float *out) {
    hipLaunchKernelGGL(tonemap_kernel, dim3((unsigned)((n + POST_BLOCK - 1) / POST_BLOCK)), dim3(POST_BLOCK), 0, s, n, in, out);
}
void launch_post_static_blend(hipStream_t s, int npix, unsigned frame, const float *color, float *hist_color, float *hist_moment) {
    hipLaunchKernelGGL(static_blend_kernel, dim3((npix + POST_BLOCK - 1) / POST_BLOCK), dim3(POST_BLOCK), 0, s, npix, frame, color,
                       hist_color, hist_moment);
}

}  // namespace rta
