// Reads and writes of the per-slot path state that several kernels share (the extend / shadow
// launches of trace_kernels.hip and the per-wave loop of tail_kernels.hip), so that they cannot
// drift apart: the arithmetic here decides output bits.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.h"
#include "trace_core.h"

namespace rta {

__device__ __forceinline__ void load_path_ray(const PathRay *rays, unsigned px, float o[3], float d[3]) {
    const float4 *p = reinterpret_cast<const float4 *>(rays + px);
    const float4 a = p[0], b = p[1];
    o[0] = a.x; o[1] = a.y; o[2] = a.z;
    d[0] = b.x; d[1] = b.y; d[2] = b.z;
}

__device__ __forceinline__ void write_path_hit(PathHit *hits, unsigned px, const LaneHit &h) {
    float4 *p = reinterpret_cast<float4 *>(hits + px);
    p[0] = make_float4(h.t, h.u, h.v, __int_as_float(h.tri));
    p[1] = make_float4(__int_as_float(h.inst), 0.0f, 0.0f, 0.0f);
}

// The sky light sample of a path vertex once its occlusion query has ended with `h`
// (shader/pathtracing.glsl:519-525): with the sky as the only light there is one pending sample
// per path, so whoever ends the query adds it to the path's accumulator; the order of the float
// additions per path stays the reference's (this sample, then whatever later vertices add).
__device__ __forceinline__ void apply_sky_sample(const ShadowEntry *shadow, const PathThroughput *thr, PathAccum *acc, unsigned px,
                                                 const LaneHit &h) {
    const float4 *e = reinterpret_cast<const float4 *>(shadow + px);
    const float4 e0 = e[0], e1 = e[1];
    if(!(h.tri < 0 || h.t * h.t > e0.w)) return;  // occluded
    const float4 t0 = *reinterpret_cast<const float4 *>(thr + px);
    float4 *a = reinterpret_cast<float4 *>(acc + px);
    float4 g = a[1];  // gi and its padding word: one load, one store
    g.x = g.x + t0.x * e1.x;
    g.y = g.y + t0.y * e1.y;
    g.z = g.z + t0.z * e1.z;
    a[1] = g;
    if(__float_as_uint(e1.w) & 2u) {  // di (first_exist behind it stays)
        struct __attribute__((packed, aligned(4))) F3 { float x, y, z; } d = {e1.x, e1.y, e1.z};
        *reinterpret_cast<F3 *>(acc + px) = d;
    }
}

}  // namespace rta
