// The three users of the traversal core: stand-alone closest hit (pt_pass_intersect), the
// path tracer's extend stage and its shadow stage.
#include "kernels.h"

namespace rta {
namespace {

// ---- stand-alone closest hit -----------------------------------------------------------
struct RaySource {
    const float *rays;
    __device__ bool load(unsigned idx, float o[3], float d[3], bool &any_hit) const {
        const float2 *p = reinterpret_cast<const float2 *>(rays + (size_t)idx * 6);
        const float2 a = p[0], b = p[1], c = p[2];
        o[0] = a.x; o[1] = a.y; o[2] = b.x;
        d[0] = b.y; d[1] = c.x; d[2] = c.y;
        any_hit = false;
        return true;
    }
};
struct HitSink {
    pt_hit *hits;
    __device__ void store(unsigned idx, const LaneHit &h) const {
        pt_hit r;
        r.instance = h.inst;
        r.triangle = h.tri;
        r.t = h.t;
        r.u = h.u;
        r.v = h.v;
        hits[idx] = r;
    }
};

__global__ __launch_bounds__(TRACE_BLOCK) void intersect_kernel(TraceScene sc, const float *rays, unsigned n,
                                                                pt_hit *hits, unsigned *next_counter) {
    __shared__ uint2 s_stack[LDS_STACK * TRACE_BLOCK];
    RaySource src{rays};
    HitSink sink{hits};
    trace_persistent(sc, src, sink, next_counter, n, s_stack);
}

// ---- extend ------------------------------------------------------------------------------
struct ExtendSource {
    const unsigned *queue;
    const PathRay *rays;
    __device__ bool load(unsigned idx, float o[3], float d[3], bool &any_hit) const {
        const unsigned px = queue ? queue[idx] : idx;
        const float4 *p = reinterpret_cast<const float4 *>(rays + px);
        const float4 a = p[0], b = p[1];
        o[0] = a.x; o[1] = a.y; o[2] = a.z;
        d[0] = b.x; d[1] = b.y; d[2] = b.z;
        any_hit = false;
        return true;
    }
};
struct ExtendSink {
    const unsigned *queue;
    PathHit *hits;
    __device__ void store(unsigned idx, const LaneHit &h) const {
        const unsigned px = queue ? queue[idx] : idx;
        float4 *p = reinterpret_cast<float4 *>(hits + px);
        p[0] = make_float4(h.t, h.u, h.v, __int_as_float(h.tri));
        p[1] = make_float4(__int_as_float(h.inst), 0.0f, 0.0f, 0.0f);
    }
};

__global__ __launch_bounds__(TRACE_BLOCK) void extend_kernel(TraceScene sc, const unsigned *queue,
                                                             const unsigned *count, const PathRay *rays,
                                                             PathHit *hits, unsigned *next_counter) {
    __shared__ uint2 s_stack[LDS_STACK * TRACE_BLOCK];
    ExtendSource src{queue, rays};
    ExtendSink sink{queue, hits};
    trace_persistent(sc, src, sink, next_counter, *count, s_stack);
}

// ---- shadow ------------------------------------------------------------------------------
// Work item idx = q * n_shadow + k: light sample k of the path in slot queue[q].  The ray
// starts at that path's current ray origin (the shaded point).  Sample 0 (sky) only needs
// to know whether anything is hit; analytic lights compare the closest hit distance with
// the light distance, as shader/pathtracing.glsl:547 does.
struct ShadowSource {
    const unsigned *queue;
    const PathRay *rays;
    const ShadowEntry *shadow;
    int n_shadow, npix;
    __device__ bool load(unsigned idx, float o[3], float d[3], bool &any_hit) const {
        const unsigned q = idx / (unsigned)n_shadow, k = idx - q * (unsigned)n_shadow;
        const unsigned px = queue[q];
        const float4 *e = reinterpret_cast<const float4 *>(shadow + (size_t)k * npix + px);
        const float4 a = e[0], b = e[1];
        if(!(__float_as_uint(b.w) & 1u)) return false;
        const float4 ro = *reinterpret_cast<const float4 *>(rays + px);
        o[0] = ro.x; o[1] = ro.y; o[2] = ro.z;
        d[0] = a.x; d[1] = a.y; d[2] = a.z;
        any_hit = (k == 0);
        return true;
    }
};
struct ShadowSink {
    const unsigned *queue;
    ShadowEntry *shadow;
    int n_shadow, npix;
    __device__ void store(unsigned idx, const LaneHit &h) const {
        const unsigned q = idx / (unsigned)n_shadow, k = idx - q * (unsigned)n_shadow;
        const unsigned px = queue[q];
        ShadowEntry *e = shadow + (size_t)k * npix + px;
        const bool visible = h.tri < 0 || h.t * h.t > e->dis2;
        if(visible) e->flags |= 4u;
    }
};

__global__ __launch_bounds__(TRACE_BLOCK) void shadow_kernel(TraceScene sc, const unsigned *queue,
                                                             const unsigned *count, const PathRay *rays,
                                                             ShadowEntry *shadow, int n_shadow, int npix,
                                                             unsigned *next_counter) {
    __shared__ uint2 s_stack[LDS_STACK * TRACE_BLOCK];
    ShadowSource src{queue, rays, shadow, n_shadow, npix};
    ShadowSink sink{queue, shadow, n_shadow, npix};
    trace_persistent(sc, src, sink, next_counter, *count * (unsigned)n_shadow, s_stack);
}

}  // namespace

void launch_intersect(hipStream_t s, const TraceScene &sc, int blocks, const float *rays, unsigned n, pt_hit *hits,
                      unsigned *next_counter) {
    hipLaunchKernelGGL(intersect_kernel, dim3(blocks), dim3(TRACE_BLOCK), 0, s, sc, rays, n, hits, next_counter);
}

void launch_extend(hipStream_t s, const TraceScene &sc, int blocks, const unsigned *queue, const unsigned *count,
                   const PathRay *rays, PathHit *hits, unsigned *next_counter) {
    hipLaunchKernelGGL(extend_kernel, dim3(blocks), dim3(TRACE_BLOCK), 0, s, sc, queue, count, rays, hits,
                       next_counter);
}

void launch_shadow(hipStream_t s, const TraceScene &sc, int blocks, const unsigned *queue, const unsigned *count,
                   const PathRay *rays, ShadowEntry *shadow, int n_shadow, int npix, unsigned *next_counter) {
    hipLaunchKernelGGL(shadow_kernel, dim3(blocks), dim3(TRACE_BLOCK), 0, s, sc, queue, count, rays, shadow,
                       n_shadow, npix, next_counter);
}

}  // namespace rta
