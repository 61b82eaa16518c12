// Closest-hit / occlusion traversal of the two-level BVH, one ray per lane.
//
// Semantics are those of intersect_sceneBVH / intersect_meshBVH / intersect_aabb /
// intersect_triangle in the reference's shader/pathtracing.glsl:208-316, reproduced
// bit for bit in fp32:
//   - the visiting order is the reference's (right child first, no near/far sorting),
//   - a node is dropped when its box test fails or, at the moment it is taken off the
//     stack, its entry distance exceeds the closest hit so far (tnear > t_best),
//   - triangle leaves are tested without a box test of their own,
//   - a triangle is accepted when 0.001 <= t, 0 <= u, 0 <= v, u + v <= 1 and t < t_best.
// What differs is only how the work is laid out: both child boxes arrive with the parent
// (one 64-byte fetch per inner node, none per leaf), the box tests therefore run when the
// parent is visited and the entry distance travels with the deferred child on the stack;
// the right child is continued with directly instead of being pushed and popped.
//
// No fused multiply-add may be formed in this file (compile with -ffp-contract=off):
// the CPU checker evaluates the same expressions unfused.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "device_types.h"

namespace rta {

constexpr int TRACE_BLOCK = 256;
#ifndef RT_CORE_LDS_STACK
#define RT_CORE_LDS_STACK 12
#endif
constexpr int LDS_STACK = RT_CORE_LDS_STACK;  // deferred entries kept in LDS per lane (8 B each, 24 KiB per block)
constexpr float RT_INF = 1e18f;
constexpr float RT_RAY_MINDIS = 0.001f;

struct TraceScene {
    const char *mesh_buf;    // TNodes and TLeafs of all meshes (64-byte slots)
    const char *scene_buf;   // scene TNodes, then InstAccels (two slots each)
    SceneHeader hdr;
    uint2 *spill;            // [spill_depth][total threads of the launch]
    int spill_depth;
    unsigned *error_flag;    // set to 1 when a ray needed more stack than LDS + spill (host-visible word)
    unsigned *stats_dbg;     // RT_TRACE_STATS builds: counters from byte 16 on
};

struct LaneHit {
    float t, u, v;
    int tri, inst;
};

// intersect_aabb, shader/pathtracing.glsl:208-221.  Division by the direction is IEEE
// division (zero components give infinities / NaN exactly as they would there); min and
// max drop a NaN operand.
__device__ __forceinline__ bool slab_test(float lox, float loy, float loz, float hix, float hiy, float hiz, float ox,
                                          float oy, float oz, float dx, float dy, float dz, float &tnear) {
    float t1x = (lox - ox) / dx, t1y = (loy - oy) / dy, t1z = (loz - oz) / dz;
    float t2x = (hix - ox) / dx, t2y = (hiy - oy) / dy, t2z = (hiz - oz) / dz;
    float tmi = fmaxf(0.0f, fminf(t1x, t2x));
    tmi = fmaxf(tmi, fminf(t1y, t2y));
    tmi = fmaxf(tmi, fminf(t1z, t2z));
    float tmx = fminf(RT_INF, fmaxf(t1x, t2x));
    tmx = fminf(tmx, fmaxf(t1y, t2y));
    tmx = fminf(tmx, fmaxf(t1z, t2z));
    tnear = tmi;
    return tmx >= tmi && tmx >= RT_RAY_MINDIS;
}


// Ray indices are handed out per WAVE in batches: one returning atomic on the shared
// counter per batch instead of one per lane-refill.  A single counter word sustains only
// ~88 returning atomics per microsecond chip-wide; drawing single indices made that the
// bottleneck of every trace launch (16 Mi rays / ~24 lanes per refill = 0.7 M atomics = 8 ms).
// MIN_BATCH: the smallest batch.  64 fills a wave at once; the second-pass launches, which carry
// a few dozen long rays, use 4 so that those spread over many waves (a wave steps through all
// kinds of node its lanes hold, one after the other: the fewer rays share a wave, the sooner
// each is done).
template <unsigned MIN_BATCH>
struct WaveRayFeedT {
    unsigned base = 0, left = 0;  // wave-uniform: next index of the current batch, indices left in it
    bool dry = false;             // wave-uniform: the counter has passed n_rays
    unsigned batch;
    __device__ explicit WaveRayFeedT(unsigned n_rays) {
        // enough batches for every wave of the launch to get several; at least one ray per lane
        const unsigned waves = gridDim.x * (blockDim.x / 64u);
        unsigned b = n_rays / (waves * 4u + 1u);
        b = b < MIN_BATCH ? MIN_BATCH : (b > 1024u ? 1024u : b);
        batch = b >= 64u ? (b & ~63u) : b;
    }
    // Every lane of the wave must call this together.  Returns true and sets idx for lanes
    // that asked for a ray and got one; lanes that asked but got none ask again next time.
    __device__ bool take(bool want, unsigned *next_ray, unsigned n_rays, unsigned &idx) {
        const unsigned long long m = __ballot(want);
        if(m == 0ull) return false;
        if(left == 0u && !dry) {
            unsigned b = 0;
            if((threadIdx.x & 63u) == 0u) b = atomicAdd(next_ray, batch);
            b = __builtin_amdgcn_readfirstlane(b);
            base = b;
            left = b < n_rays ? (n_rays - b < batch ? n_rays - b : batch) : 0u;
            dry = left == 0u;
        }
        const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        const bool got = want && rank < left;
        idx = base + rank;
        const unsigned asked = (unsigned)__popcll(m);
        const unsigned taken = asked < left ? asked : left;
        base += taken;
        left -= taken;
        return got;
    }
};

using WaveRayFeed = WaveRayFeedT<64u>;
using SparseRayFeed = WaveRayFeedT<4u>;

// The number of busy lanes below which a wave refills: the kernel's own, unless the source names one.
template <class S, class = void>
struct has_refill_below : std::false_type {};
template <class S>
struct has_refill_below<S, std::void_t<decltype(S::REFILL_BELOW)>> : std::true_type {};
template <class S>
constexpr int refill_below(int kernel_default) {
    if constexpr(has_refill_below<S>::value) return S::REFILL_BELOW;
    else return kernel_default;
}

// The same interface over a set of rays that belongs to the calling wave alone (indices
// 0 .. n_rays-1, no counter in memory): for kernels in which a wave traces the rays of its own
// paths between two shading steps.
struct LocalRayFeed {
    unsigned base = 0, left;
    bool dry;
    __device__ explicit LocalRayFeed(unsigned n_rays) : left(n_rays), dry(n_rays == 0u) {}
    __device__ bool take(bool want, unsigned *, unsigned, unsigned &idx) {
        const unsigned long long m = __ballot(want);
        if(m == 0ull) return false;
        const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        const bool got = want && rank < left;
        idx = base + rank;
        const unsigned asked = (unsigned)__popcll(m);
        const unsigned taken = asked < left ? asked : left;
        base += taken;
        left -= taken;
        dry = left == 0u;
        return got;
    }
};

// A ray with a NaN in its origin or direction cannot hit anything (every triangle test
// compares NaN and fails), but NaN-dropping min/max let it pass every box: it would visit the
// whole scene to report a miss.  Such rays (the reference's refract() makes them) are
// finished as misses without a walk.
__device__ __forceinline__ bool ray_has_nan(const float o[3], const float d[3]) {
    return o[0] != o[0] || o[1] != o[1] || o[2] != o[2] || d[0] != d[0] || d[1] != d[1] || d[2] != d[2];
}

// Persistent-wave traversal.  Every lane owns one ray at a time; a lane whose ray is
// finished takes the next ray index (WaveRayFeed) as soon as fewer than REFILL_BELOW lanes
// of its wave are still busy,
// so long rays do not hold 63 idle lanes hostage.  Every wave leaves the loop once the
// counter has passed n_rays and its own lanes are done.
//
// Source: bool load(unsigned idx, float o[3], float d[3], bool &any_hit)   false = no ray for idx
// Sink:   void store(unsigned idx, const LaneHit &h)                        h.tri < 0 = miss
template <class Feed = WaveRayFeed, class Source, class Sink>
__device__ __forceinline__ void trace_persistent(const TraceScene &sc, Source &src, Sink &sink, unsigned *next_ray,
                                                 unsigned n_rays, uint2 *s_stack) {
    constexpr int REFILL_BELOW = 44;
    const unsigned tid = threadIdx.x;
    const unsigned gtid = blockIdx.x * blockDim.x + tid;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;

    bool active = false, have = false, exhausted = false, any_hit = false;
    bool pending = false;  // the lane's ray has ended; its result is handed over when the wave refills (all such lanes at once)
    Feed feed(n_rays);
    unsigned ray_idx = 0;
    int sp = 0;
    float wox = 0, woy = 0, woz = 0, wdx = 0, wdy = 0, wdz = 0;  // world-space ray
    float lox = 0, loy = 0, loz = 0, ldx = 0, ldy = 0, ldz = 0;  // ray in the current instance's space
    LaneHit best;
    best.t = RT_INF; best.u = 0; best.v = 0; best.tri = -1; best.inst = -1;
    int cur_inst = -1;
    unsigned ref = 0;
    float tn = 0;

    for(;;) {
        {
            if(pending) {
                sink.store(ray_idx, best);
                pending = false;
            }
            unsigned idx = 0;
            if(feed.take(!active, next_ray, n_rays, idx)) {
                float o[3], d[3];
                bool ah = false;
                if(src.load(idx, o, d, ah)) {
                    ray_idx = idx;
                    any_hit = ah;
                    wox = o[0]; woy = o[1]; woz = o[2];
                    wdx = d[0]; wdy = d[1]; wdz = d[2];
                    best.t = RT_INF; best.u = 0; best.v = 0; best.tri = -1; best.inst = -1;
                    // A ray the ordered traversal could not decide comes with that pass's closest
                    // distance.  What the reference returns lies within 5e-4 of it, and a subtree
                    // whose box is entered beyond 1.001 times that distance can neither hold the
                    // answer nor prune a subtree that does (its hits are farther than every such
                    // subtree's entry), so it is skipped: start from that bound instead of infinity.
                    const float provisional = src.bound(idx);
                    if(provisional < RT_INF)
                        best.t = provisional * 1.001f + 8e-6f * (fabsf(wox) + fabsf(woy) + fabsf(woz) + sc.hdr.extent);
                    active = true;
                    sp = 0;
                    float t0;
                    have = slab_test(sc.hdr.root_lo[0], sc.hdr.root_lo[1], sc.hdr.root_lo[2], sc.hdr.root_hi[0],
                                     sc.hdr.root_hi[1], sc.hdr.root_hi[2], wox, woy, woz, wdx, wdy, wdz, t0);
                    if(ray_has_nan(o, d)) have = false;
                    ref = sc.hdr.root_ref;
                    tn = t0;
                }
            }
            exhausted = feed.dry;
        }
        if(!__any(active)) {
            if(__all(exhausted)) break;
            continue;  // only rays that needed no traversal were drawn: draw again
        }
        const bool drained = __any(exhausted);

        for(;;) {
            if(active && !have) {
                if(sp > 0) {
                    --sp;
                    uint2 e = sp < LDS_STACK ? s_stack[sp * TRACE_BLOCK + tid]
                                             : sc.spill[(size_t)(sp - LDS_STACK) * nthreads + gtid];
                    ref = e.x;
                    tn = __uint_as_float(e.y);
                    have = true;
                } else {
                    if(best.tri < 0) best.t = RT_INF;  // (a bound is not a hit)
                    pending = true;
                    active = false;
                }
            }
            const int nact = __popcll(__ballot(active));
            if(nact == 0) break;
            if(!drained && nact < REFILL_BELOW) break;
            if(!have) continue;

            const unsigned kind = ref >> 30, idx = ref & REF_INDEX_MASK;
            if(kind == REF_TRIANGLE) {
                // intersect_triangle, shader/pathtracing.glsl:224-249 (Moller-Trumbore)
                // TLeaf: the parent box (not needed in this order), then v0, e1, e2 from float 6 on
                const float4 *tp = reinterpret_cast<const float4 *>(sc.mesh_buf + ((size_t)idx << 6));
                const float4 a = tp[1], b = tp[2], c = tp[3];
                const float v0x = a.z, v0y = a.w, v0z = b.x;
                const float e1x = b.y, e1y = b.z, e1z = b.w;
                const float e2x = c.x, e2y = c.y, e2z = c.z;
                const float sx = lox - v0x, sy = loy - v0y, sz = loz - v0z;
                const float s1x = ldy * e2z - e2y * ldz, s1y = ldz * e2x - e2z * ldx, s1z = ldx * e2y - e2x * ldy;
                const float s2x = sy * e1z - e1y * sz, s2y = sz * e1x - e1z * sx, s2z = sx * e1y - e1x * sy;
                const float k = 1.0f / (s1x * e1x + s1y * e1y + s1z * e1z);
                const float t = (s2x * e2x + s2y * e2y + s2z * e2z) * k;
                const float u = (s1x * sx + s1y * sy + s1z * sz) * k;
                const float v = (s2x * ldx + s2y * ldy + s2z * ldz) * k;
                have = false;
                if(RT_RAY_MINDIS <= t && 0.0f <= u && 0.0f <= v && u + v <= 1.0f && t < best.t) {
                    best.t = t; best.u = u; best.v = v;
                    best.tri = __float_as_int(c.w);  // TLeaf::triangle
                    best.inst = cur_inst;
                    if(any_hit) sp = 0;  // occlusion query: the first accepted hit decides
                }
            } else if(tn > best.t) {
                have = false;  // pruned when taken off the stack
            } else if(kind == REF_INSTANCE) {
                // entry into intersect_meshBVH, shader/pathtracing.glsl:254-264
                const float4 *ip = reinterpret_cast<const float4 *>(sc.scene_buf + ((size_t)idx << 6));
                // rows, (root_ref, material), root box: lo.xyz hi.x | hi.yz ..
                const float4 r0 = ip[0], r1 = ip[1], r2 = ip[2], rr = ip[3], bx0 = ip[4], bx1 = ip[5];
                lox = (r0.x * wox + r0.y * woy) + (r0.z * woz + r0.w);
                loy = (r1.x * wox + r1.y * woy) + (r1.z * woz + r1.w);
                loz = (r2.x * wox + r2.y * woy) + (r2.z * woz + r2.w);
                ldx = (r0.x * wdx + r0.y * wdy) + r0.z * wdz;
                ldy = (r1.x * wdx + r1.y * wdy) + r1.z * wdz;
                ldz = (r2.x * wdx + r2.y * wdy) + r2.z * wdz;
                cur_inst = (int)((idx - sc.hdr.inst_base) >> 1);
                ref = __float_as_uint(rr.x);
                if((ref >> 30) != REF_TRIANGLE) {
                    float t0;
                    have = slab_test(bx0.x, bx0.y, bx0.z, bx0.w, bx1.x, bx1.y, lox, loy, loz, ldx, ldy, ldz, t0);
                    tn = t0;
                }
            } else {
                const bool mesh = kind == REF_MESH_INNER;
                const float4 *np = reinterpret_cast<const float4 *>((mesh ? sc.mesh_buf : sc.scene_buf) + ((size_t)idx << 6));
                const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
                const float rox = mesh ? lox : wox, roy = mesh ? loy : woy, roz = mesh ? loz : woz;
                const float rdx = mesh ? ldx : wdx, rdy = mesh ? ldy : wdy, rdz = mesh ? ldz : wdz;
                float t0, t1;
                bool p0 = slab_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, rox, roy, roz, rdx, rdy, rdz, t0);
                bool p1 = slab_test(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, rox, roy, roz, rdx, rdy, rdz, t1);
                const unsigned c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
                // triangle leaves carry no box test of their own in the reference
                if((c0 >> 30) == REF_TRIANGLE) { p0 = true; t0 = 0.0f; }
                if((c1 >> 30) == REF_TRIANGLE) { p1 = true; t1 = 0.0f; }
                if(p1) {
                    ref = c1;
                    tn = t1;
                    if(p0) {
                        const uint2 e = make_uint2(c0, __float_as_uint(t0));
                        if(sp < LDS_STACK) {
                            s_stack[sp * TRACE_BLOCK + tid] = e;
                            ++sp;
                        } else if(sp - LDS_STACK < sc.spill_depth) {
                            sc.spill[(size_t)(sp - LDS_STACK) * nthreads + gtid] = e;
                            ++sp;
                        } else {
                            *sc.error_flag = 1u;  // host sizes the spill from the tree height; never expected
                        }
                    }
                } else if(p0) {
                    ref = c0;
                    tn = t0;
                } else {
                    have = false;
                }
            }
        }
    }
}

}  // namespace rta
