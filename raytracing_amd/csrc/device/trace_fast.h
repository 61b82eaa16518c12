// Ordered closest-hit / occlusion traversal: the fast path.
//
// What the reference's traversal (shader/pathtracing.glsl:254-316) returns can be stated
// without its visiting order.  Call a triangle a CANDIDATE of a ray when the reference's
// box tests pass on every node above it; because a child's box lies inside its parent's
// and fp32 subtraction and division are monotone, that is the case exactly when the box of
// the triangle's parent node passes (in the instance's space) and the box of the
// instance's scene leaf passes (in world space).  The reference returns the candidate hit
// with the smallest t -- unless its order-dependent prune `tnear > t_best` discarded that
// winner's subtree, which needs another candidate hit whose t lies between the winner's t
// and the entry distance of the winner's own parent box: a near-tie in t (shared edges,
// coplanar duplicates), since the two differ only by rounding.
//
// So this kernel is free to visit nodes in any order and to cull with cheaper arithmetic,
// as long as
//   (1) culling is conservative: boxes are tested with a per-ray reciprocal and the
//       comparisons are relaxed by 1e-6, three times the arithmetic's error; boxes may be
//       larger than the reference's (quantised nodes round outwards) but never smaller;
//   (2) a hit is accepted only for a candidate: the parent box (and, where a scene node
//       hands over to an instance, the scene leaf box) is re-tested with the reference's
//       exact IEEE-division slab test whenever the relaxed test did not pass with the same
//       margin to spare;
//   (3) subtrees are pruned only beyond t_best * (1 + 1e-3) (+ an absolute term scaled by
//       the scene size), so every hit inside the near-tie band is seen;
//   (4) the smallest t among all OTHER hits is tracked, and a ray whose runner-up ties with
//       the winner or lies below the entry distance of the winner's parent box or scene leaf
//       box (the only results that could have pruned the winner away) is reported as
//       ambiguous: the caller re-traces it with the reference-order kernel (trace_core.h),
//       which is exact by construction.
//   (5) every direction component the ray is traced with -- in world space and in the space of
//       each instance it enters -- is a NORMAL number.  With a zero component the reference's
//       slab test divides 0 by 0 where the origin lies on a box plane, and its NaN-dropping
//       min / max then pass a box of no thickness whose ancestors FAIL (a child passing no longer
//       implies its parents do); with a denormal component the hardware reciprocal is infinite
//       where the quotient is not.  Such a ray is reported as ambiguous at once -- closest-hit
//       and occlusion queries alike -- and left to the reference-order kernel.
// The triangle test itself is the reference's Moller-Trumbore, bit for bit.
// Occlusion queries ("is there any candidate hit") have no order dependence at all.
// (The host refuses this kernel for buffers whose child boxes are not inside their parents'.)
//
// Because only the parent box of a triangle and the scene leaf box of an instance matter
// exactly, the boxes in between need not be visited one by one: the kernel walks the
// reference's binary trees collapsed to FOUR children per node (device_types.h: QNode, 64
// bytes with boxes on a 256-step grid, or WNode, 128 bytes with the fp32 boxes), which halves
// the dependent fetches per ray.  A triangle child carries its reference parent's box.
//
// Loop shape: every lane issues the loads of ONE record per iteration, whatever it is about
// to visit (wide node, triangle leaf, instance record), the wave waits once, and only then the
// three kinds are told apart.  Children that pass are sorted by entry distance; the nearest
// is visited next, the others are deferred so that they come back nearest first.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.h"
#include "trace_core.h"

namespace rta {

constexpr int FAST_BLOCK = 256;
#ifndef RT_FAST_LDS_STACK
#define RT_FAST_LDS_STACK 16
#endif
constexpr int FAST_LDS_STACK = RT_FAST_LDS_STACK;  // 8-byte entries per lane in LDS: 2 KiB per entry and block

struct FastScene {
    const char *mesh_buf;    // TNodes and TLeafs of all meshes (64-byte slots)
    const char *scene_buf;   // scene TNodes, then InstAccels (two slots each)
    SceneHeader hdr;
    // the flattened tree (trace_flat.h), or null: FlatLeafs (two slots each), then its QNodes
    const char *flat_buf;
    uint32_t flat_root_ref;
    float flat_pad;    // box padding per unit of |origin| + flat_reach, see trace_flat.h
    float flat_reach;  // sum over the axes of the largest |coordinate| of the scene box
    uint2 *spill;
    int spill_depth;
    unsigned *error_flag;
    unsigned *stats_dbg;
};

// 1/x to one ulp (v_rcp_f32), for the relaxed tests only.  With the subtraction and the
// product that is at most 2.5 ulp = 3e-7 away from the reference's IEEE quotient, a third of
// the 1e-6 the comparisons are relaxed by.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// some component is zero or denormal (rule (5) above; NaN components are the caller's business)
__device__ __forceinline__ bool has_subnormal_component(float x, float y, float z) {
    return fminf(fabsf(x), fminf(fabsf(y), fabsf(z))) < 1.17549435e-38f;
}

// Slab distances with a reciprocal direction: entry (>= 0) and exit.
__device__ __forceinline__ void slab_range(float lox, float loy, float loz, float hix, float hiy, float hiz, float ox,
                                           float oy, float oz, float ix, float iy, float iz, float &tmi, float &tmx) {
    const float t1x = (lox - ox) * ix, t2x = (hix - ox) * ix;
    const float t1y = (loy - oy) * iy, t2y = (hiy - oy) * iy;
    const float t1z = (loz - oz) * iz, t2z = (hiz - oz) * iz;
    tmi = fmaxf(fmaxf(0.0f, fminf(t1x, t2x)), fmaxf(fminf(t1y, t2y), fminf(t1z, t2z)));
    tmx = fminf(fminf(RT_INF, fmaxf(t1x, t2x)), fminf(fmaxf(t1y, t2y), fmaxf(t1z, t2z)));
}
// The reference passes a box when tmx >= tmi && tmx >= RAY_MINDIS (pathtracing.glsl:208-221).
// relaxed_pass never fails a box the reference passes; clear_pass never passes one it fails.
__device__ __forceinline__ bool relaxed_pass(float tmi, float tmx) {
    const float up = tmx * 1.000002f;
    return up >= tmi && up >= RT_RAY_MINDIS;
}
__device__ __forceinline__ bool clear_pass(float tmi, float tmx) {
    const float dn = tmx * 0.999998f;
    return dn >= tmi && dn >= 0.001001f;
}
// relaxed slab test; tn = entry distance (>= 0)
__device__ __forceinline__ bool slab_fast(float lox, float loy, float loz, float hix, float hiy, float hiz, float ox,
                                          float oy, float oz, float ix, float iy, float iz, float &tn, bool &clear) {
    float tmx;
    slab_range(lox, loy, loz, hix, hiy, hiz, ox, oy, oz, ix, iy, iz, tn, tmx);
    clear = clear_pass(tn, tmx);
    return relaxed_pass(tn, tmx);
}

// Source: bool load(unsigned idx, float o[3], float d[3], bool &any_hit)
// Sink:   void store(unsigned idx, const LaneHit &h, bool ambiguous)
// QUANT: the wide nodes are QNodes (64 bytes, decoded on the fly) instead of WNodes.
template <bool QUANT, class Feed = WaveRayFeed, class Source, class Sink>
__device__ __forceinline__ void trace_ordered(const FastScene &sc, Source &src, Sink &sink, unsigned *next_ray,
                                              unsigned n_rays, uint2 *s_stack) {
#ifndef RT_FAST_REFILL
#define RT_FAST_REFILL 40
#endif
    constexpr int REFILL_BELOW = RT_FAST_REFILL;  // refill when fewer lanes than this still own a ray
#ifndef RT_LEAF_BATCH
#define RT_LEAF_BATCH 12
#endif
#ifndef RT_INST_BATCH
#define RT_INST_BATCH 6
#endif
#ifndef RT_SERVE_ALL_BELOW
#define RT_SERVE_ALL_BELOW 16
#endif
    constexpr int LEAF_BATCH = RT_LEAF_BATCH, INST_BATCH = RT_INST_BATCH, SERVE_ALL_BELOW = RT_SERVE_ALL_BELOW;
    const unsigned tid = threadIdx.x;
    const unsigned gtid = blockIdx.x * blockDim.x + tid;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;

    bool active = false, have = false, exhausted = false, any_hit = false;
    bool uncertain = false;  // rule (5): this ray is the reference-order kernel's
    // A ray that has ended keeps its result in the lane until the wave refills, where the results of all the lanes
    // that have ended since are handed over at once (instead of for one to three lanes in almost every step).
    bool pending = false, pending_ambiguous = false;
    Feed feed(n_rays);
    unsigned ray_idx = 0;
    int sp = 0;
    float wox = 0, woy = 0, woz = 0, wdx = 0, wdy = 0, wdz = 0;  // world ray
    float cox = 0, coy = 0, coz = 0, cdx = 0, cdy = 0, cdz = 0;  // ray in the current space
    float cix = 0, ciy = 0, ciz = 0;                             // 1 / direction in the current space
    bool in_mesh = false;
    LaneHit best;
    best.t = RT_INF; best.u = 0; best.v = 0; best.tri = -1; best.inst = -1;
    float second_t = RT_INF, abs_slack = 0.0f;
    float inst_tw = 0.0f;   // entry distance of the current instance's scene leaf box
    float best_gate = 0.0f; // for the closest hit: the larger entry distance of its parent box and its scene leaf box
    int cur_inst = -1;
    unsigned ref = 0;
    float tn = 0;

#ifdef RT_TRACE_STATS
    unsigned long long st_slots = 0, st_inner = 0, st_leaf = 0, st_inst = 0, st_iter = 0, st_leaf_runs = 0, st_inst_runs = 0, st_scene = 0;
#endif
    for(;;) {
        {
            if(pending) {
                sink.store(ray_idx, best, pending_ambiguous);
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
                    cox = wox; coy = woy; coz = woz;
                    cdx = wdx; cdy = wdy; cdz = wdz;
                    cix = fast_rcp(wdx); ciy = fast_rcp(wdy); ciz = fast_rcp(wdz);
                    in_mesh = false;
                    best.t = RT_INF; best.u = 0; best.v = 0; best.tri = -1; best.inst = -1;
                    second_t = RT_INF;
                    abs_slack = 1e-6f * (fabsf(wox) + fabsf(woy) + fabsf(woz) + sc.hdr.extent);
                    active = true;
                    sp = 0;
                    // a single-instance scene: the root's box is the scene leaf box, whose exact
                    // test decides candidates; otherwise the boxes below the root imply it
                    have = true;
                    tn = 0.0f;
                    if((sc.hdr.root_ref >> 30) == REF_INSTANCE) {
                        float t0;
                        have = slab_test(sc.hdr.root_lo[0], sc.hdr.root_lo[1], sc.hdr.root_lo[2], sc.hdr.root_hi[0],
                                         sc.hdr.root_hi[1], sc.hdr.root_hi[2], wox, woy, woz, wdx, wdy, wdz, t0);
                        tn = t0;
                    }
                    uncertain = false;
                    if(ray_has_nan(o, d)) {
                        have = false;  // a certain miss, see trace_core.h
                    } else if(has_subnormal_component(wdx, wdy, wdz)) {
                        uncertain = true;
                        have = false;
                    }
                    ref = sc.hdr.wide_root_ref;
                }
            }
            exhausted = feed.dry;
        }
        if(!__any(active)) {
            if(__all(exhausted)) break;
            continue;
        }
        const bool drained = __any(exhausted);

        for(;;) {
            // ---- take the next deferred entry that is still worth visiting, or finish the ray
            const float prune_t = best.t * 1.001f + 8.0f * abs_slack;
            // (one entry per step: a lane whose entry has fallen behind its closest hit
            // tries again in the next step instead of making the wave loop here)
            if(active && !have) {
                if(sp > 0) {
                    --sp;
                    const uint2 e = sp < FAST_LDS_STACK ? s_stack[sp * FAST_BLOCK + tid]
                                                        : sc.spill[(size_t)(sp - FAST_LDS_STACK) * nthreads + gtid];
                    ref = e.x;
                    tn = __uint_as_float(e.y);
                    have = !(tn > prune_t);
                } else {
                    // The reference loses the closest candidate only by pruning a box above it,
                    // which needs a result closer than that box's entry distance: the runner-up
                    // below best_gate (relaxed by the error of the reciprocal arithmetic), or an
                    // exact tie (then the visiting order decides).  A gate well beyond the hit
                    // itself means box and triangle arithmetic disagree: nothing is assumed then,
                    // and the second pass starts without a bound.
                    const bool odd = best_gate > best.t * 1.0005f + 4.0f * abs_slack;
                    const bool ambiguous = uncertain || (!any_hit && best.tri >= 0 && (odd || second_t <= best.t || second_t < best_gate * 1.000004f));
                    if(ambiguous && odd) best.t = RT_INF;
                    if(uncertain) {  // nothing found so far counts: the second pass starts without a bound
                        best.t = RT_INF; best.tri = -1; best.inst = -1;
                    }
                    pending = true;
                    pending_ambiguous = ambiguous;
                    active = false;
                }
            }
            const int nact = __popcll(__ballot(active));
            if(nact == 0) break;
            if(!drained && nact < REFILL_BELOW) break;

            // ---- which kinds of entry this iteration serves.  Inner nodes always; triangle
            // leaves and instance entries only once enough lanes hold one (they wait, holding
            // their entry), because each kind's code runs for the whole wave however few lanes
            // need it: a ray meets six inner nodes per leaf and ten per instance entry.
            const unsigned kind = ref >> 30;
            const bool pair = kind == REF_TRIANGLE && (ref & REF_PAIR) != 0u;  // two sibling leaves in adjacent slots
            const unsigned idx = kind == REF_TRIANGLE ? (ref & REF_SLOT_MASK) : (ref & REF_INDEX_MASK);
            const bool is_leaf = have && kind == REF_TRIANGLE, is_inst = have && kind == REF_INSTANCE;
            const bool is_inner = have && !is_leaf && !is_inst;
            const int n_inner = __popcll(__ballot(is_inner));
#ifdef RT_PICK_ONE
            // one kind of entry per step: the one that most lanes hold (weighted), so that a step runs one
            // body at the best lane count on offer instead of up to three at whatever each finds
            const int n_leaf = __popcll(__ballot(is_leaf)), n_inst = __popcll(__ballot(is_inst));
            const int s_in = n_inner * RT_W_INNER, s_lf = n_leaf * RT_W_LEAF, s_is = n_inst * RT_W_INST;
            const bool serve_inner = s_in >= s_lf && s_in >= s_is;
            const bool serve_leaf = !serve_inner && s_lf >= s_is;
            const bool serve_inst = !serve_inner && !serve_leaf;
            const bool go = (is_inner && serve_inner) || (is_leaf && serve_leaf) || (is_inst && serve_inst);
#else
            const bool serve_inner = true;
            const bool starving = n_inner < SERVE_ALL_BELOW;
            const bool serve_leaf = starving || __popcll(__ballot(is_leaf)) >= LEAF_BATCH;
            const bool serve_inst = starving || __popcll(__ballot(is_inst)) >= INST_BATCH;
            const bool go = is_inner || (is_leaf && serve_leaf) || (is_inst && serve_inst);
#endif
#ifdef RT_TRACE_STATS
            st_slots += 1;
            {
                const bool leaf_runs = serve_leaf && __ballot(is_leaf) != 0ull, inst_runs = serve_inst && __ballot(is_inst) != 0ull;
                if((threadIdx.x & 63u) == 0u) {
                    st_iter++;
                    if(leaf_runs) st_leaf_runs++;
                    if(inst_runs) st_inst_runs++;
                }
            }
            if(go) {
                if(kind == REF_TRIANGLE) st_leaf++;
                else if(kind == REF_INSTANCE) st_inst++;
                else if(RT_TRACE_STATS == 3 && kind == REF_SCENE_INNER) st_scene++;
                else st_inner++;
            }
#endif
            // ---- one fetch per served lane: 64 bytes of a leaf, 128 of a wide node or an instance
            float4 q0, q1, q2, q3, q4, q5, q6;
            if(go) {
                const char *base = ((kind & 2u) ? sc.scene_buf : sc.mesh_buf) + ((size_t)idx << 6);
                const float4 *rp = reinterpret_cast<const float4 *>(base);
                q0 = rp[0]; q1 = rp[1]; q2 = rp[2]; q3 = rp[3];
                if(QUANT ? is_inst : !is_leaf) { q4 = rp[4]; q5 = rp[5]; q6 = rp[6]; }
            }  // (lanes that are not served leave them unset and never read them)
            // scene-level entries are met with the world ray
            if(go && kind >= REF_SCENE_INNER && in_mesh) {
                cox = wox; coy = woy; coz = woz;
                cdx = wdx; cdy = wdy; cdz = wdz;
                cix = fast_rcp(wdx); ciy = fast_rcp(wdy); ciz = fast_rcp(wdz);
                in_mesh = false;
            }

            auto push = [&](unsigned r, unsigned key) {
                const uint2 e = make_uint2(r, key);
                if(sp < FAST_LDS_STACK) {
                    s_stack[sp * FAST_BLOCK + tid] = e;
                    ++sp;
                } else if(sp - FAST_LDS_STACK < sc.spill_depth) {
                    sc.spill[(size_t)(sp - FAST_LDS_STACK) * nthreads + gtid] = e;
                    ++sp;
                } else {
                    *sc.error_flag = 1u;
                }
            };
            if(serve_inner && is_inner) {
                // one wide node: up to four child boxes; the nearest child next, the others
                // deferred so that they come back nearest first.  Entry distances are >= 0, so
                // their bit patterns order like the numbers and 0xFFFFFFFF marks "not taken".
                float t0, t1, t2, t3, x0, x1, x2, x3;
                unsigned r0, r1, r2, r3;
                if(QUANT) {
                    // q0 = origin.xyz, scale.x; q1 = scale.yz, references 0 1; q2 = references 2 3, then
                    // lo.x, lo.y of the four children (one byte each); q3 = lo.z, hi.x, hi.y, hi.z
                    const float sx = q0.w, sy = q1.x, sz = q1.y;
                    const unsigned lx = __float_as_uint(q2.z), ly = __float_as_uint(q2.w), lz = __float_as_uint(q3.x);
                    const unsigned hx = __float_as_uint(q3.y), hy = __float_as_uint(q3.z), hz = __float_as_uint(q3.w);
                    // The host stores lo <= hi on every axis, so the sign of the direction tells
                    // which plane is met first: one select per axis for all four children instead
                    // of a min and a max per child and axis.  A NaN plane distance (0 * inf)
                    // drops out of max3 / min3 and only lets more rays through.
                    const bool ngx = (int)__float_as_uint(cix) < 0, ngy = (int)__float_as_uint(ciy) < 0, ngz = (int)__float_as_uint(ciz) < 0;
                    const unsigned nx = ngx ? hx : lx, fx = ngx ? lx : hx;
                    const unsigned ny = ngy ? hy : ly, fy = ngy ? ly : hy;
                    const unsigned nz = ngz ? hz : lz, fz = ngz ? lz : hz;
#define RT_Q(packed, i, scale, org, o, inv) ((__builtin_fmaf((float)(((packed) >> (8 * (i))) & 0xFFu), scale, org) - (o)) * (inv))
#define RT_QBOX(i, T_IN, T_OUT)                                                                                              \
    T_IN = fmaxf(0.0f, fmaxf(fmaxf(RT_Q(nx, i, sx, q0.x, cox, cix), RT_Q(ny, i, sy, q0.y, coy, ciy)), RT_Q(nz, i, sz, q0.z, coz, ciz))); \
    T_OUT = fminf(RT_INF, fminf(fminf(RT_Q(fx, i, sx, q0.x, cox, cix), RT_Q(fy, i, sy, q0.y, coy, ciy)), RT_Q(fz, i, sz, q0.z, coz, ciz)))
                    RT_QBOX(0, t0, x0);
                    RT_QBOX(1, t1, x1);
                    RT_QBOX(2, t2, x2);
                    RT_QBOX(3, t3, x3);
#undef RT_QBOX
#undef RT_Q
                    r0 = __float_as_uint(q1.z); r1 = __float_as_uint(q1.w); r2 = __float_as_uint(q2.x); r3 = __float_as_uint(q2.y);
                } else {
                    slab_range(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, cox, coy, coz, cix, ciy, ciz, t0, x0);
                    slab_range(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, cox, coy, coz, cix, ciy, ciz, t1, x1);
                    slab_range(q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, cox, coy, coz, cix, ciy, ciz, t2, x2);
                    slab_range(q5.z, q5.w, q6.x, q6.y, q6.z, q6.w, cox, coy, coz, cix, ciy, ciz, t3, x3);
                    r0 = __float_as_uint(q3.x); r1 = __float_as_uint(q3.y); r2 = __float_as_uint(q3.z); r3 = __float_as_uint(q3.w);
                }
                unsigned k0 = relaxed_pass(t0, x0) && t0 <= prune_t ? __float_as_uint(t0) : 0xFFFFFFFFu;
                unsigned k1 = relaxed_pass(t1, x1) && t1 <= prune_t ? __float_as_uint(t1) : 0xFFFFFFFFu;
                unsigned k2 = r2 != REF_NONE && relaxed_pass(t2, x2) && t2 <= prune_t ? __float_as_uint(t2) : 0xFFFFFFFFu;
                unsigned k3 = r3 != REF_NONE && relaxed_pass(t3, x3) && t3 <= prune_t ? __float_as_uint(t3) : 0xFFFFFFFFu;
                auto order = [](unsigned &ka, unsigned &ra, unsigned &kb, unsigned &rb) {
                    const bool sw = kb < ka;
                    const unsigned kl = sw ? kb : ka, kh = sw ? ka : kb, rl = sw ? rb : ra, rh = sw ? ra : rb;
                    ka = kl; kb = kh; ra = rl; rb = rh;
                };
                order(k0, r0, k1, r1);
                order(k2, r2, k3, r3);
                order(k0, r0, k2, r2);
                order(k1, r1, k3, r3);
                order(k1, r1, k2, r2);
                if(k0 == 0xFFFFFFFFu) {
                    have = false;
                } else {
                    if(k3 != 0xFFFFFFFFu) push(r3, k3);
                    if(k2 != 0xFFFFFFFFu) push(r2, k2);
                    if(k1 != 0xFFFFFFFFu) push(r1, k1);
                    ref = r0;
                    tn = __uint_as_float(k0);
                }
            } else if(serve_inst && is_inst) {
                // the instance's box is its scene leaf box, whose EXACT test decides whether
                // anything inside the instance is a candidate (world_lo / world_hi of InstAccel)
                bool clear;
                float tw;
                bool pass = slab_fast(q5.z, q5.w, q6.x, q6.y, q6.z, q6.w, cox, coy, coz, cix, ciy, ciz, tw, clear);
                if(pass && !clear) pass = slab_test(q5.z, q5.w, q6.x, q6.y, q6.z, q6.w, cox, coy, coz, cdx, cdy, cdz, tw);
                if(!pass) {
                    have = false;
                } else {
                    // entry into the instance: rows of world-to-local, then the mesh root
                    cox = (q0.x * wox + q0.y * woy) + (q0.z * woz + q0.w);
                    coy = (q1.x * wox + q1.y * woy) + (q1.z * woz + q1.w);
                    coz = (q2.x * wox + q2.y * woy) + (q2.z * woz + q2.w);
                    cdx = (q0.x * wdx + q0.y * wdy) + q0.z * wdz;
                    cdy = (q1.x * wdx + q1.y * wdy) + q1.z * wdz;
                    cdz = (q2.x * wdx + q2.y * wdy) + q2.z * wdz;
                    cix = fast_rcp(cdx); ciy = fast_rcp(cdy); ciz = fast_rcp(cdz);
                    in_mesh = true;
                    inst_tw = tw;
                    cur_inst = (int)((idx - sc.hdr.inst_base) >> 1);
                    ref = __float_as_uint(q3.z);  // wide mesh root; its own box is implied by the boxes below it
                    if(has_subnormal_component(cdx, cdy, cdz)) {  // rule (5): give the whole ray up
                        uncertain = true;
                        have = false;
                        sp = 0;
                    }
                }
            } else if(serve_leaf && is_leaf) {
                // triangle leaf: q0, q1.xy parent box; then v0, e1, e2.  The first leaf of a pair hands
                // over to the second one, in the next slot, without going through the stack.
                have = pair;
                if(pair) ref = (REF_TRIANGLE << 30) | (idx + 1u);
                auto test = [&](float v0x, float v0y, float v0z, float e1x, float e1y, float e1z, float e2x, float e2y, float e2z, int tri) {
                    // intersect_triangle, shader/pathtracing.glsl:224-249
                    const float sx = cox - v0x, sy = coy - v0y, sz = coz - v0z;
                    const float s1x = cdy * e2z - e2y * cdz, s1y = cdz * e2x - e2z * cdx, s1z = cdx * e2y - e2x * cdy;
                    const float s2x = sy * e1z - e1y * sz, s2y = sz * e1x - e1z * sx, s2z = sx * e1y - e1x * sy;
                    const float k = 1.0f / (s1x * e1x + s1y * e1y + s1z * e1z);
                    const float t = (s2x * e2x + s2y * e2y + s2z * e2z) * k;
                    const float u = (s1x * sx + s1y * sy + s1z * sz) * k;
                    const float v = (s2x * cdx + s2y * cdy + s2z * cdz) * k;
                    if(RT_RAY_MINDIS <= t && 0.0f <= u && 0.0f <= v && u + v <= 1.0f) {
                        if(t < best.t) {
                            // candidate only if the reference would have reached this leaf
                            bool clear;
                            float t0;
                            bool pass = slab_fast(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, cox, coy, coz, cix, ciy, ciz, t0, clear);
                            if(pass && !clear) {
                                float te;
                                pass = slab_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, cox, coy, coz, cdx, cdy, cdz, te);
                            }
                            if(pass) {
                                second_t = fminf(second_t, best.t);
                                best.t = t; best.u = u; best.v = v;
                                best.tri = tri;
                                best.inst = cur_inst;
                                best_gate = fmaxf(t0, inst_tw);
                                if(any_hit) {
                                    sp = 0;
                                    have = false;
                                }
                            }
                        } else {
                            second_t = fminf(second_t, t);
                        }
                    }
                };
                test(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, __float_as_int(q3.w));
            }
        }
    }
#ifdef RT_TRACE_STATS
    unsigned long long *dbg = reinterpret_cast<unsigned long long *>(sc.stats_dbg + 4);
    if(RT_TRACE_STATS == 2) {  // wave-level: iterations, iterations that ran the leaf / instance code
        atomicAdd(dbg + 0, st_iter);
        atomicAdd(dbg + 1, st_leaf_runs);
        atomicAdd(dbg + 2, st_inst_runs);
    } else if(RT_TRACE_STATS == 3) {  // steps by level: mesh wide nodes, scene wide nodes, instance entries
        atomicAdd(dbg + 0, st_inner);
        atomicAdd(dbg + 1, st_scene);
        atomicAdd(dbg + 2, st_inst);
    } else {
        atomicAdd(dbg + 0, st_slots);
        atomicAdd(dbg + 1, st_inner + st_inst);
        atomicAdd(dbg + 2, st_leaf);
    }
#endif
}

}  // namespace rta
