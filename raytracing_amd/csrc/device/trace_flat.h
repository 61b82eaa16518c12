// Ordered closest-hit / occlusion traversal over ONE tree for the whole scene.
//
// trace_fast.h walks the reference's two levels: scene nodes in world space, an instance entry
// that moves the ray into the instance's space, then the mesh's nodes.  Of the 21 steps a
// path ray of the reference's demo scene takes there, 5.5 are scene nodes and instance
// entries.  Here every (instance, triangle) pair is a leaf of one four-wide tree built in
// WORLD space (host: build_accel_flat); the ray stays in world space while it walks, and is
// moved into an instance's space only where a triangle is tested -- with the same arithmetic,
// so the test itself is the reference's, bit for bit.
//
// Why this still returns the reference's answer: trace_fast.h's argument needs (1) culling that
// never drops a CANDIDATE (a triangle whose reference parent box passes in the instance's space
// and whose instance's scene leaf box passes in world space), (2) a hit accepted only for a
// candidate, (3) pruning beyond the near-tie band only, (4) the runner-up bookkeeping.  (2)-(4)
// are kept as they are: a leaf record carries the exact parent box, the instance record the
// exact scene leaf box, and both are tested with the reference's IEEE slab test before a closer
// hit is accepted.  For (1) the tree's boxes must be met by every ray for which the reference
// could accept a hit on the leaf's triangle.  The reference accepts when the barycentrics it
// computes for the local ray o_l = fl(W o), d_l = fl(W d) (W: the instance's world-to-local
// matrix) lie in the unit simplex.  Their rounding error is g |o_l - v0| / (h sin a) for a ray
// that meets the triangle's plane under the angle a (h: the triangle's height, g a few u), i.e.
// the plane is crossed within g |o_l - v0| / sin a of the triangle -- and a ray that crosses a
// plane under the angle a that close to a triangle passes the triangle itself within
// g |o_l - v0| / cos a, however flat a is: the local ray misses the triangle by a few u of its
// distance from the origin at most.  With A = W^-1:
//     o + t d = A p_l - A (e_o + t e_d),   |e_o| + t |e_d| <= 4 u |W| (|o| + t |d|)
// so the leaf's world box is the box of the triangle's three vertices under A (computed in
// double, rounded outwards; never larger than the box of the eight corners of the reference
// parent box under A, which it is cut against), and what remains is an ABSOLUTE error of a few
// u cond (|o| + t|d|), cond = || |A| |W| ||.  It is covered per ray: every box is widened by
//     pad = flat_pad * (|o|_1 + flat_reach),   flat_pad = 64 u max cond over the instances,
// flat_reach = the largest |coordinate| sum of the scene box (t |d| <= |o| + reach for any hit),
// at no cost per node: near planes are met from the origin moved by pad along the ray's
// direction signs, far planes from the origin moved the other way.  The host builds no flat
// tree when an instance's matrix is singular or cond exceeds 256.
// Pruning by entry distance assumes, as trace_fast.h does for the parent box, that the distance
// the reference computes for a hit is not SMALLER than the ray's entry into the padded box of
// what it hit by more than the near-tie band (0.1 % + 8e-6 (|o| + extent)); a hit found the other
// way round (entry distance of the parent box beyond the hit) sends the ray to the
// reference-order kernel.
//
// The same padding pays for cheaper box arithmetic.  A ray never changes space while it walks, so
// the slab distance (p - o) / d of a grid plane p = origin + step * scale is evaluated as
//     step * (scale / d) + (origin / d - o / d)
// with o / d made once per ray and the two node terms once per node and axis: a conversion and
// one multiply-add per plane instead of a conversion, the decode, a subtraction and a product.
// Its rounding moves a plane by at most 3 u (|o| + reach) -- inside pad.  All of it stays finite:
// |1/d| is capped at 2^80, the host refuses coordinates beyond 1e12, and a ray whose largest
// direction component lies outside [2^-40, 2^40] or whose origin lies beyond 2^40 is walked
// without culling (the triangle tests decide, as ever).
#pragma once
#include "trace_fast.h"

namespace rta {

template <class Feed = WaveRayFeed, class Source, class Sink>
__device__ __forceinline__ void trace_flat(const FastScene &sc, Source &src, Sink &sink, unsigned *next_ray,
                                           unsigned n_rays, uint2 *s_stack) {
    // (a source whose refill is dear -- two dependent trips to memory and the hand-over of a light sample -- may ask
    // for fewer of them: Source::REFILL_BELOW)
    constexpr int REFILL_BELOW = refill_below<Source>(RT_FAST_REFILL);
#ifndef RT_FLAT_LEAF_BATCH
#define RT_FLAT_LEAF_BATCH 12
#endif
    constexpr int LEAF_BATCH = RT_FLAT_LEAF_BATCH, SERVE_ALL_BELOW = RT_SERVE_ALL_BELOW;
    const unsigned tid = threadIdx.x;
    const unsigned gtid = blockIdx.x * blockDim.x + tid;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;

    bool active = false, have = false, exhausted = false, any_hit = false;
    bool uncertain = false;  // rule (5) of trace_fast.h: this ray is the reference-order kernel's
    // A ray that has ended keeps its result in the lane until the wave refills: handing results over where each ray
    // ends runs that code -- for an occlusion query: fetch the light sample, the throughput and the accumulator, add,
    // store, with a wait on memory -- for the one to three lanes that end in a step, in almost every step; at the
    // refill it runs once for the two dozen lanes that have ended since.
    bool pending = false, pending_ambiguous = false;
    Feed feed(n_rays);
    unsigned ray_idx = 0;
    int sp = 0;
    float wox = 0, woy = 0, woz = 0, wdx = 0, wdy = 0, wdz = 0;  // world ray
    float wix = 0, wiy = 0, wiz = 0;                             // 1 / direction
    float cnx = 0, cny = 0, cnz = 0, cfx = 0, cfy = 0, cfz = 0;  // -(origin as near / far planes see it) / direction
    LaneHit best;
    best.t = RT_INF; best.u = 0; best.v = 0; best.tri = -1; best.inst = -1;
    float second_t = RT_INF, abs_slack = 0.0f;
    float best_gate = 0.0f;  // for the closest hit: the larger entry distance of its parent box and its scene leaf box
    unsigned ref = 0;
    float tn = 0;

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
                    const float o1 = fabsf(wox) + fabsf(woy) + fabsf(woz);
                    const float dm = fmaxf(fabsf(wdx), fmaxf(fabsf(wdy), fabsf(wdz)));
                    // The slab distance of a plane p is (p - o) / d = p * (1/d) - o * (1/d): with the ray fixed for
                    // its whole walk, the second product is made once here and a plane costs one multiply-add.
                    // 1/d is kept finite (|1/d| <= 2^80; a component below 2^-80 of a direction whose largest one
                    // is at least 2^-40 can only matter at distances no hit has), so that no product is inf - inf.
                    wix = fminf(fmaxf(fast_rcp(wdx), -0x1p80f), 0x1p80f);
                    wiy = fminf(fmaxf(fast_rcp(wdy), -0x1p80f), 0x1p80f);
                    wiz = fminf(fmaxf(fast_rcp(wdz), -0x1p80f), 0x1p80f);
                    const float pad = sc.flat_pad * (o1 + sc.flat_reach);
                    cnx = -((wox + copysignf(pad, wdx)) * wix); cfx = -((wox - copysignf(pad, wdx)) * wix);
                    cny = -((woy + copysignf(pad, wdy)) * wiy); cfy = -((woy - copysignf(pad, wdy)) * wiy);
                    cnz = -((woz + copysignf(pad, wdz)) * wiz); cfz = -((woz - copysignf(pad, wdz)) * wiz);
                    if(!(dm >= 0x1p-40f && dm <= 0x1p40f && o1 <= 0x1p40f)) {
                        // a ray far outside the numbers above: no culling at all (every box passes), the
                        // triangle tests decide as ever
                        wix = wiy = wiz = 0.0f;
                        cnx = cny = cnz = -1.0f;
                        cfx = cfy = cfz = 3e38f;
                    }
                    best.t = RT_INF; best.u = 0; best.v = 0; best.tri = -1; best.inst = -1;
                    second_t = RT_INF;
                    abs_slack = 1e-6f * (o1 + sc.hdr.extent);
                    active = true;
                    sp = 0;
                    tn = 0.0f;
                    // a NaN component: a certain miss (trace_core.h); an infinite origin or a zero direction:
                    // every triangle test of the reference computes NaN coordinates and fails
                    have = !ray_has_nan(o, d) && o1 < 3e38f && dm > 0.0f;
                    uncertain = false;
                    if(have && has_subnormal_component(wdx, wdy, wdz)) {
                        uncertain = true;
                        have = false;
                    }
                    ref = sc.flat_root_ref;
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
            const float prune_t = best.t * 1.001f + 8.0f * abs_slack;
            if(active && !have) {
                if(sp > 0) {
                    --sp;
                    const uint2 e = sp < FAST_LDS_STACK ? s_stack[sp * FAST_BLOCK + tid]
                                                        : sc.spill[(size_t)(sp - FAST_LDS_STACK) * nthreads + gtid];
                    ref = e.x;
                    tn = __uint_as_float(e.y);
                    have = !(tn > prune_t);
                } else {
                    // (the same decision as at the end of a ray in trace_ordered)
                    const bool odd = best_gate > best.t * 1.0005f + 4.0f * abs_slack;
                    const bool ambiguous = uncertain || (!any_hit && best.tri >= 0 && (odd || second_t <= best.t || second_t < best_gate * 1.000004f));
                    if(ambiguous && odd) best.t = RT_INF;
                    if(uncertain) {
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

            // wide nodes always; triangle leaves once enough lanes hold one (or few hold a node)
            const bool is_leaf = have && (ref >> 30) == REF_TRIANGLE;
            const bool is_inner = have && !is_leaf;
            const bool pair = is_leaf && (ref & REF_PAIR) != 0u;
            const unsigned idx = is_leaf ? (ref & REF_SLOT_MASK) : (ref & REF_INDEX_MASK);
            const int n_inner = __popcll(__ballot(is_inner));
            const bool serve_leaf = n_inner < SERVE_ALL_BELOW || __popcll(__ballot(is_leaf)) >= LEAF_BATCH;
            const bool go = is_inner || (is_leaf && serve_leaf);

            float4 q0, q1, q2, q3, q4, q5, q6;
            unsigned inst = 0;
            if(go) {
                const char *base = sc.flat_buf + ((size_t)idx << 6);
                const float4 *rp = reinterpret_cast<const float4 *>(base);
                q0 = rp[0]; q1 = rp[1]; q2 = rp[2]; q3 = rp[3];
                if(is_leaf) {
                    q4 = rp[4]; q5 = rp[5]; q6 = rp[6];
                    inst = *reinterpret_cast<const unsigned *>(base + 112);
                }
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
            if(is_inner) {
                // a QNode, decoded as in trace_ordered; near planes against the origin moved forwards by
                // the padding, far planes against the origin moved backwards
                float t0, t1, t2, t3, x0, x1, x2, x3;
                const float sx = q0.w, sy = q1.x, sz = q1.y;
                const unsigned lx = __float_as_uint(q2.z), ly = __float_as_uint(q2.w), lz = __float_as_uint(q3.x);
                const unsigned hx = __float_as_uint(q3.y), hy = __float_as_uint(q3.z), hz = __float_as_uint(q3.w);
                const bool ngx = (int)__float_as_uint(wix) < 0, ngy = (int)__float_as_uint(wiy) < 0, ngz = (int)__float_as_uint(wiz) < 0;
                const unsigned nx = ngx ? hx : lx, fx = ngx ? lx : hx;
                const unsigned ny = ngy ? hy : ly, fy = ngy ? ly : hy;
                const unsigned nz = ngz ? hz : lz, fz = ngz ? lz : hz;
                // per node and axis: distance per grid step, and the distances of the grid's origin
                const float ax = sx * wix, ay = sy * wiy, az = sz * wiz;
                const float bnx = __builtin_fmaf(q0.x, wix, cnx), bny = __builtin_fmaf(q0.y, wiy, cny), bnz = __builtin_fmaf(q0.z, wiz, cnz);
                const float bfx = __builtin_fmaf(q0.x, wix, cfx), bfy = __builtin_fmaf(q0.y, wiy, cfy), bfz = __builtin_fmaf(q0.z, wiz, cfz);
#define RT_Q(packed, i, a, b) __builtin_fmaf((float)(((packed) >> (8 * (i))) & 0xFFu), a, b)
#define RT_QBOX(i, T_IN, T_OUT)                                                                                  \
    T_IN = fmaxf(0.0f, fmaxf(fmaxf(RT_Q(nx, i, ax, bnx), RT_Q(ny, i, ay, bny)), RT_Q(nz, i, az, bnz)));          \
    T_OUT = fminf(RT_INF, fminf(fminf(RT_Q(fx, i, ax, bfx), RT_Q(fy, i, ay, bfy)), RT_Q(fz, i, az, bfz)))
                RT_QBOX(0, t0, x0);
                RT_QBOX(1, t1, x1);
                RT_QBOX(2, t2, x2);
                RT_QBOX(3, t3, x3);
#undef RT_QBOX
#undef RT_Q
                unsigned r0 = __float_as_uint(q1.z), r1 = __float_as_uint(q1.w), r2 = __float_as_uint(q2.x), r3 = __float_as_uint(q2.y);
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
            } else if(serve_leaf && is_leaf) {
                // FlatLeaf: q0, q1.xy parent box; v0, e1, e2, triangle index; q4..q6 rows of world-to-local.
                // The first leaf of a pair hands over to the second one, two slots on.
                have = pair;
                if(pair) ref = (REF_TRIANGLE << 30) | (idx + 2u);
                // The instance's scene leaf box is asked for now, for every lane of the step: where a closer hit is
                // found it is needed at once, and asking for it there made the wave wait on memory for one or two lanes
                // in almost every leaf step.
                const float4 *ip = reinterpret_cast<const float4 *>(sc.scene_buf + ((size_t)(sc.hdr.inst_base + 2u * inst) << 6));
                const float4 b5 = ip[5], b6 = ip[6];
                // the ray in the instance's space, as the reference forms it (pathtracing.glsl:279-281;
                // the same expressions as the instance entry of trace_ordered)
                const float lox = (q4.x * wox + q4.y * woy) + (q4.z * woz + q4.w);
                const float loy = (q5.x * wox + q5.y * woy) + (q5.z * woz + q5.w);
                const float loz = (q6.x * wox + q6.y * woy) + (q6.z * woz + q6.w);
                const float ldx = (q4.x * wdx + q4.y * wdy) + q4.z * wdz;
                const float ldy = (q5.x * wdx + q5.y * wdy) + q5.z * wdz;
                const float ldz = (q6.x * wdx + q6.y * wdy) + q6.z * wdz;
                if(has_subnormal_component(ldx, ldy, ldz)) {  // rule (5): give the whole ray up
                    uncertain = true;
                    have = false;
                    sp = 0;
                }
                const float v0x = q1.z, v0y = q1.w, v0z = q2.x, e1x = q2.y, e1y = q2.z, e1z = q2.w, e2x = q3.x, e2y = q3.y, e2z = q3.z;
                // intersect_triangle, shader/pathtracing.glsl:224-249
                const float sx = lox - v0x, sy = loy - v0y, sz = loz - v0z;
                const float s1x = ldy * e2z - e2y * ldz, s1y = ldz * e2x - e2z * ldx, s1z = ldx * e2y - e2x * ldy;
                const float s2x = sy * e1z - e1y * sz, s2y = sz * e1x - e1z * sx, s2z = sx * e1y - e1x * sy;
                const float k = 1.0f / (s1x * e1x + s1y * e1y + s1z * e1z);
                const float t = (s2x * e2x + s2y * e2y + s2z * e2z) * k;
                const float u = (s1x * sx + s1y * sy + s1z * sz) * k;
                const float v = (s2x * ldx + s2y * ldy + s2z * ldz) * k;
                if(!uncertain && RT_RAY_MINDIS <= t && 0.0f <= u && 0.0f <= v && u + v <= 1.0f) {
                    if(t < best.t) {
                        // a candidate only if the reference would have reached this leaf: its parent's box in
                        // the instance's space and the instance's scene leaf box in world space, exactly
                        // (as in trace_ordered: the relaxed test first, the twelve IEEE divisions of the exact
                        // ones only when it neither fails nor passes with its margin to spare -- this code runs
                        // for the one or two lanes of a leaf step that found a closer hit; fresh reciprocals,
                        // since the per-ray ones are capped for the walk)
                        float te, tw = 0.0f;
                        bool clear;
                        bool pass = slab_fast(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, lox, loy, loz, fast_rcp(ldx), fast_rcp(ldy), fast_rcp(ldz), te, clear);
                        if(pass && !clear) {
                            float exact;
                            pass = slab_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, lox, loy, loz, ldx, ldy, ldz, exact);
                        }
                        if(pass) {
                            pass = slab_fast(b5.z, b5.w, b6.x, b6.y, b6.z, b6.w, wox, woy, woz, fast_rcp(wdx), fast_rcp(wdy), fast_rcp(wdz), tw, clear);
                            if(pass && !clear) {
                                float exact;
                                pass = slab_test(b5.z, b5.w, b6.x, b6.y, b6.z, b6.w, wox, woy, woz, wdx, wdy, wdz, exact);
                            }
                        }
                        if(pass) {
                            second_t = fminf(second_t, best.t);
                            best.t = t; best.u = u; best.v = v;
                            best.tri = __float_as_int(q3.w);
                            best.inst = (int)inst;
                            best_gate = fmaxf(te, tw);
                            if(any_hit) {
                                sp = 0;
                                have = false;
                            }
                        }
                    } else {
                        second_t = fminf(second_t, t);
                    }
                }
            }
        }
    }
}

}  // namespace rta
