// Coherent rays through the flattened tree -- the camera rays of an 8x8-pixel tile, or what leaves such a tile in nearly
// one direction -- one WAVE per group of 64 walking ONE node sequence.
//
// trace_flat.h gives every lane a ray of its own: lanes hold different nodes, some a wide node, some a leaf, some
// nothing until the wave refills, and a vector instruction of that kernel does work in 26 to 37 of its 64 lanes.  The
// camera rays of a tile are as alike as rays get -- one origin, directions within a quarter of a degree -- so here the
// 64 lanes of a wave visit the SAME node in every step: the node is named by a wave-uniform reference (its record is
// fetched once for the wave), every lane tests its own ray against the node's four child boxes, a child is visited when
// ANY lane's ray passes it, and a leaf's triangle is tested by all lanes at once.  The wave's stack of deferred nodes
// is wave-uniform too: it lives in two vector registers, lane j holding entry j (v_writelane / v_readlane), beside the
// smallest entry distance any lane found for the node; an entry is dropped when it comes back if that distance lies
// beyond what every lane still accepts.  No LDS, no spill area, no refill inside a walk.
//
// The answer is trace_flat.h's.  Culling only decides which leaves a lane gets to see: a lane here sees every leaf
// its own ray would have reached there (its own box test passes => the wave visits) and some it would not have; a hit
// is still accepted only after the exact tests of the triangle's reference parent box and of the instance's scene leaf
// box, the runner-up distance is tracked the same way, and the same rays are handed to the reference-order launch as
// undecided.  What a lane computes for a leaf is the code of trace_flat.h's leaf step, statement for statement.
#pragma once
#include "shade_vertex.h"
#include "trace_flat.h"

namespace rta {

constexpr int PACKET_STACK = 64;  // entries of the wave's stack: one per lane of the two registers that hold it

// minimum / maximum over the 64 lanes of a wave (every lane must be active), as wave-uniform values
__device__ __forceinline__ unsigned wave_umin(unsigned v) {
    // prefix minima inside each row of 16 lanes (row_shr 1, 2, 4, 8), then lane 15 of a row into the next row
    // (row_bcast15, rows 1 and 3), then lane 31 into rows 2 and 3 (row_bcast31): lane 63 holds the minimum
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_umax(unsigned v) { return ~wave_umin(~v); }

// Rays: bool make(unsigned item, unsigned &slot, V3 &o, V3 &d) -- the ray of work item `item` (a lane of a group of 64)
// and the path slot it belongs to; false: the item names no ray.
// Sink: void tile(unsigned slot, bool pixel, const LaneHit &h, bool ambiguous), called by ALL lanes of the wave at the end
// of a group's walk (pixel: the lane has a ray; the other arguments mean something only then), and
// void finish(), called by all lanes when the wave has no group left.
// n_items: a multiple of 64.
struct CameraRays {  // the camera rays of the slots [base, base + n_items), shader/pathtracing.glsl:604-608
    const DrawParams &dp;
    unsigned base;
    __device__ bool make(unsigned item, unsigned &slot, V3 &o, V3 &d) const {
        slot = base + item;
        unsigned px, py, k;
        if(!state_pixel(dp, slot, px, py, k)) return false;
        camera_ray(dp.p, dp.samples[k].jitter[0], dp.samples[k].jitter[1], dp.tan_half_fov, px, py, o, d);
        return true;
    }
};

// ANY_HIT: occlusion queries -- a lane's ray ends at the first hit it accepts.
template <bool ANY_HIT = false, class Rays, class Sink>
__device__ __forceinline__ void trace_flat_packets(const FastScene &sc, const Rays &rays, Sink &sink, unsigned *next_tile,
                                                   unsigned n_items) {
    WaveRayFeed feed(n_items);  // hands out whole groups: batch sizes and n_items are multiples of 64
    const unsigned lane = threadIdx.x & 63u;
    for(;;) {
        unsigned item = 0;
        const bool got = feed.take(true, next_tile, n_items, item);
        if(!__any(got)) break;
        // ---- this lane's ray and what trace_flat.h derives from a ray
        unsigned slot = 0;
        V3 ro = v3(0.0f), rd = v3(0.0f, 0.0f, 1.0f);
        const bool pixel = got && rays.make(item, slot, ro, rd);
        const float wox = ro.x, woy = ro.y, woz = ro.z, wdx = rd.x, wdy = rd.y, wdz = rd.z;
        const float o1 = fabsf(wox) + fabsf(woy) + fabsf(woz);
        const float dm = fmaxf(fabsf(wdx), fmaxf(fabsf(wdy), fabsf(wdz)));
        float wix = fminf(fmaxf(fast_rcp(wdx), -0x1p80f), 0x1p80f);
        float wiy = fminf(fmaxf(fast_rcp(wdy), -0x1p80f), 0x1p80f);
        float wiz = fminf(fmaxf(fast_rcp(wdz), -0x1p80f), 0x1p80f);
        const float pad = sc.flat_pad * (o1 + sc.flat_reach);
        float cnx = -((wox + copysignf(pad, wdx)) * wix), cfx = -((wox - copysignf(pad, wdx)) * wix);
        float cny = -((woy + copysignf(pad, wdy)) * wiy), cfy = -((woy - copysignf(pad, wdy)) * wiy);
        float cnz = -((woz + copysignf(pad, wdz)) * wiz), cfz = -((woz - copysignf(pad, wdz)) * wiz);
        if(!(dm >= 0x1p-40f && dm <= 0x1p40f && o1 <= 0x1p40f)) {
            wix = wiy = wiz = 0.0f;
            cnx = cny = cnz = -1.0f;
            cfx = cfy = cfz = 3e38f;
        }
        const float abs_slack = 1e-6f * (o1 + sc.hdr.extent);
        float o3[3] = {wox, woy, woz}, d3[3] = {wdx, wdy, wdz};
        bool uncertain = false;
        // live: this lane's ray takes part in the walk (a NaN component, an infinite origin or a zero direction is a
        // certain miss, trace_core.h; a zero or denormal component sends the ray to the reference-order launch, rule (5))
        bool live = pixel && !ray_has_nan(o3, d3) && o1 < 3e38f && dm > 0.0f;
        if(live && has_subnormal_component(wdx, wdy, wdz)) {
            uncertain = true;
            live = false;
        }
        LaneHit best;
        best.t = RT_INF; best.u = 0; best.v = 0; best.tri = -1; best.inst = -1;
        float second_t = RT_INF, best_gate = 0.0f;

        // ---- the wave's walk
        int st_ref = 0, st_key = 0;  // the stack: lane j holds entry j
        int sp = 0;
        unsigned ref = sc.flat_root_ref;
        bool have = __any(live);
        unsigned max_prune = 0xFFFFFFFFu;  // the largest distance any lane still accepts (bits of a float >= 0); stale = larger
        bool prune_dirty = false;
        while(have) {
            ref = (unsigned)__builtin_amdgcn_readfirstlane((int)ref);
            const float prune_t = best.t * 1.001f + 8.0f * abs_slack;
            if((ref >> 30) == REF_TRIANGLE) {
                // ---- a leaf: every live lane runs trace_flat.h's leaf step on it
                const unsigned idx = ref & REF_SLOT_MASK;
                const bool pair = (ref & REF_PAIR) != 0u;
                const char *base = sc.flat_buf + ((size_t)idx << 6);
                const float4 *rp = reinterpret_cast<const float4 *>(base);
                const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3], q4 = rp[4], q5 = rp[5], q6 = rp[6];
                const unsigned inst = *reinterpret_cast<const unsigned *>(base + 112);
                const float4 *ip = reinterpret_cast<const float4 *>(sc.scene_buf + ((size_t)(sc.hdr.inst_base + 2u * inst) << 6));
                const float4 b5 = ip[5], b6 = ip[6];
                if(live) {
                    const float lox = (q4.x * wox + q4.y * woy) + (q4.z * woz + q4.w);
                    const float loy = (q5.x * wox + q5.y * woy) + (q5.z * woz + q5.w);
                    const float loz = (q6.x * wox + q6.y * woy) + (q6.z * woz + q6.w);
                    const float ldx = (q4.x * wdx + q4.y * wdy) + q4.z * wdz;
                    const float ldy = (q5.x * wdx + q5.y * wdy) + q5.z * wdz;
                    const float ldz = (q6.x * wdx + q6.y * wdy) + q6.z * wdz;
                    if(has_subnormal_component(ldx, ldy, ldz)) {  // rule (5): give the whole ray up
                        uncertain = true;
                        live = false;
                    }
                    const float v0x = q1.z, v0y = q1.w, v0z = q2.x, e1x = q2.y, e1y = q2.z, e1z = q2.w, e2x = q3.x, e2y = q3.y, e2z = q3.z;
                    // intersect_triangle, shader/pathtracing.glsl:224-249
                    const float sx = lox - v0x, sy = loy - v0y, sz = loz - v0z;
                    const float s1x = ldy * e2z - e2y * ldz, s1y = ldz * e2x - e2z * ldx, s1z = ldx * e2y - e2x * ldy;
                    const float s2x = sy * e1z - e1y * sz, s2y = sz * e1x - e1z * sx, s2z = sx * e1y - e1x * sy;
                    const float kk = 1.0f / (s1x * e1x + s1y * e1y + s1z * e1z);
                    const float t = (s2x * e2x + s2y * e2y + s2z * e2z) * kk;
                    const float u = (s1x * sx + s1y * sy + s1z * sz) * kk;
                    const float v = (s2x * ldx + s2y * ldy + s2z * ldz) * kk;
                    if(!uncertain && RT_RAY_MINDIS <= t && 0.0f <= u && 0.0f <= v && u + v <= 1.0f) {
                        if(t < best.t) {
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
                                prune_dirty = true;
                                if(ANY_HIT) live = false;  // (occluded: nothing more to ask)
                            }
                        } else {
                            second_t = fminf(second_t, t);
                        }
                    }
                }
                if(pair) {
                    ref = (REF_TRIANGLE << 30) | (idx + 2u);  // the second leaf of the pair, two slots on
                    continue;
                }
                have = false;
            } else {
                // ---- a wide node: every lane tests its ray against the four child boxes (the decode of trace_flat.h)
                const unsigned idx = ref & REF_INDEX_MASK;
                const float4 *rp = reinterpret_cast<const float4 *>(sc.flat_buf + ((size_t)idx << 6));
                const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
                float t0, t1, t2, t3, x0, x1, x2, x3;
                const float sx = q0.w, sy = q1.x, sz = q1.y;
                const unsigned lx = __float_as_uint(q2.z), ly = __float_as_uint(q2.w), lz = __float_as_uint(q3.x);
                const unsigned hx = __float_as_uint(q3.y), hy = __float_as_uint(q3.z), hz = __float_as_uint(q3.w);
                const bool ngx = (int)__float_as_uint(wix) < 0, ngy = (int)__float_as_uint(wiy) < 0, ngz = (int)__float_as_uint(wiz) < 0;
                const unsigned nx = ngx ? hx : lx, fx = ngx ? lx : hx;
                const unsigned ny = ngy ? hy : ly, fy = ngy ? ly : hy;
                const unsigned nz = ngz ? hz : lz, fz = ngz ? lz : hz;
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
                // the four references are the same in every lane: as scalars, so that choosing among them costs no vector work
                unsigned r0 = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(q1.z)), r1 = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(q1.w));
                unsigned r2 = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(q2.x)), r3 = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(q2.y));
                const bool p0 = live && relaxed_pass(t0, x0) && t0 <= prune_t, p1 = live && relaxed_pass(t1, x1) && t1 <= prune_t;
                const bool p2 = r2 != REF_NONE && live && relaxed_pass(t2, x2) && t2 <= prune_t, p3 = r3 != REF_NONE && live && relaxed_pass(t3, x3) && t3 <= prune_t;
                const bool a0 = __any(p0), a1 = __any(p1), a2 = __any(p2), a3 = __any(p3);
                const int n_pass = (int)a0 + (int)a1 + (int)a2 + (int)a3;
                if(n_pass == 0) {
                    have = false;
                } else if(n_pass == 1) {
                    // the common case deep in the tree: one child, nothing to order, nothing to defer
                    ref = a0 ? r0 : (a1 ? r1 : (a2 ? r2 : r3));
                } else {
                    // per child: the smallest entry distance among the lanes whose ray passes it (no lane: 0xFFFFFFFF)
                    unsigned k0 = a0 ? wave_umin(p0 ? __float_as_uint(t0) : 0xFFFFFFFFu) : 0xFFFFFFFFu;
                    unsigned k1 = a1 ? wave_umin(p1 ? __float_as_uint(t1) : 0xFFFFFFFFu) : 0xFFFFFFFFu;
                    unsigned k2 = a2 ? wave_umin(p2 ? __float_as_uint(t2) : 0xFFFFFFFFu) : 0xFFFFFFFFu;
                    unsigned k3 = a3 ? wave_umin(p3 ? __float_as_uint(t3) : 0xFFFFFFFFu) : 0xFFFFFFFFu;
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
                    auto push = [&](unsigned r, unsigned key) {
                        if(sp < PACKET_STACK) {
                            if((int)lane == sp) {  // (one select each: lane sp takes the entry)
                                st_ref = (int)r;
                                st_key = (int)key;
                            }
                            ++sp;
                        } else {
                            *sc.error_flag = 1u;
                        }
                    };
                    if(k3 != 0xFFFFFFFFu) push(r3, k3);
                    if(k2 != 0xFFFFFFFFu) push(r2, k2);
                    if(k1 != 0xFFFFFFFFu) push(r1, k1);
                    ref = r0;
                }
            }
            // ---- the next deferred node that some lane still wants
            if(ANY_HIT && !have && !__any(live)) sp = 0;  // (every query of the group is answered)
            while(!have && sp > 0) {
                --sp;
                ref = (unsigned)__builtin_amdgcn_readlane(st_ref, sp);
                const unsigned key = (unsigned)__builtin_amdgcn_readlane(st_key, sp);
                if(__any(prune_dirty)) {
                    max_prune = wave_umax(live ? __float_as_uint(best.t * 1.001f + 8.0f * abs_slack) : 0u);
                    prune_dirty = false;
                }
                have = key <= max_prune;
            }
        }
        // ---- the end of every ray of the tile (the decision trace_flat.h makes when a lane's stack runs empty)
        bool ambiguous = false;
        if(pixel) {
            const bool odd = best_gate > best.t * 1.0005f + 4.0f * abs_slack;
            // (occlusion queries have no order dependence: only a ray given up by rule (5) is undecided)
            ambiguous = uncertain || (!ANY_HIT && best.tri >= 0 && (odd || second_t <= best.t || second_t < best_gate * 1.000004f));
            if(ambiguous && odd) best.t = RT_INF;
            if(uncertain) {
                best.t = RT_INF; best.tri = -1; best.inst = -1;
            }
        }
        sink.tile(slot, pixel, best, ambiguous);
    }
    sink.finish();
}

}  // namespace rta
