// Users of the two traversal cores: stand-alone closest hit (pt_pass_intersect), the path
// tracer's extend stage and its shadow stage.  Each exists in a reference-order form
// (trace_core.h, exact by construction) and an ordered form (trace_fast.h) that hands the
// rays it cannot decide (near-ties in t) to the reference-order form through a redo queue.
#include "kernels.h"
#include "path_state_io.h"
#include "shade_vertex.h"
#include "trace_fast.h"
#include "trace_flat.h"
#include "trace_packet.h"

namespace rta {
namespace {

// The ordered kernels must fit 4 waves per SIMD (128 VGPRs); one register more costs a quarter
// of the occupancy.
#ifndef RT_FAST_WAVES
#define RT_FAST_WAVES 4
#endif
#define RT_FAST_OCC __attribute__((amdgpu_waves_per_eu(RT_FAST_WAVES)))

// MODE of the ordered kernels: 0 = two levels with WNodes, 1 = two levels with QNodes, 2 = the flattened tree
template <int MODE, class Source, class Sink>
__device__ __forceinline__ void trace_fast_any(const FastScene &sc, Source &src, Sink &sink, unsigned *next_ray, unsigned n_rays,
                                               uint2 *s_stack) {
    if constexpr(MODE == 2) trace_flat(sc, src, sink, next_ray, n_rays, s_stack);
    else trace_ordered<MODE == 1>(sc, src, sink, next_ray, n_rays, s_stack);
}
#define RT_PICK_FAST(kernel) (sc.flat_buf ? kernel<2> : (sc.hdr.quantised ? kernel<1> : kernel<0>))

// ---- stand-alone closest hit -----------------------------------------------------------
// `redo`, when given, lists the ray indices to trace (the ordered kernel's undecided rays).
struct RaySource {
    const float *rays;
    const unsigned *redo;
    const pt_hit *provisional;  // with redo: what the ordered pass found for those rays
    __device__ float bound(unsigned idx) const { return redo ? provisional[redo[idx]].t : RT_INF; }
    __device__ bool load(unsigned idx, float o[3], float d[3], bool &any_hit) const {
        const unsigned r = redo ? redo[idx] : idx;
        // streamed once: keep it from displacing the scene in the caches
        const float *p = rays + (size_t)r * 6;
        o[0] = __builtin_nontemporal_load(p + 0);
        o[1] = __builtin_nontemporal_load(p + 1);
        o[2] = __builtin_nontemporal_load(p + 2);
        d[0] = __builtin_nontemporal_load(p + 3);
        d[1] = __builtin_nontemporal_load(p + 4);
        d[2] = __builtin_nontemporal_load(p + 5);
        any_hit = false;
        return true;
    }
};
__device__ __forceinline__ void write_hit(pt_hit *hits, unsigned r, const LaneHit &h) {
    pt_hit *x = hits + r;
    __builtin_nontemporal_store(h.inst, &x->instance);
    __builtin_nontemporal_store(h.tri, &x->triangle);
    __builtin_nontemporal_store(h.t, &x->t);
    __builtin_nontemporal_store(h.u, &x->u);
    __builtin_nontemporal_store(h.v, &x->v);
}
struct HitSink {
    pt_hit *hits;
    const unsigned *redo;
    __device__ void store(unsigned idx, const LaneHit &h) const { write_hit(hits, redo ? redo[idx] : idx, h); }
};
struct HitSinkFast {
    pt_hit *hits;
    unsigned *redo, *redo_count;
    __device__ void store(unsigned idx, const LaneHit &h, bool ambiguous) const {
        write_hit(hits, idx, h);  // (provisional for an ambiguous ray: bounds its second pass)
        if(ambiguous) redo[atomicAdd(redo_count, 1u)] = idx;
    }
};

__global__ __launch_bounds__(TRACE_BLOCK) void intersect_kernel(TraceScene sc, const float *rays, unsigned n,
                                                                const unsigned *redo, const unsigned *redo_count,
                                                                pt_hit *hits, unsigned *next_counter,
                                                                unsigned long long *retraced_stat) {
    __shared__ uint2 s_stack[LDS_STACK * TRACE_BLOCK];
    if(redo && blockIdx.x == 0 && threadIdx.x == 0 && *redo_count) atomicAdd(retraced_stat, (unsigned long long)*redo_count);
    RaySource src{rays, redo, hits};
    HitSink sink{hits, redo};
    if(redo) trace_persistent<SparseRayFeed>(sc, src, sink, next_counter, *redo_count, s_stack);
    else trace_persistent(sc, src, sink, next_counter, n, s_stack);
}

template <int MODE>
__global__ __launch_bounds__(FAST_BLOCK) RT_FAST_OCC void intersect_fast_kernel(FastScene sc, const float *rays, unsigned n,
                                                                    pt_hit *hits, unsigned *next_counter,
                                                                    unsigned *redo, unsigned *redo_count) {
    __shared__ uint2 s_stack[FAST_LDS_STACK * FAST_BLOCK];
    RaySource src{rays, nullptr, nullptr};
    HitSinkFast sink{hits, redo, redo_count};
    trace_fast_any<MODE>(sc, src, sink, next_counter, n, s_stack);
}

// ---- extend ------------------------------------------------------------------------------
struct ExtendSource {
    const unsigned *queue;
    const PathRay *rays;
    const PathHit *provisional;  // second pass over the ordered traversal's undecided rays: its hits, else null
    __device__ float bound(unsigned idx) const { return provisional ? provisional[queue ? queue[idx] : idx].t : RT_INF; }
    __device__ bool load(unsigned idx, float o[3], float d[3], bool &any_hit) const {
        const unsigned px = queue ? queue[idx] : idx;
        load_path_ray(rays, px, o, d);
        any_hit = false;
        return true;
    }
};
struct ExtendSink {
    const unsigned *queue;
    PathHit *hits;
    __device__ void store(unsigned idx, const LaneHit &h) const { write_path_hit(hits, queue ? queue[idx] : idx, h); }
};
struct ExtendSinkFast {
    const unsigned *queue;
    PathHit *hits;
    unsigned *redo, *redo_count;  // redo lists PIXEL slots, so the reference-order extend can take it as its queue
    __device__ void store(unsigned idx, const LaneHit &h, bool ambiguous) const {
        const unsigned px = queue ? queue[idx] : idx;
        write_path_hit(hits, px, h);  // (provisional for an ambiguous ray: bounds its second pass)
        if(ambiguous) redo[atomicAdd(redo_count, 1u)] = px;
    }
};

// second_pass: the queue lists the rays the ordered kernel left undecided, with its hits in place
__global__ __launch_bounds__(TRACE_BLOCK) void extend_kernel(TraceScene sc, const unsigned *queue,
                                                             const unsigned *count, const PathRay *rays,
                                                             PathHit *hits, unsigned *next_counter, bool second_pass) {
    __shared__ uint2 s_stack[LDS_STACK * TRACE_BLOCK];
    ExtendSource src{queue, rays, second_pass ? hits : nullptr};
    ExtendSink sink{queue, hits};
    if(second_pass) trace_persistent<SparseRayFeed>(sc, src, sink, next_counter, *count, s_stack);
    else trace_persistent(sc, src, sink, next_counter, *count, s_stack);
}

template <int MODE>
__global__ __launch_bounds__(FAST_BLOCK) RT_FAST_OCC void extend_fast_kernel(FastScene sc, const unsigned *queue,
                                                                 const unsigned *count, const PathRay *rays,
                                                                 PathHit *hits, unsigned *next_counter,
                                                                 unsigned *redo, unsigned *redo_count) {
    __shared__ uint2 s_stack[FAST_LDS_STACK * FAST_BLOCK];
    ExtendSource src{queue, rays, nullptr};
    ExtendSinkFast sink{queue, hits, redo, redo_count};
    trace_fast_any<MODE>(sc, src, sink, next_counter, *count, s_stack);
}

// ---- primary rays: generate + extend(0) in one launch --------------------------------------------
// The lane that is handed slot i makes that pixel's camera ray (shader/pathtracing.glsl:595-608),
// initialises the slot's path state and traces the ray, instead of a launch of its own writing
// all rays first and this one reading them back.
struct PrimarySource {
    const DrawParams &dp;
    unsigned base;  // work item i is slot base + i (a launch may cover a range of the draw's slots)
    __device__ bool load(unsigned i, float o[3], float d[3], bool &any_hit) const {
        unsigned px, py, k;
        if(!state_pixel(dp, base + i, px, py, k)) return false;  // a slot of an edge tile that names no pixel: nobody reads it
        V3 ro, rd;
        camera_ray(dp.p, dp.samples[k].jitter[0], dp.samples[k].jitter[1], dp.tan_half_fov, px, py, ro, rd);
        // (nothing of the path state is stored: the launch that shades vertex 0 makes the same ray again and knows
        // what throughput, sampler dimension and accumulators start with; a ray the reference-order traversal has to
        // trace again is stored by the sink)
        o[0] = ro.x; o[1] = ro.y; o[2] = ro.z;
        d[0] = rd.x; d[1] = rd.y; d[2] = rd.z;
        any_hit = false;
        return true;
    }
};

struct PrimarySinkFast {
    const DrawParams &dp;
    unsigned *redo, *redo_count;
    unsigned base;  // as in PrimarySource
    __device__ void store(unsigned item, const LaneHit &h, bool ambiguous) const {
        const unsigned i = base + item;
        write_path_hit(dp.hits, i, h);  // (provisional for an ambiguous ray: bounds its second pass)
        if(ambiguous) {
            // the reference-order launch behind this one reads its rays from the path state: store this one
            unsigned px, py, k;
            state_pixel(dp, i, px, py, k);
            V3 ro, rd;
            camera_ray(dp.p, dp.samples[k].jitter[0], dp.samples[k].jitter[1], dp.tan_half_fov, px, py, ro, rd);
            float4 *r = reinterpret_cast<float4 *>(dp.rays + i);
            r[0] = make_float4(ro.x, ro.y, ro.z, 0.0f);
            r[1] = make_float4(rd.x, rd.y, rd.z, 0.0f);
            redo[atomicAdd(redo_count, 1u)] = i;
        }
    }
    // (the form the walk of whole tiles calls, trace_packet.h: it names slots)
    __device__ void tile(unsigned slot, bool pixel, const LaneHit &h, bool ambiguous) const {
        if(pixel) store(slot - base, h, ambiguous);
    }
    __device__ void finish() const {}
};

// The slots [base, base + count) of the draw (all of them, or -- when the host times the two kinds of launch against
// each other on one draw -- the samples of one half); books: this launch keeps the draw's books.
template <int MODE>
__global__ __launch_bounds__(FAST_BLOCK) RT_FAST_OCC void primary_fast_kernel(FastScene sc, DrawParams dp, unsigned *next_counter,
                                                                  unsigned *redo, unsigned *redo_count, unsigned base, unsigned count,
                                                                  bool books) {
    __shared__ uint2 s_stack[FAST_LDS_STACK * FAST_BLOCK];
    if(books && blockIdx.x == 0 && threadIdx.x == 0) {
        // the books generate_kernel keeps (path_kernels.hip)
        atomicAdd(&dp.stats[0], (unsigned long long)dp.W * (unsigned long long)dp.H * (unsigned long long)dp.batch);
        atomicAdd(&dp.stats[3], (unsigned long long)dp.batch);
        dp.counters[0] = (unsigned)dp.npix;  // every slot is in the first queue
    }
    PrimarySource src{dp, base};
    PrimarySinkFast sink{dp, redo, redo_count, base};
    trace_fast_any<MODE>(sc, src, sink, next_counter, count, s_stack);
}

// The same launch for a pass that walks the flattened tree: one wave per tile of 64 slots, all lanes on one node
// sequence (trace_packet.h).  No LDS.
__global__ __launch_bounds__(FAST_BLOCK) RT_FAST_OCC void primary_packet_kernel(FastScene sc, DrawParams dp, unsigned *next_counter,
                                                                    unsigned *redo, unsigned *redo_count, unsigned base, unsigned count,
                                                                    bool books) {
    if(books && blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&dp.stats[0], (unsigned long long)dp.W * (unsigned long long)dp.H * (unsigned long long)dp.batch);
        atomicAdd(&dp.stats[3], (unsigned long long)dp.batch);
        dp.counters[0] = (unsigned)dp.npix;  // every slot is in the first queue
    }
    PrimarySinkFast sink{dp, redo, redo_count, base};
    trace_flat_packets(sc, CameraRays{dp, base}, sink, next_counter, count);
}

// The packet launch and the shade launch of vertex 0 as ONE kernel.  The walk of a tile is bound by vector-ALU issue,
// the shading of its 64 hits by memory (gathers of triangle, material, texel and sky records, the path state it
// writes): as two launches each leaves the other resource idle; in one kernel the waves of a SIMD are at different
// points -- some walking, some waiting for their gathers -- and the hardware overlaps them.  The hit record of vertex 0
// stays in registers (32 bytes per pixel less each way).  A wave walks a tile, shades its pixels where the walk ends
// (shade_vertex with the hit handed over), keeps what the pixels append to the two queues in LDS, and updates the queue
// counters once per PRIMARY_SHADE_ROUND tiles (path_kernels.hip has the reason: the chip completes about ninety
// same-word atomics per microsecond).  A ray the walk cannot decide is stored and listed as the packet launch does;
// the reference-order launch behind this one traces it and a small shade launch over that list shades it.
#ifndef RT_PRIMARY_SHADE_ROUND
#define RT_PRIMARY_SHADE_ROUND 8
#endif
constexpr int PRIMARY_SHADE_ROUND = RT_PRIMARY_SHADE_ROUND;

// Where the rays that LEAVE a tile are as alike as the camera rays that reached it -- a tile on a polished floor or a
// pane: the demo scene's loader turns roughnessFactor 0 into the 0.001 lobe, whose samples stay within a tenth of a
// degree of the mirror direction -- they are worth walking as a group again.  The wave that shaded the tile knows: when
// at least PACKET_MIN_LANES of its paths go on and their new directions lie within PACKET_SPREAD of each other in every
// component, the tile's paths are listed as a group of 64 entries (PACKET_NONE for the lanes that do not go on) in a
// queue of their own instead of the queue of vertex 1; extend_packet_kernel walks such groups, the merged launch the rest.
#ifndef RT_PACKET_SPREAD
#define RT_PACKET_SPREAD 0.06f
#endif
#ifndef RT_PACKET_MIN_LANES
#define RT_PACKET_MIN_LANES 32
#endif
#ifndef RT_SHADOW_CONE_COS
#define RT_SHADOW_CONE_COS 0.995f
#endif
constexpr unsigned PACKET_NONE = 0xFFFFFFFFu;

// floats as unsigned integers in the same order (NaNs beyond the infinities)
__device__ __forceinline__ unsigned float_order(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_of_order(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

struct PrimaryShadeSink {
    const DrawParams &dp;
    unsigned *redo, *redo_count;
    const unsigned *s_T;   // the Sobol table in LDS
    unsigned *stage;       // this wave's PRIMARY_SHADE_ROUND x 64 words of LDS
    unsigned *queue_out, *count_out, *shadow_queue, *shadow_count;
    unsigned *packet_queue, *packet_count, *packet_valid;  // groups of coherent bounce rays (null: none are made)
    unsigned *sgroup_queue, *sgroup_count, *sgroup_valid;  // groups of coherent occlusion queries (null: none are made)
    unsigned staged, cont_bits, shad_bits, null_count;
    unsigned group_tiles;  // bit u: staged tile u goes to the packet queue (the same in every lane)
    unsigned group_bits;   // bit u: this lane's path of staged tile u is one of that group
    unsigned sgroup_tiles, sgroup_bits;  // the same for the occlusion queries of the tile's light samples

    __device__ void tile(unsigned i, bool pixel, const LaneHit &h, bool ambiguous) {
        const unsigned lane = threadIdx.x & 63u;
        bool continues = false, has_shadow = false;
        V3 nd = v3(0.0f), sd = v3(0.0f);
        if(pixel) {
            if(ambiguous) {
                PrimarySinkFast{dp, redo, redo_count, 0u}.store(i, h, true);
            } else {
                unsigned null_shadow;
                shade_vertex<0, true>(dp, 0, i, s_T, continues, has_shadow, null_shadow, &h, &nd, &sd);
                null_count += null_shadow;
            }
        }
        // The light samples of a tile: an importance-sampled sky sends most of them to its few brightest texels (the sun),
        // i.e. from neighbouring points in nearly one direction.  Those that lie within a cone about one of them -- the
        // first lane's sample, or, when that one is a stray, the first one outside its cone -- are listed as a group.
        if(sgroup_queue && __popcll(__ballot(has_shadow)) >= RT_PACKET_MIN_LANES) {
            unsigned long long todo = __ballot(has_shadow);
            bool in = false;
#pragma unroll 1
            for(int attempt = 0; attempt < 2 && todo; attempt++) {
                const int ref = __builtin_ctzll(todo);
                const float ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sd.x), ref));
                const float ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sd.y), ref));
                const float az = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sd.z), ref));
                in = has_shadow && (sd.x * ax + sd.y * ay) + sd.z * az > RT_SHADOW_CONE_COS;
                const unsigned long long m = __ballot(in);
                if(__popcll(m) >= RT_PACKET_MIN_LANES) break;
                in = false;
                todo &= ~m;
                todo &= ~(1ull << ref);
            }
            if(__any(in)) {
                sgroup_tiles |= 1u << staged;
                sgroup_bits |= (in ? 1u : 0u) << staged;
                if(in) has_shadow = false;  // not in the queue of occlusion queries
            }
        }
        if(packet_queue && __popcll(__ballot(continues)) >= RT_PACKET_MIN_LANES) {
            const float dx = float_of_order(wave_umax(continues ? float_order(nd.x) : 0u)) - float_of_order(wave_umin(continues ? float_order(nd.x) : 0xFFFFFFFFu));
            const float dy = float_of_order(wave_umax(continues ? float_order(nd.y) : 0u)) - float_of_order(wave_umin(continues ? float_order(nd.y) : 0xFFFFFFFFu));
            const float dz = float_of_order(wave_umax(continues ? float_order(nd.z) : 0u)) - float_of_order(wave_umin(continues ? float_order(nd.z) : 0xFFFFFFFFu));
            if(fmaxf(dx, fmaxf(dy, dz)) < RT_PACKET_SPREAD) {  // (a NaN anywhere: no)
                group_tiles |= 1u << staged;
                group_bits |= (continues ? 1u : 0u) << staged;
                continues = false;  // not in the queue of vertex 1
            }
        }
        stage[staged * 64u + lane] = i;
        cont_bits |= (continues ? 1u : 0u) << staged;
        shad_bits |= (has_shadow ? 1u : 0u) << staged;
        if(++staged == (unsigned)PRIMARY_SHADE_ROUND) flush();
    }
    __device__ void flush() {
        const unsigned lane = threadIdx.x & 63u;
        if(group_tiles) {
            const unsigned n_groups = (unsigned)__popc(group_tiles);
            unsigned valid = 0u;
#pragma unroll 1
            for(unsigned u = 0; u < staged; u++) valid += (unsigned)__popcll(__ballot((group_bits >> u) & 1u));
            unsigned g = 0u;
            if(lane == 0u) {
                g = atomicAdd(packet_count, n_groups);
                atomicAdd(packet_valid, valid);
            }
            g = __builtin_amdgcn_readfirstlane(g);
#pragma unroll 1
            for(unsigned u = 0; u < staged; u++)
                if((group_tiles >> u) & 1u) {
                    packet_queue[(size_t)g * 64u + lane] = ((group_bits >> u) & 1u) ? stage[u * 64u + lane] : PACKET_NONE;
                    ++g;
                }
            group_tiles = group_bits = 0u;
        }
        if(sgroup_tiles) {
            const unsigned n_groups = (unsigned)__popc(sgroup_tiles);
            unsigned valid = 0u;
#pragma unroll 1
            for(unsigned u = 0; u < staged; u++) valid += (unsigned)__popcll(__ballot((sgroup_bits >> u) & 1u));
            unsigned g = 0u;
            if(lane == 0u) {
                g = atomicAdd(sgroup_count, n_groups);
                atomicAdd(sgroup_valid, valid);
            }
            g = __builtin_amdgcn_readfirstlane(g);
#pragma unroll 1
            for(unsigned u = 0; u < staged; u++)
                if((sgroup_tiles >> u) & 1u) {
                    sgroup_queue[(size_t)g * 64u + lane] = ((sgroup_bits >> u) & 1u) ? stage[u * 64u + lane] : PACKET_NONE;
                    ++g;
                }
            sgroup_tiles = sgroup_bits = 0u;
        }
        unsigned tot_c = 0u, tot_s = 0u;
#pragma unroll 1
        for(unsigned u = 0; u < staged; u++) {
            tot_c += (unsigned)__popcll(__ballot((cont_bits >> u) & 1u));
            tot_s += (unsigned)__popcll(__ballot((shad_bits >> u) & 1u));
        }
        unsigned base_c = 0u, base_s = 0u;
        if(lane == 0u) {
            if(tot_c) base_c = atomicAdd(count_out, tot_c);
            if(tot_s) base_s = atomicAdd(shadow_count, tot_s);
        }
        base_c = __builtin_amdgcn_readfirstlane(base_c);
        base_s = __builtin_amdgcn_readfirstlane(base_s);
#pragma unroll 1
        for(unsigned u = 0; u < staged; u++) {
            const bool c = (cont_bits >> u) & 1u, sh = (shad_bits >> u) & 1u;
            const unsigned long long m_c = __ballot(c), m_s = __ballot(sh);
            const unsigned slot = (c || sh) ? stage[u * 64u + lane] : 0u;
            if(c) queue_out[base_c + __builtin_amdgcn_mbcnt_hi((unsigned)(m_c >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_c, 0u))] = slot;
            if(sh) shadow_queue[base_s + __builtin_amdgcn_mbcnt_hi((unsigned)(m_s >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_s, 0u))] = slot;
            base_c += (unsigned)__popcll(m_c);
            base_s += (unsigned)__popcll(m_s);
        }
        staged = 0u;
        cont_bits = shad_bits = 0u;
    }
    __device__ void finish() {
        if(staged) flush();
    }
};

__global__ __launch_bounds__(FAST_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void primary_shade_packet_kernel(
    FastScene sc, DrawParams dp, unsigned *next_counter, unsigned *redo, unsigned *redo_count, unsigned *queue_out, unsigned *count_out,
    unsigned *shadow_queue, unsigned *shadow_count, unsigned *packet_queue, unsigned *packet_count, unsigned *packet_valid,
    unsigned *sgroup_queue, unsigned *sgroup_count, unsigned *sgroup_valid) {
    __shared__ __align__(16) unsigned s_V[SOBOL_NIBBLE_WORDS];
    __shared__ unsigned s_stage[PRIMARY_SHADE_ROUND * FAST_BLOCK];
    __shared__ unsigned s_null;
    if(threadIdx.x == 0) s_null = 0u;
    load_sobol_to_lds(s_V);
    if(blockIdx.x == 0 && threadIdx.x == 0) {
        // the books of the two launches this one stands for (generate / primary, and shade of vertex 0)
        const unsigned long long rays = (unsigned long long)dp.W * (unsigned long long)dp.H * (unsigned long long)dp.batch;
        atomicAdd(&dp.stats[0], rays);
        atomicAdd(&dp.stats[3], (unsigned long long)dp.batch);
        atomicAdd(&dp.stats[1], rays);
        dp.counters[0] = (unsigned)dp.npix;  // every slot is in the first queue
    }
    PrimaryShadeSink sink{dp, redo, redo_count, s_V, s_stage + (threadIdx.x >> 6) * (PRIMARY_SHADE_ROUND * 64), queue_out, count_out,
                          shadow_queue, shadow_count, packet_queue, packet_count, packet_valid, sgroup_queue, sgroup_count, sgroup_valid,
                          0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    trace_flat_packets(sc, CameraRays{dp, 0u}, sink, next_counter, (unsigned)dp.npix);
    if(sink.null_count) atomicAdd(&s_null, sink.null_count);
    __syncthreads();
    if(threadIdx.x == 0 && s_null) atomicAdd(&dp.stats[5], (unsigned long long)s_null);
}

// The extend rays of the groups the launch above listed: one wave per group on one node sequence.
struct QueuedRays {
    const unsigned *queue;
    const PathRay *rays;
    __device__ bool make(unsigned item, unsigned &slot, V3 &o, V3 &d) const {
        const unsigned e = queue[item];
        if(e == PACKET_NONE) return false;
        slot = e;
        float oo[3], dd[3];
        load_path_ray(rays, e, oo, dd);
        o = v3(oo[0], oo[1], oo[2]);
        d = v3(dd[0], dd[1], dd[2]);
        return true;
    }
};
struct GroupExtendSink {
    PathHit *hits;
    unsigned *redo, *redo_count;
    __device__ void tile(unsigned slot, bool has_ray, const LaneHit &h, bool ambiguous) const {
        if(!has_ray) return;
        write_path_hit(hits, slot, h);  // (provisional for an ambiguous ray: bounds its second pass)
        if(ambiguous) redo[atomicAdd(redo_count, 1u)] = slot;
    }
    __device__ void finish() const {}
};
__global__ __launch_bounds__(FAST_BLOCK) RT_FAST_OCC void extend_packet_kernel(FastScene sc, const unsigned *queue, const unsigned *group_count,
                                                                   const PathRay *rays, PathHit *hits, unsigned *next_counter,
                                                                   unsigned *redo, unsigned *redo_count) {
    GroupExtendSink sink{hits, redo, redo_count};
    trace_flat_packets(sc, QueuedRays{queue, rays}, sink, next_counter, *group_count * 64u);
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
    __device__ float bound(unsigned) const { return RT_INF; }  // occlusion queries have no second pass
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

// The same work items through the ordered traversal (a vertex with analytic lights): the sky's sample is an occlusion
// query, a light's sample asks for the closest hit, whose distance is compared with the light's (ShadowSink).  What
// the ordered traversal leaves undecided is listed by work item and traced in the reference's order behind it.
struct ShadowSinkFast {
    ShadowSink base;
    unsigned *redo, *redo_count;
    __device__ void store(unsigned idx, const LaneHit &h, bool ambiguous) const {
        if(ambiguous) redo[atomicAdd(redo_count, 1u)] = idx;
        else base.store(idx, h);
    }
};
template <int MODE>
__global__ __launch_bounds__(FAST_BLOCK) RT_FAST_OCC void shadow_fast_kernel(FastScene sc, const unsigned *queue, const unsigned *count,
                                                                 const PathRay *rays, ShadowEntry *shadow, int n_shadow, int npix,
                                                                 unsigned *next_counter, unsigned *redo, unsigned *redo_count) {
    __shared__ uint2 s_stack[FAST_LDS_STACK * FAST_BLOCK];
    ShadowSource src{queue, rays, shadow, n_shadow, npix};
    ShadowSinkFast sink{ShadowSink{queue, shadow, n_shadow, npix}, redo, redo_count};
    trace_fast_any<MODE>(sc, src, sink, next_counter, *count * (unsigned)n_shadow, s_stack);
}
struct ShadowRedoSource {
    ShadowSource s;
    const unsigned *redo;
    __device__ float bound(unsigned) const { return RT_INF; }
    __device__ bool load(unsigned i, float o[3], float d[3], bool &any_hit) const { return s.load(redo[i], o, d, any_hit); }
};
struct ShadowRedoSink {
    ShadowSink s;
    const unsigned *redo;
    __device__ void store(unsigned i, const LaneHit &h) const { s.store(redo[i], h); }
};
__global__ __launch_bounds__(TRACE_BLOCK) void shadow_redo_kernel(TraceScene sc, const unsigned *queue, const PathRay *rays,
                                                                  ShadowEntry *shadow, int n_shadow, int npix, const unsigned *redo,
                                                                  const unsigned *redo_count, unsigned *next_counter) {
    __shared__ uint2 s_stack[LDS_STACK * TRACE_BLOCK];
    ShadowRedoSource src{ShadowSource{queue, rays, shadow, n_shadow, npix}, redo};
    ShadowRedoSink sink{ShadowSink{queue, shadow, n_shadow, npix}, redo};
    trace_persistent<SparseRayFeed>(sc, src, sink, next_counter, *redo_count, s_stack);
}

// ---- shadow rays of vertex d and extend rays of vertex d+1 in ONE launch ---------------------
// Both are produced by shade(d) and neither reads what the other writes (shadow rays start at
// the origin of the new ray, extend writes hits, a visible light sample goes straight into its
// path's accumulator), so they share
// one persistent launch: work item idx < n_extend is an extend ray, the rest are shadow rays.
// Halves the number of trace launches per draw; late vertices carry few rays and each launch
// costs at least the latency of its longest ray.
#ifndef RT_PATH_REFILL
#define RT_PATH_REFILL 28
#endif
struct PathSource {
    static constexpr int REFILL_BELOW = RT_PATH_REFILL;  // read by the flattened tree's walk (trace_flat.h)
    ExtendSource e;
    ShadowSource s;
    unsigned n_extend;
    __device__ bool load(unsigned idx, float o[3], float d[3], bool &any_hit) const {
        return idx < n_extend ? e.load(idx, o, d, any_hit) : s.load(idx - n_extend, o, d, any_hit);
    }
};
// The sky-only form of shadow_apply_kernel (path_kernels.hip), done where the occlusion query
// ends: one pending sample per path, so nobody else touches that path's accumulator meanwhile.
struct ShadowApplySink {
    const unsigned *queue;
    const ShadowEntry *shadow;
    const PathThroughput *thr;
    PathAccum *acc;
    __device__ void store(unsigned idx, const LaneHit &h) const {
        apply_sky_sample(shadow, thr, acc, queue[idx], h);
    }
};
// An occlusion query the ordered kernel gives up (a zero or denormal direction component, trace_fast.h) goes
// on the same list as the undecided extend rays, tagged: the reference-order launch behind traces both kinds.
constexpr unsigned REDO_SHADOW = 0x80000000u;
struct PathSink {
    ExtendSinkFast e;
    ShadowApplySink s;
    unsigned n_extend;
    __device__ void store(unsigned idx, const LaneHit &h, bool ambiguous) const {
        if(idx < n_extend) e.store(idx, h, ambiguous);
        else if(ambiguous) e.redo[atomicAdd(e.redo_count, 1u)] = s.queue[idx - n_extend] | REDO_SHADOW;
        else s.store(idx - n_extend, h);
    }
};
// the rays on that list
struct RedoSource {
    const unsigned *queue;
    const PathRay *rays;
    const PathHit *hits;  // what the ordered pass found for the extend rays: bounds their second pass
    const ShadowEntry *shadow;
    __device__ float bound(unsigned idx) const {
        const unsigned w = queue[idx];
        return (w & REDO_SHADOW) ? RT_INF : hits[w].t;
    }
    __device__ bool load(unsigned idx, float o[3], float d[3], bool &any_hit) const {
        const unsigned w = queue[idx], px = w & ~REDO_SHADOW;
        if(!(w & REDO_SHADOW)) {
            load_path_ray(rays, px, o, d);
            any_hit = false;
            return true;
        }
        const float4 *e = reinterpret_cast<const float4 *>(shadow + px);  // (sample 0: the sky's)
        const float4 a = e[0], b = e[1];
        if(!(__float_as_uint(b.w) & 1u)) return false;
        const float4 ro = *reinterpret_cast<const float4 *>(rays + px);
        o[0] = ro.x; o[1] = ro.y; o[2] = ro.z;
        d[0] = a.x; d[1] = a.y; d[2] = a.z;
        any_hit = true;
        return true;
    }
};
struct RedoSink {
    const unsigned *queue;
    PathHit *hits;
    const ShadowEntry *shadow;
    const PathThroughput *thr;
    PathAccum *acc;
    __device__ void store(unsigned idx, const LaneHit &h) const {
        const unsigned w = queue[idx];
        if(w & REDO_SHADOW) apply_sky_sample(shadow, thr, acc, w & ~REDO_SHADOW, h);
        else write_path_hit(hits, w, h);
    }
};

__global__ __launch_bounds__(TRACE_BLOCK) void path_redo_kernel(TraceScene sc, const unsigned *queue, const unsigned *count,
                                                                const PathRay *rays, PathHit *hits, const ShadowEntry *shadow,
                                                                const PathThroughput *thr, PathAccum *acc, unsigned *next_counter) {
    __shared__ uint2 s_stack[LDS_STACK * TRACE_BLOCK];
    RedoSource src{queue, rays, hits, shadow};
    RedoSink sink{queue, hits, shadow, thr, acc};
    trace_persistent<SparseRayFeed>(sc, src, sink, next_counter, *count, s_stack);
}

// The occlusion queries of the groups the launch of vertex 0 listed (sky samples of a tile that point one way): one wave
// per group on one node sequence, every lane out at its first hit; a visible sample goes into its path's accumulator
// where the walk ends, as in the merged launch, and a query given up by rule (5) onto the same list, tagged.
struct QueuedShadowRays {
    const unsigned *queue;
    const PathRay *rays;
    const ShadowEntry *shadow;
    __device__ bool make(unsigned item, unsigned &slot, V3 &o, V3 &d) const {
        const unsigned e = queue[item];
        if(e == PACKET_NONE) return false;
        slot = e;
        const float4 a = reinterpret_cast<const float4 *>(shadow + e)[0];
        const float4 ro = *reinterpret_cast<const float4 *>(rays + e);
        o = v3(ro.x, ro.y, ro.z);
        d = v3(a.x, a.y, a.z);
        return true;
    }
};
struct GroupShadowSink {
    const ShadowEntry *shadow;
    const PathThroughput *thr;
    PathAccum *acc;
    unsigned *redo, *redo_count;
    __device__ void tile(unsigned slot, bool has_ray, const LaneHit &h, bool ambiguous) const {
        if(!has_ray) return;
        if(ambiguous) redo[atomicAdd(redo_count, 1u)] = slot | REDO_SHADOW;
        else apply_sky_sample(shadow, thr, acc, slot, h);
    }
    __device__ void finish() const {}
};
__global__ __launch_bounds__(FAST_BLOCK) RT_FAST_OCC void shadow_packet_kernel(FastScene sc, const unsigned *queue, const unsigned *group_count,
                                                                   const unsigned *valid_count, const PathRay *rays,
                                                                   const ShadowEntry *shadow, const PathThroughput *thr, PathAccum *acc,
                                                                   unsigned *next_counter, unsigned *redo, unsigned *redo_count,
                                                                   unsigned long long *stats) {
    if(blockIdx.x == 0 && threadIdx.x == 0 && *valid_count) {
        atomicAdd(stats + 2, (unsigned long long)*valid_count);  // occlusion queries traced
        atomicAdd(stats + 7, (unsigned long long)*valid_count);  // ... as groups
    }
    GroupShadowSink sink{shadow, thr, acc, redo, redo_count};
    trace_flat_packets<true>(sc, QueuedShadowRays{queue, rays, shadow}, sink, next_counter, *group_count * 64u);
}

template <int MODE>
__global__ __launch_bounds__(FAST_BLOCK) RT_FAST_OCC void path_trace_fast_kernel(
    FastScene sc, const unsigned *queue_e, const unsigned *count_e, const unsigned *queue_s, const unsigned *count_s,
    const PathRay *rays, PathHit *hits, ShadowEntry *shadow, int npix, unsigned *next_counter, unsigned *redo,
    unsigned *redo_count, const PathThroughput *thr, PathAccum *acc, unsigned long long *shadow_rays_stat) {
    __shared__ uint2 s_stack[FAST_LDS_STACK * FAST_BLOCK];
    const unsigned n_e = *count_e, n_s = *count_s;
    if(blockIdx.x == 0 && threadIdx.x == 0 && n_s) atomicAdd(shadow_rays_stat, (unsigned long long)n_s);
    PathSource src{ExtendSource{queue_e, rays, nullptr}, ShadowSource{queue_s, rays, shadow, 1, npix}, n_e};
    PathSink sink{ExtendSinkFast{queue_e, hits, redo, redo_count}, ShadowApplySink{queue_s, shadow, thr, acc}, n_e};
    trace_fast_any<MODE>(sc, src, sink, next_counter, n_e + n_s, s_stack);
}

}  // namespace

void launch_path_trace_fast(hipStream_t s, const FastScene &sc, int blocks, const unsigned *queue_e, const unsigned *count_e,
                            const unsigned *queue_s, const unsigned *count_s, const PathRay *rays, PathHit *hits,
                            ShadowEntry *shadow, int npix, unsigned *next_counter, unsigned *redo, unsigned *redo_count,
                            const PathThroughput *thr, PathAccum *acc, unsigned long long *shadow_rays_stat) {
    hipLaunchKernelGGL(RT_PICK_FAST(path_trace_fast_kernel), dim3(blocks), dim3(FAST_BLOCK), 0, s, sc, queue_e, count_e, queue_s, count_s, rays,
                       hits, shadow, npix, next_counter, redo, redo_count, thr, acc, shadow_rays_stat);
}

void launch_primary_fast(hipStream_t s, const FastScene &sc, int blocks, const DrawParams &dp, unsigned *next_counter, unsigned *redo,
                         unsigned *redo_count, bool packets, unsigned base, unsigned count, bool books) {
    if(packets && sc.flat_buf)
        hipLaunchKernelGGL(primary_packet_kernel, dim3(blocks), dim3(FAST_BLOCK), 0, s, sc, dp, next_counter, redo, redo_count, base,
                           count, books);
    else
        hipLaunchKernelGGL(RT_PICK_FAST(primary_fast_kernel), dim3(blocks), dim3(FAST_BLOCK), 0, s, sc, dp,
                           next_counter, redo, redo_count, base, count, books);
}
void launch_primary_shade_packets(hipStream_t s, const FastScene &sc, int blocks, const DrawParams &dp, unsigned *next_counter,
                                  unsigned *redo, unsigned *redo_count, unsigned *queue_out, unsigned *count_out,
                                  unsigned *shadow_queue, unsigned *shadow_count, unsigned *packet_queue, unsigned *packet_count,
                                  unsigned *packet_valid, unsigned *sgroup_queue, unsigned *sgroup_count, unsigned *sgroup_valid) {
    hipLaunchKernelGGL(primary_shade_packet_kernel, dim3(blocks), dim3(FAST_BLOCK), 0, s, sc, dp, next_counter, redo, redo_count,
                       queue_out, count_out, shadow_queue, shadow_count, packet_queue, packet_count, packet_valid, sgroup_queue, sgroup_count,
                       sgroup_valid);
}
void launch_shadow_packets(hipStream_t s, const FastScene &sc, int blocks, const unsigned *queue, const unsigned *group_count,
                           const unsigned *valid_count, const PathRay *rays, const ShadowEntry *shadow, const PathThroughput *thr,
                           PathAccum *acc, unsigned *next_counter, unsigned *redo, unsigned *redo_count, unsigned long long *stats) {
    hipLaunchKernelGGL(shadow_packet_kernel, dim3(blocks), dim3(FAST_BLOCK), 0, s, sc, queue, group_count, valid_count, rays, shadow, thr, acc,
                       next_counter, redo, redo_count, stats);
}
void launch_extend_packets(hipStream_t s, const FastScene &sc, int blocks, const unsigned *queue, const unsigned *group_count,
                           const PathRay *rays, PathHit *hits, unsigned *next_counter, unsigned *redo, unsigned *redo_count) {
    hipLaunchKernelGGL(extend_packet_kernel, dim3(blocks), dim3(FAST_BLOCK), 0, s, sc, queue, group_count, rays, hits, next_counter, redo,
                       redo_count);
}

int packet_stack_entries() { return PACKET_STACK; }

void launch_intersect(hipStream_t s, const TraceScene &sc, int blocks, const float *rays, unsigned n,
                      const unsigned *redo, const unsigned *redo_count, pt_hit *hits, unsigned *next_counter,
                      unsigned long long *retraced_stat) {
    hipLaunchKernelGGL(intersect_kernel, dim3(blocks), dim3(TRACE_BLOCK), 0, s, sc, rays, n, redo, redo_count, hits,
                       next_counter, retraced_stat);
}

void launch_intersect_fast(hipStream_t s, const FastScene &sc, int blocks, const float *rays, unsigned n,
                           pt_hit *hits, unsigned *next_counter, unsigned *redo, unsigned *redo_count) {
    hipLaunchKernelGGL(RT_PICK_FAST(intersect_fast_kernel), dim3(blocks), dim3(FAST_BLOCK), 0, s, sc, rays, n, hits, next_counter,
                       redo, redo_count);
}

void launch_extend(hipStream_t s, const TraceScene &sc, int blocks, const unsigned *queue, const unsigned *count,
                   const PathRay *rays, PathHit *hits, unsigned *next_counter, bool second_pass) {
    hipLaunchKernelGGL(extend_kernel, dim3(blocks), dim3(TRACE_BLOCK), 0, s, sc, queue, count, rays, hits,
                       next_counter, second_pass);
}

void launch_extend_fast(hipStream_t s, const FastScene &sc, int blocks, const unsigned *queue, const unsigned *count,
                        const PathRay *rays, PathHit *hits, unsigned *next_counter, unsigned *redo,
                        unsigned *redo_count) {
    hipLaunchKernelGGL(RT_PICK_FAST(extend_fast_kernel), dim3(blocks), dim3(FAST_BLOCK), 0, s, sc, queue, count, rays, hits,
                       next_counter, redo, redo_count);
}

void launch_shadow(hipStream_t s, const TraceScene &sc, int blocks, const unsigned *queue, const unsigned *count,
                   const PathRay *rays, ShadowEntry *shadow, int n_shadow, int npix, unsigned *next_counter) {
    hipLaunchKernelGGL(shadow_kernel, dim3(blocks), dim3(TRACE_BLOCK), 0, s, sc, queue, count, rays, shadow,
                       n_shadow, npix, next_counter);
}

void launch_shadow_fast(hipStream_t s, const FastScene &sc, int blocks, const unsigned *queue, const unsigned *count,
                        const PathRay *rays, ShadowEntry *shadow, int n_shadow, int npix, unsigned *next_counter, unsigned *redo,
                        unsigned *redo_count) {
    hipLaunchKernelGGL(RT_PICK_FAST(shadow_fast_kernel), dim3(blocks), dim3(FAST_BLOCK), 0, s, sc, queue, count, rays, shadow,
                       n_shadow, npix, next_counter, redo, redo_count);
}

void launch_shadow_redo(hipStream_t s, const TraceScene &sc, int blocks, const unsigned *queue, const PathRay *rays,
                        ShadowEntry *shadow, int n_shadow, int npix, const unsigned *redo, const unsigned *redo_count,
                        unsigned *next_counter) {
    hipLaunchKernelGGL(shadow_redo_kernel, dim3(blocks), dim3(TRACE_BLOCK), 0, s, sc, queue, rays, shadow, n_shadow, npix, redo,
                       redo_count, next_counter);
}

void launch_path_redo(hipStream_t s, const TraceScene &sc, int blocks, const unsigned *queue, const unsigned *count,
                      const PathRay *rays, PathHit *hits, const ShadowEntry *shadow, const PathThroughput *thr, PathAccum *acc,
                      unsigned *next_counter) {
    hipLaunchKernelGGL(path_redo_kernel, dim3(blocks), dim3(TRACE_BLOCK), 0, s, sc, queue, count, rays, hits, shadow, thr, acc,
                       next_counter);
}

}  // namespace rta
