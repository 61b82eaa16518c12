// The thin tail of a draw in ONE launch.
//
// A draw is a chain of launches, vertex by vertex: shade(d) -> trace(d) -> shade(d+1) ...
// (path_kernels.hip, trace_kernels.hip).  Late vertices carry few paths -- the demo scene is open,
// most paths have left it after two or three bounces -- but every launch of the chain costs at
// least the latency of its slowest ray (a few hundred dependent node visits, one after the
// other), so six thin vertices cost six times that, with the chip nearly empty.  The dependency
// is per path, though, not per vertex: nothing a path does at vertex d+1 waits for another path's
// vertex d.  Here a wave takes a handful of live paths and carries them through ALL their
// remaining vertices -- shade, trace its own shadow and extend rays, shade again ... -- so the
// draw waits for the slowest PATH instead of for the sum of the slowest rays of every vertex.
//
// Each step is the code the per-vertex launches run (shade_vertex, trace_ordered and, for rays
// the ordered traversal leaves undecided, trace_persistent), on the same path state in the same
// order per path, so the outputs are the same bits; only which lane runs what differs.
#include "kernels.h"
#include "path_state_io.h"
#include "shade_vertex.h"
#include "trace_fast.h"
#include "trace_flat.h"

namespace rta {
namespace {

#ifndef RT_TAIL_PATHS_PER_WAVE
#define RT_TAIL_PATHS_PER_WAVE 64
#endif
constexpr unsigned TAIL_PPW = RT_TAIL_PATHS_PER_WAVE;  // paths a wave carries at a time: 2 * PPW rays for its 64 lanes
static_assert(TAIL_PPW >= 1 && TAIL_PPW <= 64, "a wave carries at most one path per lane");
constexpr int TAIL_BLOCK = 256;
static_assert(TAIL_BLOCK == FAST_BLOCK && TAIL_BLOCK == TRACE_BLOCK, "the traversal stacks are laid out for 256 threads");
// (the kernel's one LDS stack, sized for the ordered traversal, also serves the reference-order one)
static_assert(LDS_STACK <= FAST_LDS_STACK, "trace_persistent indexes s_stack up to LDS_STACK entries per thread");

// one word per carried path, in LDS: slot number and what shade left for it
constexpr unsigned TP_SLOT = (1u << 28) - 1u;
constexpr unsigned TP_AMBIGUOUS_S = 1u << 28;  // its occlusion query was given up by the ordered traversal (trace_fast.h, rule (5))
constexpr unsigned TP_AMBIGUOUS = 1u << 29;  // its extend ray must be traced again in the reference's order
constexpr unsigned TP_CONTINUES = 1u << 30;  // an extend ray is due
constexpr unsigned TP_SHADOW = 1u << 31;     // a sky light sample waits for its occlusion ray

// rays of the wave: idx < PPW the extend ray of path idx, else the shadow ray of path idx - PPW
struct TailSource {
    const unsigned *wp;
    const PathRay *rays;
    const ShadowEntry *shadow;
    __device__ bool load(unsigned idx, float o[3], float d[3], bool &any_hit) const {
        const bool ext = idx < TAIL_PPW;
        const unsigned w = wp[ext ? idx : idx - TAIL_PPW], px = w & TP_SLOT;
        if(ext) {
            if(!(w & TP_CONTINUES)) return false;
            load_path_ray(rays, px, o, d);
            any_hit = false;
            return true;
        }
        if(!(w & TP_SHADOW)) return false;
        const float4 *e = reinterpret_cast<const float4 *>(shadow + px);
        const float4 a = e[0], b = e[1];
        if(!(__float_as_uint(b.w) & 1u)) return false;
        const float4 ro = *reinterpret_cast<const float4 *>(rays + px);
        o[0] = ro.x; o[1] = ro.y; o[2] = ro.z;
        d[0] = a.x; d[1] = a.y; d[2] = a.z;
        any_hit = true;
        return true;
    }
};
struct TailSink {
    unsigned *wp;
    PathHit *hits;
    const ShadowEntry *shadow;
    const PathThroughput *thr;
    PathAccum *acc;
    __device__ void store(unsigned idx, const LaneHit &h, bool ambiguous) const {
        if(idx < TAIL_PPW) {
            const unsigned w = wp[idx];
            write_path_hit(hits, w & TP_SLOT, h);  // (provisional for an ambiguous ray: bounds its second pass)
            if(ambiguous) atomicOr(&wp[idx], TP_AMBIGUOUS);  // (the path's occlusion query may flag the same word from another lane)
        } else if(ambiguous) {
            atomicOr(&wp[idx - TAIL_PPW], TP_AMBIGUOUS_S);
        } else {
            apply_sky_sample(shadow, thr, acc, wp[idx - TAIL_PPW] & TP_SLOT, h);
        }
    }
};
// second pass over the wave's undecided rays, in the reference's order: idx < PPW the extend ray of path idx,
// else the occlusion query of path idx - PPW
struct TailRedoSource {
    const unsigned *wp;
    const PathRay *rays;
    const PathHit *hits;
    const ShadowEntry *shadow;
    __device__ float bound(unsigned idx) const { return idx < TAIL_PPW ? hits[wp[idx] & TP_SLOT].t : RT_INF; }
    __device__ bool load(unsigned idx, float o[3], float d[3], bool &any_hit) const {
        const bool ext = idx < TAIL_PPW;
        const unsigned w = wp[ext ? idx : idx - TAIL_PPW], px = w & TP_SLOT;
        if(ext) {
            if(!(w & TP_AMBIGUOUS)) return false;
            load_path_ray(rays, px, o, d);
            any_hit = false;
            return true;
        }
        if(!(w & TP_AMBIGUOUS_S)) return false;
        const float4 *e = reinterpret_cast<const float4 *>(shadow + px);
        const float4 a = e[0];
        const float4 ro = *reinterpret_cast<const float4 *>(rays + px);
        o[0] = ro.x; o[1] = ro.y; o[2] = ro.z;
        d[0] = a.x; d[1] = a.y; d[2] = a.z;
        any_hit = true;
        return true;
    }
};
struct TailRedoSink {
    const unsigned *wp;
    PathHit *hits;
    const ShadowEntry *shadow;
    const PathThroughput *thr;
    PathAccum *acc;
    __device__ void store(unsigned idx, const LaneHit &h) const {
        if(idx < TAIL_PPW) write_path_hit(hits, wp[idx] & TP_SLOT, h);
        else apply_sky_sample(shadow, thr, acc, wp[idx - TAIL_PPW] & TP_SLOT, h);
    }
};

// what one lane stores must be seen by whichever lane of the wave reads it next
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

#ifdef RT_TAIL_WAVES
#define RT_TAIL_OCC __attribute__((amdgpu_waves_per_eu(RT_TAIL_WAVES, RT_TAIL_WAVES)))
#else
#define RT_TAIL_OCC
#endif
// MODE: 0 WNodes, 1 QNodes, 2 the flattened tree (as in trace_kernels.hip)
template <int MODE>
__global__ __launch_bounds__(TAIL_BLOCK) RT_TAIL_OCC void path_tail_kernel(FastScene fsc, TraceScene tsc, DrawParams dp, int dep0,
                                                               const unsigned *queue_in, const unsigned *count_in,
                                                               unsigned *next_batch) {
    __shared__ uint2 s_stack[FAST_LDS_STACK * FAST_BLOCK];
    __shared__ __align__(16) unsigned s_T[SOBOL_NIBBLE_WORDS];
    __shared__ unsigned s_paths[TAIL_BLOCK / 64][TAIL_PPW];
    const unsigned n = *count_in;
    const unsigned waves_per_block = TAIL_BLOCK / 64, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // every wave starts with the batch its position names; the counter hands out the ones behind
    if((size_t)blockIdx.x * waves_per_block * TAIL_PPW >= n && blockIdx.x != 0) return;
    load_sobol_to_lds(s_T);
    if(blockIdx.x == 0 && threadIdx.x == 0) {
        // the books shade_kernel keeps for the vertex this launch starts with
        atomicAdd(&dp.stats[1], (unsigned long long)n);
        if(dp.redo_counter_base >= 0) atomicAdd(&dp.stats[4], (unsigned long long)dp.counters[dp.redo_counter_base + dep0]);
    }
    unsigned *wp = s_paths[wave];
    unsigned long long n_extend = 0, n_shadow = 0, n_redo = 0, n_null = 0;  // wave-uniform
    unsigned batch = blockIdx.x * waves_per_block + wave;
    const unsigned first_dynamic = gridDim.x * waves_per_block;
    for(;;) {
        const unsigned base = batch * TAIL_PPW;
        if(base >= n) break;
        const unsigned cnt = n - base < TAIL_PPW ? n - base : TAIL_PPW;
        bool alive = lane < cnt;
        const unsigned slot = alive ? queue_in[base + lane] : 0u;
        for(int dep = dep0;; dep++) {
            bool continues = false, has_shadow = false;
            unsigned null_shadow = 0u;  // (0 or 1 here: this launch runs with the sky as the only light)
            if(alive) shade_vertex<0, false>(dp, dep, slot, s_T, continues, has_shadow, null_shadow);   // (dep0 >= 1)
            alive = continues;
            const unsigned long long m_c = __ballot(continues), m_s = __ballot(has_shadow);
            n_null += (unsigned)__popcll(__ballot(null_shadow != 0u));
            if((m_c | m_s) == 0ull) break;  // (at MAX_DEPTH shade_vertex lets no path continue)
            n_extend += (unsigned)__popcll(m_c);
            n_shadow += (unsigned)__popcll(m_s);
            if(lane < TAIL_PPW) wp[lane] = slot | (continues ? TP_CONTINUES : 0u) | (has_shadow ? TP_SHADOW : 0u);
            wave_sync();
            {
                TailSource src{wp, dp.rays, dp.shadow};
                TailSink sink{wp, dp.hits, dp.shadow, dp.thr, dp.acc};
                if constexpr(MODE == 2) trace_flat<LocalRayFeed>(fsc, src, sink, nullptr, 2u * TAIL_PPW, s_stack);
                else trace_ordered<MODE == 1, LocalRayFeed>(fsc, src, sink, nullptr, 2u * TAIL_PPW, s_stack);
            }
            wave_sync();
            const unsigned flags = lane < TAIL_PPW ? wp[lane] : 0u;
            const unsigned long long m_a = __ballot((flags & TP_AMBIGUOUS) != 0u), m_as = __ballot((flags & TP_AMBIGUOUS_S) != 0u);
            if((m_a | m_as) != 0ull) {
                n_redo += (unsigned)(__popcll(m_a) + __popcll(m_as));
                TailRedoSource src{wp, dp.rays, dp.hits, dp.shadow};
                TailRedoSink sink{wp, dp.hits, dp.shadow, dp.thr, dp.acc};
                trace_persistent<LocalRayFeed>(tsc, src, sink, nullptr, 2u * TAIL_PPW, s_stack);
                wave_sync();
            }
        }
        unsigned b = 0;
        if(lane == 0u) b = first_dynamic + atomicAdd(next_batch, 1u);
        batch = __builtin_amdgcn_readfirstlane(b);
    }
    if(lane == 0u) {
        if(n_extend) atomicAdd(&dp.stats[1], n_extend);
        if(n_shadow) atomicAdd(&dp.stats[2], n_shadow);
        if(n_redo) atomicAdd(&dp.stats[4], n_redo);
        if(n_null) atomicAdd(&dp.stats[5], n_null);
    }
}

}  // namespace

void launch_path_tail(hipStream_t s, const FastScene &fsc, const TraceScene &tsc, const DrawParams &dp, int dep0, int blocks,
                      const unsigned *queue_in, const unsigned *count_in, unsigned *next_batch) {
    hipLaunchKernelGGL(fsc.flat_buf ? path_tail_kernel<2> : (fsc.hdr.quantised ? path_tail_kernel<1> : path_tail_kernel<0>), dim3(blocks), dim3(TAIL_BLOCK), 0, s, fsc, tsc,
                       dp, dep0, queue_in, count_in, next_batch);
}

int tail_paths_per_block() { return (TAIL_BLOCK / 64) * (int)TAIL_PPW; }

}  // namespace rta
