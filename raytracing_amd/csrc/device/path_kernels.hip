// Wavefront stages of the path tracer other than ray traversal: generate, shade,
// shadow-apply and finalize.  Together with extend / shadow (trace_kernels.hip) they
// compute, per pixel, exactly what main() and shade_mis_advanced() of the reference's
// shader/pathtracing.glsl (:585-691, :489-582) compute in one fragment invocation.
//
// Mapping of the reference's bounce loop onto stages, for path vertex `dep`:
//   extend(dep)        the intersect_sceneBVH that produced this vertex (:627 or :568)
//   shade(dep)         :569-580 of the PREVIOUS iteration (escape to the sky, or throughput
//                      update), then :507-565 of this one: light samples are evaluated and
//                      parked as shadow entries, the BSDF sample becomes the next ray
//   shadow(dep)        the occlusion rays of :519 and :546
//   shadow_apply(dep)  :524-525 and :551-552, in the reference's order
// Paths that end simply are not appended to the next queue (queue compaction).
#include "shade_vertex.h"

namespace rta {
namespace {

// (128 threads per block measured 1.5 % (casa) and 0.6 % (10 M triangles) faster than 256 with ten draws in flight,
// 64 the same as 128, 512 12 % slower: small blocks find room next to the persistent traversal grids sooner)
#ifndef RT_SHADE_BLOCK
#define RT_SHADE_BLOCK 128
#endif
constexpr int SHADE_BLOCK = RT_SHADE_BLOCK;
#ifndef RT_SHADE_GROUP_BLOCK
#define RT_SHADE_GROUP_BLOCK 256
#endif
constexpr int SHADE_GROUP_BLOCK = RT_SHADE_GROUP_BLOCK;
#ifndef RT_SHADE_ROUND
#define RT_SHADE_ROUND 8
#endif
constexpr int SHADE_ROUND = RT_SHADE_ROUND;  // batches of 64 paths a wave shades between two updates of the queue counters

__global__ __launch_bounds__(SHADE_BLOCK) void generate_kernel(DrawParams dp) {
    const unsigned nthreads = gridDim.x * blockDim.x;
    for(unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)dp.npix; i += nthreads) {
        unsigned px, py, k;
        V3 o, d;
        if(state_pixel(dp, i, px, py, k)) {
            camera_ray(dp.p, dp.samples[k].jitter[0], dp.samples[k].jitter[1], dp.tan_half_fov, px, py, o, d);
        } else {
            // no pixel: a ray with a NaN direction ends as a miss without any traversal, and
            // shade() drops the slot
            o = v3(0.0f);
            d = v3(__uint_as_float(0x7FC00000u));
        }
        // (the rays only: the launch that shades vertex 0 knows what throughput, sampler dimension and accumulators
        // start with, shade_vertex.h)
        float4 *r = reinterpret_cast<float4 *>(dp.rays + i);
        r[0] = make_float4(o.x, o.y, o.z, 0.0f);
        r[1] = make_float4(d.x, d.y, d.z, 0.0f);
    }
    if(blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&dp.stats[0], (unsigned long long)dp.W * (unsigned long long)dp.H * (unsigned long long)dp.batch);
        atomicAdd(&dp.stats[3], (unsigned long long)dp.batch);
        dp.counters[0] = (unsigned)dp.npix;  // every slot is in the first queue
    }
}

// RASTER (see shade_vertex): 1 and 2 are the instances launched for vertex 0 when the raster pass's images are in
// play; kept apart so that the common instance carries none of their registers.
// Four waves per SIMD (128 registers, a few of them spilled) hide the latency of the gathers
// better than the three the kernel would get unconstrained (145 registers); five spill too much.
#ifndef SHADE_WAVES
#define SHADE_WAVES 4
#endif
// GROUP: every block first sorts its paths into two groups, so that a wave mostly holds one kind (below); such
// launches use blocks of SHADE_GROUP_BLOCK threads (four waves to sort among; measured on the demo scene, shade
// time per draw: 0.93 ms ungrouped, 0.85 / 0.82 / 0.88 ms with blocks of 128 / 256 / 512).
// FIRST: the launch of vertex 0 (dep == 0), see shade_vertex
template <int RASTER, bool GROUP, bool FIRST>
__global__ __launch_bounds__(GROUP ? SHADE_GROUP_BLOCK : SHADE_BLOCK) __attribute__((amdgpu_waves_per_eu(SHADE_WAVES, SHADE_WAVES))) void shade_kernel(DrawParams dp, int dep, const unsigned *queue_in,
                                                            const unsigned *count_in, unsigned *queue_out,
                                                            unsigned *count_out, unsigned *shadow_queue,
                                                            unsigned *shadow_count, bool group_materials, int what,
                                                            const unsigned *valid_count) {
    constexpr int BLOCK = GROUP ? SHADE_GROUP_BLOCK : SHADE_BLOCK;
    __shared__ __align__(16) unsigned s_V[SOBOL_NIBBLE_WORDS];
    __shared__ unsigned s_null;  // light samples of this block that needed no occlusion query
    // (SHADE_GROUPS: the queue holds groups of 64 entries, some of them 0xFFFFFFFF = no path)
    const unsigned n = what == SHADE_GROUPS ? *count_in * 64u : *count_in;
    const unsigned nthreads = gridDim.x * blockDim.x;
    {
        // (late bounces fill few blocks; block 0 books the counters)
        const unsigned per_thread0 = (n + nthreads - 1u) / nthreads;
        const unsigned U0 = per_thread0 < 1u ? 1u : (per_thread0 > (unsigned)SHADE_ROUND ? (unsigned)SHADE_ROUND : per_thread0);
        if(blockIdx.x * blockDim.x * U0 >= n && blockIdx.x != 0) return;
    }
    if(threadIdx.x == 0) s_null = 0u;
    unsigned my_null = 0u;
    load_sobol_to_lds(s_V);

    if(blockIdx.x == 0 && threadIdx.x == 0) {
        // (with RasterizaionFor1st the first vertex comes from images: no rays were traced for it)
        if(RASTER != 1 && what != SHADE_LISTED)
            atomicAdd(&dp.stats[1], FIRST ? (unsigned long long)dp.W * (unsigned long long)dp.H * (unsigned long long)dp.batch
                                          : (unsigned long long)(what == SHADE_GROUPS ? *valid_count : n));
        if(dp.redo_counter_base >= 0 && what != SHADE_GROUPS) atomicAdd(&dp.stats[4], (unsigned long long)dp.counters[dp.redo_counter_base + dep]);
        if(what == SHADE_GROUPS) {
            atomicAdd(&dp.counters[dep], *valid_count);  // (read from here on only as the vertex's count of live paths)
            atomicAdd(&dp.stats[6], (unsigned long long)*valid_count);
        }
    }

    __shared__ unsigned s_slot[GROUP ? BLOCK : 1];
    __shared__ unsigned s_count[GROUP ? BLOCK / 64 : 1];
    // Queue appends.  Every counter update is an atomic on ONE word, and the chip completes about ninety of those per
    // microsecond whoever issues them: with one update per wave and 64 paths a launch over 33 M paths (sixteen samples
    // of a 1080p frame) spent 5.9 ms waiting for its half a million updates and 3.3 ms without them.  So a wave shades
    // up to SHADE_ROUND batches of 64 paths, keeps what they append in LDS (one word per thread and batch, read back by
    // the thread that wrote it), and then makes ONE update per queue for all of them.  A launch with fewer paths than
    // threads runs one batch per round, as before: thin launches want every block busy.
    __shared__ unsigned s_stage[SHADE_ROUND * BLOCK];
    const unsigned per_thread = (n + nthreads - 1u) / nthreads;
    const unsigned U = per_thread < 1u ? 1u : (per_thread > (unsigned)SHADE_ROUND ? (unsigned)SHADE_ROUND : per_thread);
    const unsigned lane = threadIdx.x & 63u;
    for(unsigned r0 = blockIdx.x * blockDim.x * U; r0 < n; r0 += nthreads * U) {
        unsigned cont_bits = 0u, shad_bits = 0u;
#pragma unroll 1
        for(unsigned u = 0; u < U; u++) {
            const unsigned q0 = r0 + u * blockDim.x;
            if(q0 >= n) break;  // (the same for every thread of the block)
            const unsigned q = q0 + threadIdx.x;
            unsigned i = q < n ? (queue_in ? queue_in[q] : q) : 0u;
            bool mine = q < n;
            if(i == 0xFFFFFFFFu) {  // (an empty place of a group)
                mine = false;
                i = 0u;
            }
            if(GROUP) {
                // One group goes to the back of the block's paths, the other stays in front, order kept
                // within each group; which lane shades a path changes nothing about what is computed for it.
                bool back = false;
                if(mine) {
                    const int inst = RASTER == 1 ? -1 : __float_as_int(reinterpret_cast<const float4 *>(dp.hits + i)[1].x);
                    const int tri = RASTER == 1 ? -1 : __float_as_int(reinterpret_cast<const float4 *>(dp.hits + i)[0].w);
                    // escaped paths (a sky lookup, then done) behind the ones that hit something (the long branch); among
                    // those, when asked for, dielectrics behind conductors
                    back = tri < 0;
                    if(group_materials && tri >= 0 && inst >= 0 && inst < dp.n_instances) back = dp.inst[inst].material_type == 2u;
                }
                const unsigned wave = threadIdx.x >> 6;
                const unsigned long long m_back = __ballot(back), m_front = __ballot(mine && !back);
                if(lane == 0u) s_count[wave] = (unsigned)__popcll(m_front) | ((unsigned)__popcll(m_back) << 16);
                __syncthreads();
                unsigned front_before = 0, back_before = 0, front_total = 0;
                for(unsigned w = 0; w < BLOCK / 64; w++) {
                    const unsigned c = s_count[w];
                    if(w < wave) { front_before += c & 0xFFFFu; back_before += c >> 16; }
                    front_total += c & 0xFFFFu;
                }
                const unsigned lane_lt = __builtin_amdgcn_mbcnt_hi((unsigned)((back ? m_back : m_front) >> 32),
                                                                 __builtin_amdgcn_mbcnt_lo((unsigned)(back ? m_back : m_front), 0u));
                if(mine) s_slot[back ? front_total + back_before + lane_lt : front_before + lane_lt] = i;
                __syncthreads();
                const unsigned n_here = n - q0 < (unsigned)BLOCK ? n - q0 : (unsigned)BLOCK;
                mine = threadIdx.x < n_here;
                i = mine ? s_slot[threadIdx.x] : 0u;
                __syncthreads();  // (s_slot and s_count are reused by the next batch)
            }
            if(!mine) continue;
            bool continues, has_shadow;
            unsigned null_shadow;
            shade_vertex<RASTER, FIRST>(dp, dep, i, s_V, continues, has_shadow, null_shadow);
            my_null += null_shadow;
            // paths that end simply are not appended (queue compaction)
            s_stage[u * BLOCK + threadIdx.x] = i;
            cont_bits |= (continues ? 1u : 0u) << u;
            shad_bits |= (has_shadow ? 1u : 0u) << u;
        }
        // this wave's appends of the round: one counter update per queue
        unsigned tot_c = 0u, tot_s = 0u;
#pragma unroll 1
        for(unsigned u = 0; u < U; u++) {
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
        for(unsigned u = 0; u < U; u++) {
            const bool c = (cont_bits >> u) & 1u, sh = (shad_bits >> u) & 1u;
            const unsigned long long m_c = __ballot(c), m_s = __ballot(sh);
            const unsigned slot = (c || sh) ? s_stage[u * BLOCK + threadIdx.x] : 0u;
            if(c) queue_out[base_c + __builtin_amdgcn_mbcnt_hi((unsigned)(m_c >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_c, 0u))] = slot;
            if(sh) shadow_queue[base_s + __builtin_amdgcn_mbcnt_hi((unsigned)(m_s >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_s, 0u))] = slot;
            base_c += (unsigned)__popcll(m_c);
            base_s += (unsigned)__popcll(m_s);
        }
    }
    // one counter update per block (per wave they would queue up on one cache line: 32 K waves shade a 1080p vertex)
    if(my_null) atomicAdd(&s_null, my_null);
    __syncthreads();
    if(threadIdx.x == 0 && s_null) atomicAdd(&dp.stats[5], (unsigned long long)s_null);
}

// :524-525 and :551-552 for every pending light sample of a path vertex, sky first, then the
// lights in order; the last visible one of the first vertex becomes the direct light.
__global__ __launch_bounds__(SHADE_BLOCK) void shadow_apply_kernel(DrawParams dp, const unsigned *shadow_queue,
                                                                   const unsigned *shadow_count) {
    const unsigned n = *shadow_count;
    const unsigned nthreads = gridDim.x * blockDim.x;
    unsigned long long traced = 0;
    for(unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += nthreads) {
        const unsigned i = shadow_queue[q];
        const float4 t0 = *reinterpret_cast<const float4 *>(dp.thr + i);
        const V3 history = v3(t0.x, t0.y, t0.z);
        PathAccum *acc = dp.acc + i;
        V3 gi = v3(acc->gi[0], acc->gi[1], acc->gi[2]);
        for(int k = 0; k < dp.n_shadow; k++) {
            const float4 e1 = reinterpret_cast<const float4 *>(dp.shadow + (size_t)k * dp.npix + i)[1];
            const unsigned flags = __float_as_uint(e1.w);
            if(!(flags & 1u)) continue;
            traced++;
            if(!(flags & 4u)) continue;
            const V3 recv = v3(e1.x, e1.y, e1.z);
            gi = gi + history * recv;
            if(flags & 2u) {
                acc->di[0] = recv.x;
                acc->di[1] = recv.y;
                acc->di[2] = recv.z;
            }
        }
        acc->gi[0] = gi.x;
        acc->gi[1] = gi.y;
        acc->gi[2] = gi.z;
    }
    if(traced) atomicAdd(&dp.stats[2], traced);
}

// :633-649 and :657-680, for the samples of the draw in their order: one thread per pixel folds them into the outputs
// one after the other, as consecutive draws of one sample each would
__global__ __launch_bounds__(SHADE_BLOCK) void finalize_kernel(DrawParams dp) {
    const pt_params &P = dp.p;
    const unsigned nthreads = gridDim.x * blockDim.x;
    for(unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)dp.npix1; i += nthreads) {
        unsigned px, py, k0;
        if(!state_pixel(dp, i, px, py, k0)) continue;
        const size_t o3 = ((size_t)py * (unsigned)dp.W + px) * 3;
        V3 outDI = v3(0.0f), outIDI = v3(0.0f), outAlbedo = v3(0.0f);
        bool have = false;  // the outputs' values are in registers
        for(int k = 0; k < dp.batch; k++) {
            const float4 *a = reinterpret_cast<const float4 *>(dp.acc + (size_t)k * (unsigned)dp.npix1 + i);
            const float4 a0 = a[0], a1 = a[1], a2 = a[2];
            V3 resultDI = v3(a0.x, a0.y, a0.z), resultGI = v3(a1.x, a1.y, a1.z);
            const bool first_exist = a0.w != 0.0f;
            if(any_nan(resultDI)) resultDI = v3(0.0f);
            if(any_nan(resultGI)) resultGI = v3(0.0f);
            V3 resultIDI = resultGI - resultDI;
            V3 albedo = first_exist ? v3(a2.x, a2.y, a2.z) : resultGI;
            resultDI = resultDI / vmax(v3(S_EPS), albedo);
            resultIDI = resultIDI / vmax(v3(S_EPS), albedo);
            const unsigned spp = dp.samples[k].currentspp;
            if(spp != 1u) {
                if(!have) {
                    const PackedF3 d = *reinterpret_cast<const PackedF3 *>(dp.o_direct + o3), q = *reinterpret_cast<const PackedF3 *>(dp.o_indirect + o3),
                                   a = *reinterpret_cast<const PackedF3 *>(dp.o_albedo + o3);
                    outDI = v3(d.x, d.y, d.z);
                    outIDI = v3(q.x, q.y, q.z);
                    outAlbedo = v3(a.x, a.y, a.z);
                }
                if(P.accum_mode == 1) {
                    resultDI = outDI + resultDI;
                    resultIDI = outIDI + resultIDI;
                    albedo = outAlbedo + albedo;
                } else {
                    const float w = 1.0f / (float)spp;
                    resultDI = mixv(outDI, resultDI, w);
                    resultIDI = mixv(outIDI, resultIDI, w);
                    albedo = mixv(outAlbedo, albedo, w);
                }
            }
            outDI = resultDI; outIDI = resultIDI; outAlbedo = albedo;
            have = true;
        }
        PackedF3 w;
        w.x = outDI.x; w.y = outDI.y; w.z = outDI.z;
        *reinterpret_cast<PackedF3 *>(dp.o_direct + o3) = w;
        w.x = outIDI.x; w.y = outIDI.y; w.z = outIDI.z;
        *reinterpret_cast<PackedF3 *>(dp.o_indirect + o3) = w;
        w.x = outAlbedo.x; w.y = outAlbedo.y; w.z = outAlbedo.z;
        *reinterpret_cast<PackedF3 *>(dp.o_albedo + o3) = w;
    }
    if(dp.snapshot && blockIdx.x == 0 && threadIdx.x == 0) {
        // queue lengths of this draw for the host's choice of the next draw's tail vertex (QueueSnapshot)
        volatile QueueSnapshot *q = reinterpret_cast<volatile QueueSnapshot *>(dp.snapshot);
        const unsigned seq = q->seq_end + 1u;
        q->seq_begin = seq;
        __threadfence_system();
        q->max_depth = (unsigned)P.MAX_DEPTH;
        q->tail_from = (unsigned)dp.tail_from;
        q->npix = (unsigned)dp.npix;
        for(int d = 0; d <= P.MAX_DEPTH && d < 66; d++) q->alive[d] = dp.counters[d];
        __threadfence_system();
        q->seq_end = seq;
    }
}

int shade_grid(int n) {
    int b = (n + SHADE_BLOCK - 1) / SHADE_BLOCK;
    return b < 1 ? 1 : (b > 4096 ? 4096 : b);
}

}  // namespace

void launch_generate(hipStream_t s, const DrawParams &dp) {
    hipLaunchKernelGGL(generate_kernel, dim3(shade_grid(dp.npix)), dim3(SHADE_BLOCK), 0, s, dp);
}

void launch_shade(hipStream_t s, const DrawParams &dp, int dep, const unsigned *queue_in, const unsigned *count_in,
                  unsigned *queue_out, unsigned *count_out, unsigned *shadow_queue, unsigned *shadow_count, bool group, bool group_materials,
                  ShadeWhat what, const unsigned *valid_count) {
    const dim3 grid(shade_grid(dp.npix)), block(SHADE_BLOCK);  // (blocks beyond the queue's end leave at once)
    if(dep == 0 && dp.p.RasterizaionFor1st != 0)
        hipLaunchKernelGGL((shade_kernel<1, false, true>), grid, block, 0, s, dp, dep, queue_in, count_in, queue_out, count_out, shadow_queue, shadow_count, false, (int)what, valid_count);
    else if(dep == 0 && dp.raster_outputs)
        hipLaunchKernelGGL((shade_kernel<2, false, true>), grid, block, 0, s, dp, dep, queue_in, count_in, queue_out, count_out, shadow_queue, shadow_count, false, (int)what, valid_count);
    else if(group) {
        int b = (dp.npix + SHADE_GROUP_BLOCK - 1) / SHADE_GROUP_BLOCK;
        b = b < 1 ? 1 : (b > 4096 ? 4096 : b);
        if(dep == 0)
            hipLaunchKernelGGL((shade_kernel<0, true, true>), dim3(b), dim3(SHADE_GROUP_BLOCK), 0, s, dp, dep, queue_in, count_in, queue_out, count_out, shadow_queue, shadow_count, group_materials, (int)what, valid_count);
        else
            hipLaunchKernelGGL((shade_kernel<0, true, false>), dim3(b), dim3(SHADE_GROUP_BLOCK), 0, s, dp, dep, queue_in, count_in, queue_out, count_out, shadow_queue, shadow_count, group_materials, (int)what, valid_count);
    } else if(dep == 0)
        hipLaunchKernelGGL((shade_kernel<0, false, true>), grid, block, 0, s, dp, dep, queue_in, count_in, queue_out, count_out, shadow_queue, shadow_count, false, (int)what, valid_count);
    else
        hipLaunchKernelGGL((shade_kernel<0, false, false>), grid, block, 0, s, dp, dep, queue_in, count_in, queue_out, count_out, shadow_queue, shadow_count, false, (int)what, valid_count);
}

void launch_shadow_apply(hipStream_t s, const DrawParams &dp, const unsigned *shadow_queue,
                         const unsigned *shadow_count) {
    hipLaunchKernelGGL(shadow_apply_kernel, dim3(shade_grid(dp.npix)), dim3(SHADE_BLOCK), 0, s, dp, shadow_queue,
                       shadow_count);
}

void launch_finalize(hipStream_t s, const DrawParams &dp) {
    hipLaunchKernelGGL(finalize_kernel, dim3(shade_grid(dp.npix1)), dim3(SHADE_BLOCK), 0, s, dp);
}

}  // namespace rta
