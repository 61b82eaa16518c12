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
#include "kernels.h"
#include "shade_core.h"

namespace rta {
namespace {

#ifndef RT_SHADE_BLOCK
#define RT_SHADE_BLOCK 256
#endif
constexpr int SHADE_BLOCK = RT_SHADE_BLOCK;

__device__ const unsigned g_sobol_table[640] = {
#include "sobol_table.inc"
};

__device__ __forceinline__ void load_sobol_to_lds(unsigned *s_V) {
    for(int k = threadIdx.x; k < 640; k += blockDim.x) s_V[k] = g_sobol_table[k];
    __syncthreads();
}

__device__ __forceinline__ unsigned pixel_seed(const pt_params &p, unsigned px, unsigned py) {
    return (px * 1973u + py * 9277u + p.frameCounter * p.currentspp * 26699u) | 1u;  // :595-598
}

// camera ray, :604-608
__device__ __forceinline__ void camera_ray(const pt_params &p, float tan_half_fov, unsigned px, unsigned py, V3 &ori,
                                           V3 &dir) {
    const int W = p.SCREEN_W, H = p.SCREEN_H;
    float p_x = (float)px + p.jitter[0], p_y = (float)py + p.jitter[1];
    float disz = (float)W * 0.5f / tan_half_fov;
    const float *m = p.v2wMat;
    ori = v3(m[12], m[13], m[14]);
    float cx = p_x - (float)(W / 2), cy = p_y - (float)(H / 2), cz = -disz;
    V3 d;
    d.x = (m[0] * cx + m[4] * cy) + (m[8] * cz + m[12] * 0.0f);
    d.y = (m[1] * cx + m[5] * cy) + (m[9] * cz + m[13] * 0.0f);
    d.z = (m[2] * cx + m[6] * cy) + (m[10] * cz + m[14] * 0.0f);
    dir = normalize(d);
}

// slot -> pixel; false for the slots of an edge tile that lie outside the image
__device__ __forceinline__ bool state_pixel(const DrawParams &dp, unsigned i, unsigned &px, unsigned &py) {
    const unsigned tile = i >> 6, in = i & 63u;
    const unsigned ty = tile / (unsigned)dp.tiles_x, tx = tile - ty * (unsigned)dp.tiles_x;
    px = tx * STATE_TILE_W + in % STATE_TILE_W;
    py = ty * STATE_TILE_H + in / STATE_TILE_W;
    return px < (unsigned)dp.W && py < (unsigned)dp.H;
}

__global__ __launch_bounds__(SHADE_BLOCK) void generate_kernel(DrawParams dp) {
    const unsigned nthreads = gridDim.x * blockDim.x;
    for(unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)dp.npix; i += nthreads) {
        unsigned px, py;
        V3 o, d;
        if(state_pixel(dp, i, px, py)) {
            camera_ray(dp.p, dp.tan_half_fov, px, py, o, d);
        } else {
            // no pixel: a ray with a NaN direction ends as a miss without any traversal, and
            // shade() drops the slot
            o = v3(0.0f);
            d = v3(__uint_as_float(0x7FC00000u));
        }
        float4 *r = reinterpret_cast<float4 *>(dp.rays + i);
        r[0] = make_float4(o.x, o.y, o.z, 0.0f);
        r[1] = make_float4(d.x, d.y, d.z, 0.0f);
        float4 *t = reinterpret_cast<float4 *>(dp.thr + i);
        t[0] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        t[1] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(1u));  // sobolcurdim = 1, :602
        float4 *a = reinterpret_cast<float4 *>(dp.acc + i);
        a[0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        a[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        a[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if(blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&dp.stats[0], (unsigned long long)dp.W * (unsigned long long)dp.H);
        atomicAdd(&dp.stats[3], 1ull);
        dp.counters[0] = (unsigned)dp.npix;  // every slot is in the first queue
    }
}

__device__ __forceinline__ void write_shadow(ShadowEntry *e, V3 d, float dis2, V3 recv, unsigned flags) {
    float4 *p = reinterpret_cast<float4 *>(e);
    p[0] = make_float4(d.x, d.y, d.z, dis2);
    p[1] = make_float4(recv.x, recv.y, recv.z, __uint_as_float(flags));
}

// RASTER_FIRST: the instance launched for vertex 0 when the first hit comes from the raster images;
// kept apart so that the common instance carries none of its registers.
// Four waves per SIMD (128 registers, a few of them spilled) hide the latency of the gathers
// better than the three the kernel would get unconstrained (145 registers); five spill too much.
#ifndef SHADE_WAVES
#define SHADE_WAVES 4
#endif
template <bool RASTER_FIRST>
__global__ __launch_bounds__(SHADE_BLOCK) __attribute__((amdgpu_waves_per_eu(SHADE_WAVES, SHADE_WAVES))) void shade_kernel(DrawParams dp, int dep, const unsigned *queue_in,
                                                            const unsigned *count_in, unsigned *queue_out,
                                                            unsigned *count_out, unsigned *shadow_queue,
                                                            unsigned *shadow_count) {
    __shared__ unsigned s_V[640];
    const unsigned n = *count_in;
    if(blockIdx.x * blockDim.x >= n && blockIdx.x != 0) return;  // (late bounces fill few blocks; block 0 books the counters)
    load_sobol_to_lds(s_V);
    const pt_params &P = dp.p;
    const unsigned nthreads = gridDim.x * blockDim.x;
    const bool skyL = P.SkyboxLighting != 0, skyS = P.SkyboxSampling != 0;

    if(blockIdx.x == 0 && threadIdx.x == 0) {
        // (with RasterizaionFor1st the first vertex comes from images: no rays were traced for it)
        if(!RASTER_FIRST) atomicAdd(&dp.stats[1], dep == 0 ? (unsigned long long)dp.W * (unsigned long long)dp.H : (unsigned long long)n);
        if(dp.redo_counter_base >= 0) atomicAdd(&dp.stats[4], (unsigned long long)dp.counters[dp.redo_counter_base + dep]);
    }

    for(unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += nthreads) {
        const unsigned i = queue_in ? queue_in[q] : q;
        unsigned px, py;
        if(!state_pixel(dp, i, px, py)) continue;
        const size_t pix = (size_t)py * (unsigned)dp.W + px;  // index into the outputs

        const float4 *rp = reinterpret_cast<const float4 *>(dp.rays + i);
        const float4 r0 = rp[0], r1 = rp[1];
        const V3 ray_o = v3(r0.x, r0.y, r0.z), ray_d = v3(r1.x, r1.y, r1.z);
        const float prev_cos = r0.w;  // |wo.z| of the BSDF sample this ray follows
        // RasterizaionFor1st: the first hit comes from the raster pass's images (:606-624), no
        // primary ray was traced
        const bool raster0 = RASTER_FIRST;
        float hit_t, hit_u = 0.0f, hit_v = 0.0f;
        int hit_tri = -1, hit_inst;
        bool exist;
        V3 g_normal = v3(0.0f, 0.0f, 1.0f);
        float g_uvx = 0.0f, g_uvy = 0.0f;
        if(raster0) {
            const float suv_x0 = ((float)px + 0.5f) / (float)dp.W, suv_y0 = ((float)py + 0.5f) / (float)dp.H;
            const float wu = suv_x0 * (float)dp.W / (float)P.WINDOW_W;
            const float wv = (suv_y0 - 1.0f) * (float)dp.H / (float)P.WINDOW_H + 1.0f;
            hit_t = texture_rgb(dp.gb_depth, wu, wv).x;
            hit_inst = float_to_int(texture_nearest_r(dp.gb_inst, wu, wv) + 0.01f);
            g_normal = normalize(texture_rgb(dp.gb_normal, wu, wv));
            const V3 guv = texture_rgb(dp.gb_uv, wu, wv);
            g_uvx = guv.x;
            g_uvy = guv.y;
            // an index that names no instance is treated as "not covered" (out of bounds in the reference)
            exist = hit_t <= 100000.0f && hit_inst >= 0 && hit_inst < dp.n_instances;
        } else {
            const float4 *hp = reinterpret_cast<const float4 *>(dp.hits + i);
            const float4 h0 = hp[0], h1 = hp[1];
            hit_t = h0.x; hit_u = h0.y; hit_v = h0.z;
            hit_tri = __float_as_int(h0.w);
            hit_inst = __float_as_int(h1.x);
            exist = hit_tri >= 0;
        }
        float4 *tp = reinterpret_cast<float4 *>(dp.thr + i);
        const float4 t0 = tp[0], t1 = tp[1];
        V3 history = v3(t0.x, t0.y, t0.z);
        const float prev_pdf = t0.w;
        const V3 prev_fr = v3(t1.x, t1.y, t1.z);

        ShadeCtx cx;
        cx.materials = dp.materials;
        cx.textures = dp.textures;
        cx.n_textures = dp.n_textures;
        cx.sky = dp.sky;
        cx.sky_cache = dp.sky_cache;
        cx.sky_w = P.SKY_W;
        cx.sky_h = P.SKY_H;
        cx.sky_light_sum = P.skybox_Light_SUM;
        cx.rng.V = s_V;
        cx.rng.seed = pixel_seed(P, px, py);
        cx.rng.dim = __float_as_uint(t1.w);

        PathAccum *acc = dp.acc + i;

        if(dep == 0) {
            if(!exist) {
                // :496-499 and the G-buffer writes of :672-688 for a miss
                const V3 bg = cx.get_background_color(ray_d);
                float4 *a = reinterpret_cast<float4 *>(acc);
                a[0] = make_float4(bg.x, bg.y, bg.z, 0.0f);
                a[1] = make_float4(bg.x, bg.y, bg.z, 0.0f);
                // motion vector of the intersection's position, :652-655: (0,0,0) of the default
                // Intersection for a traced miss, ori + dir * depth for an uncovered raster pixel
                const V3 mp = raster0 ? ray_o + ray_d * hit_t : v3(0.0f);
                const float *bp = P.backprojMat;
                const float nx = (bp[0] * mp.x + bp[4] * mp.y) + (bp[8] * mp.z + bp[12]);
                const float ny = (bp[1] * mp.x + bp[5] * mp.y) + (bp[9] * mp.z + bp[13]);
                const float nw = (bp[3] * mp.x + bp[7] * mp.y) + (bp[11] * mp.z + bp[15]);
                const float suv_x = ((float)px + 0.5f) / (float)dp.W, suv_y = ((float)py + 0.5f) / (float)dp.H;
                dp.o_motion[pix * 2] = suv_x - (nx / nw + 1.0f) / 2.0f;
                dp.o_motion[pix * 2 + 1] = suv_y - (ny / nw + 1.0f) / 2.0f;
                dp.o_depth[pix] = raster0 ? hit_t : 1000000.0f;
                dp.o_normal[pix * 3] = g_normal.x;
                dp.o_normal[pix * 3 + 1] = g_normal.y;
                dp.o_normal[pix * 3 + 2] = g_normal.z;
                dp.o_inst[pix] = raster0 ? (float)hit_inst : -1.0f;
                continue;
            }
        } else {
            if(!exist) {  // :569-577 escape to the sky
                if(skyL) {
                    V3 recv = cx.get_background_color(ray_d) * prev_fr * prev_cos;
                    if(skyS) recv = recv / (prev_pdf + cx.skybox_sample_pdf(ray_d));
                    else recv = recv / prev_pdf;
                    const V3 add = history * recv;
                    acc->gi[0] += add.x;
                    acc->gi[1] += add.y;
                    acc->gi[2] += add.z;
                }
                continue;
            }
            history = history * (prev_fr * prev_cos / prev_pdf);  // :579
        }
        if(dep >= P.MAX_DEPTH && dep != 0) continue;  // loop bound of :501 (the first hit's outputs are due even at MAX_DEPTH 0)

        // ---- geometry of the hit (what intersect_triangle / intersect_meshBVH return, :239-248, :275-276)
        // first 64 bytes of the instance record: rows of world-to-local, then root, material pointer
        const float4 *ip = reinterpret_cast<const float4 *>(dp.inst + hit_inst);
        const float4 ir0 = ip[0], ir1 = ip[1], ir2 = ip[2], ir3 = ip[3];
        V3 normal = g_normal;
        float uvx = g_uvx, uvy = g_uvy;
        if(!raster0) {
            // floats 8..23 of the 96-byte record (16-byte aligned): normals 9..17, uvs 18..23
            const float4 *tr = reinterpret_cast<const float4 *>(dp.tris + hit_tri);
            const float4 ta = tr[2], tb = tr[3], tc = tr[4], td = tr[5];
            const float wgt = 1.0f - hit_u - hit_v;
            const V3 n0 = v3(ta.y, ta.z, ta.w), n1 = v3(tb.x, tb.y, tb.z), n2 = v3(tb.w, tc.x, tc.y);
            const V3 nloc = normalize(n0 * wgt + n1 * hit_u + n2 * hit_v);
            uvx = tc.z * wgt + td.x * hit_u + td.z * hit_v;
            uvy = tc.w * wgt + td.y * hit_u + td.w * hit_v;
            const V3 nn = v3(nloc.x * ir0.x + nloc.y * ir1.x + nloc.z * ir2.x,
                             nloc.x * ir0.y + nloc.y * ir1.y + nloc.z * ir2.y,
                             nloc.x * ir0.z + nloc.y * ir1.z + nloc.z * ir2.z);
            normal = normalize(nn);
        }
        const V3 position = ray_o + hit_t * ray_d;
        const int materialPtr = __float_as_int(ir3.y);  // InstAccel::material_ptr

        cx.bind_material(materialPtr, uvx, uvy);
        if(dep == 0) {
            // first-hit outputs, :640-655 and :682-688
            BSDFQueryRecord ab;
            ab.wi = v3(0.0f); ab.wo = v3(0.0f); ab.mptr = materialPtr; ab.uvx = uvx; ab.uvy = uvy; ab.eta = 1.0f;
            const V3 albedo = cx.albedo_material(ab);
            acc->albedo[0] = albedo.x;
            acc->albedo[1] = albedo.y;
            acc->albedo[2] = albedo.z;
            acc->first_exist = 1.0f;
            const float *bp = P.backprojMat;
            const float nx = (bp[0] * position.x + bp[4] * position.y) + (bp[8] * position.z + bp[12]);
            const float ny = (bp[1] * position.x + bp[5] * position.y) + (bp[9] * position.z + bp[13]);
            const float nw = (bp[3] * position.x + bp[7] * position.y) + (bp[11] * position.z + bp[15]);
            const float suv_x = ((float)px + 0.5f) / (float)dp.W, suv_y = ((float)py + 0.5f) / (float)dp.H;
            dp.o_motion[pix * 2] = suv_x - (nx / nw + 1.0f) / 2.0f;
            dp.o_motion[pix * 2 + 1] = suv_y - (ny / nw + 1.0f) / 2.0f;
            dp.o_depth[pix] = hit_t;
            dp.o_normal[pix * 3] = normal.x;
            dp.o_normal[pix * 3 + 1] = normal.y;
            dp.o_normal[pix * 3 + 2] = normal.z;
            dp.o_inst[pix] = (float)hit_inst;
            if(dep >= P.MAX_DEPTH) continue;
        }

        // ---- :507-513
        const Frame coord = create_coord(normal);
        BSDFQueryRecord bRec;
        bRec.wi = to_local(coord, -ray_d);
        bRec.wo = v3(0.0f);
        bRec.mptr = materialPtr;
        bRec.uvx = uvx;
        bRec.uvy = uvy;
        bRec.eta = 1.0f;
        const unsigned first_bit = dep == 0 ? 2u : 0u;
        bool any_shadow = false;
        float pdf;

        // ---- sky light sample, :516-528 (occlusion decided later by the shadow stage)
        {
            unsigned flags = 0u;
            V3 gwo = v3(0.0f), recv = v3(0.0f);
            if(skyL && skyS) {
                gwo = cx.skybox_sample(pdf);
                if(pdf > 0.0f) {
                    bRec.wo = to_local(coord, gwo);
                    const V3 fr = cx.eval_material(bRec);
                    recv = cx.get_background_color(gwo) * fr * fabsf(bRec.wo.z) / (pdf + cx.pdf_material(bRec));
                    flags = 1u | first_bit;
                    any_shadow = true;
                }
            }
            write_shadow(dp.shadow + i, gwo, S_INF, recv, flags);
        }
        // ---- analytic lights, :531-554
        for(int li = 0; li < P.lightCount; li++) {
            const LightInfo light = dp.lights[li];
            V3 dir, radiance;
            float dis2;
            if(light.type == 0) {
                dir = v3(light.x, light.y, light.z) - position;
                dis2 = dot(dir, dir);
                radiance = v3(light.r, light.g, light.b) / dis2;
                dir = normalize(dir);
            } else {
                dir = -normalize(v3(light.x, light.y, light.z));
                dis2 = S_INF;
                radiance = v3(light.r, light.g, light.b);
            }
            bRec.wo = to_local(coord, dir);
            const V3 fr = cx.eval_material(bRec);
            const V3 recv = radiance * fr * fabsf(bRec.wo.z);
            write_shadow(dp.shadow + (size_t)(1 + li) * dp.npix + i, dir, dis2, recv, 1u | first_bit);
            any_shadow = true;
        }

        // ---- next direction, :558-565
        V3 fr = cx.sample_material(bRec, pdf);
        if(!P.BRDFSampling) {
            bRec.wo = cx.spherical_sample(pdf);
            fr = cx.eval_material(bRec);
        }
        const V3 gwo = to_world(coord, bRec.wo);

        float4 *rw = reinterpret_cast<float4 *>(dp.rays + i);
        rw[0] = make_float4(position.x, position.y, position.z, fabsf(bRec.wo.z));
        rw[1] = make_float4(gwo.x, gwo.y, gwo.z, 0.0f);
        tp[0] = make_float4(history.x, history.y, history.z, pdf);
        tp[1] = make_float4(fr.x, fr.y, fr.z, __uint_as_float(cx.rng.dim));

        if(any_shadow) {
            const unsigned pos = atomicAdd(shadow_count, 1u);
            shadow_queue[pos] = i;
        }
        if(!(pdf <= 0.0f)) {  // `if(pdf <= 0) break;` -- a NaN pdf continues, as it does there
            const unsigned pos = atomicAdd(count_out, 1u);
            queue_out[pos] = i;
        }
    }
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

// :633-649 and :657-680
__global__ __launch_bounds__(SHADE_BLOCK) void finalize_kernel(DrawParams dp) {
    const pt_params &P = dp.p;
    const unsigned nthreads = gridDim.x * blockDim.x;
    for(unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)dp.npix; i += nthreads) {
        unsigned px, py;
        if(!state_pixel(dp, i, px, py)) continue;
        const float4 *a = reinterpret_cast<const float4 *>(dp.acc + i);
        const float4 a0 = a[0], a1 = a[1], a2 = a[2];
        V3 resultDI = v3(a0.x, a0.y, a0.z), resultGI = v3(a1.x, a1.y, a1.z);
        const bool first_exist = a0.w != 0.0f;
        if(any_nan(resultDI)) resultDI = v3(0.0f);
        if(any_nan(resultGI)) resultGI = v3(0.0f);
        V3 resultIDI = resultGI - resultDI;
        V3 albedo = first_exist ? v3(a2.x, a2.y, a2.z) : resultGI;
        resultDI = resultDI / vmax(v3(S_EPS), albedo);
        resultIDI = resultIDI / vmax(v3(S_EPS), albedo);
        const size_t o3 = ((size_t)py * (unsigned)dp.W + px) * 3;
        if(P.currentspp != 1u) {
            const V3 lastdi = v3(dp.o_direct[o3], dp.o_direct[o3 + 1], dp.o_direct[o3 + 2]);
            const V3 lastidi = v3(dp.o_indirect[o3], dp.o_indirect[o3 + 1], dp.o_indirect[o3 + 2]);
            const V3 lastalbedo = v3(dp.o_albedo[o3], dp.o_albedo[o3 + 1], dp.o_albedo[o3 + 2]);
            if(P.accum_mode == 1) {
                resultDI = lastdi + resultDI;
                resultIDI = lastidi + resultIDI;
                albedo = lastalbedo + albedo;
            } else {
                const float w = 1.0f / (float)P.currentspp;
                resultDI = mixv(lastdi, resultDI, w);
                resultIDI = mixv(lastidi, resultIDI, w);
                albedo = mixv(lastalbedo, albedo, w);
            }
        }
        dp.o_direct[o3] = resultDI.x; dp.o_direct[o3 + 1] = resultDI.y; dp.o_direct[o3 + 2] = resultDI.z;
        dp.o_indirect[o3] = resultIDI.x; dp.o_indirect[o3 + 1] = resultIDI.y; dp.o_indirect[o3 + 2] = resultIDI.z;
        dp.o_albedo[o3] = albedo.x; dp.o_albedo[o3 + 1] = albedo.y; dp.o_albedo[o3 + 2] = albedo.z;
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
                  unsigned *queue_out, unsigned *count_out, unsigned *shadow_queue, unsigned *shadow_count) {
    if(dep == 0 && dp.p.RasterizaionFor1st != 0)
        hipLaunchKernelGGL(shade_kernel<true>, dim3(shade_grid(dp.npix)), dim3(SHADE_BLOCK), 0, s, dp, dep, queue_in, count_in,
                           queue_out, count_out, shadow_queue, shadow_count);
    else
        hipLaunchKernelGGL(shade_kernel<false>, dim3(shade_grid(dp.npix)), dim3(SHADE_BLOCK), 0, s, dp, dep, queue_in, count_in,
                           queue_out, count_out, shadow_queue, shadow_count);
}

void launch_shadow_apply(hipStream_t s, const DrawParams &dp, const unsigned *shadow_queue,
                         const unsigned *shadow_count) {
    hipLaunchKernelGGL(shadow_apply_kernel, dim3(shade_grid(dp.npix)), dim3(SHADE_BLOCK), 0, s, dp, shadow_queue,
                       shadow_count);
}

void launch_finalize(hipStream_t s, const DrawParams &dp) {
    hipLaunchKernelGGL(finalize_kernel, dim3(shade_grid(dp.npix)), dim3(SHADE_BLOCK), 0, s, dp);
}

}  // namespace rta
