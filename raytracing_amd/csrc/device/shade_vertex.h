// Everything the reference's fragment shader does at ONE path vertex (shade_mis_advanced,
// shader/pathtracing.glsl:489-582, and the first-hit outputs of main(), :633-688), as a device
// function over the path state of one slot.  Users: shade_kernel (one launch per vertex for all
// live paths, path_kernels.hip) and path_tail_kernel (a wave carries its paths through all their
// remaining vertices, tail_kernels.hip).
#pragma once
#include "kernels.h"
#include "shade_core.h"

namespace rta {

__device__ __forceinline__ unsigned pixel_seed(unsigned seed_term, unsigned px, unsigned py) {
    return (px * 1973u + py * 9277u + seed_term) | 1u;  // :595-598, seed_term = frameCounter * currentspp * 26699
}

// camera ray, :604-608
__device__ __forceinline__ void camera_ray(const pt_params &p, float jitter_x, float jitter_y, float tan_half_fov, unsigned px,
                                           unsigned py, V3 &ori, V3 &dir) {
    const int W = p.SCREEN_W, H = p.SCREEN_H;
    float p_x = (float)px + jitter_x, p_y = (float)py + jitter_y;
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

// slot -> (sample of the draw, pixel); false for the slots of an edge tile that lie outside the image
__device__ __forceinline__ bool state_pixel(const DrawParams &dp, unsigned i, unsigned &px, unsigned &py, unsigned &k) {
    k = 0u;
    if(dp.batch > 1) {
        k = i / (unsigned)dp.npix1;
        i -= k * (unsigned)dp.npix1;
    }
    const unsigned tile = i >> 6, in = i & 63u;
    const unsigned ty = tile / (unsigned)dp.tiles_x, tx = tile - ty * (unsigned)dp.tiles_x;
    px = tx * STATE_TILE_W + in % STATE_TILE_W;
    py = ty * STATE_TILE_H + in / STATE_TILE_W;
    return px < (unsigned)dp.W && py < (unsigned)dp.H;
}

// the four last-write outputs of a pixel (:672-688), one store each: the eight pixels of a tile row are neighbours in
// every one of them
__device__ __forceinline__ void store_gbuffer(const DrawParams &dp, size_t pix, float motion_x, float motion_y, float depth, V3 normal,
                                              float instance) {
    *reinterpret_cast<float2 *>(dp.o_motion + pix * 2) = make_float2(motion_x, motion_y);
    dp.o_depth[pix] = depth;
    PackedF3 n;
    n.x = normal.x; n.y = normal.y; n.z = normal.z;
    *reinterpret_cast<PackedF3 *>(dp.o_normal + pix * 3) = n;
    dp.o_inst[pix] = instance;
}

__device__ __forceinline__ void write_shadow(ShadowEntry *e, V3 d, float dis2, V3 recv, unsigned flags) {
    float4 *p = reinterpret_cast<float4 *>(e);
    p[0] = make_float4(d.x, d.y, d.z, dis2);
    p[1] = make_float4(recv.x, recv.y, recv.z, __uint_as_float(flags));
}


// Shades vertex `dep` of the path in slot i (its ray in dp.rays[i], its closest hit in dp.hits[i],
// or, RASTER_FIRST, the raster images).  continues: the path goes on with the ray now in
// dp.rays[i]; has_shadow: dp.shadow[..][i] holds light samples whose occlusion rays are due;
// null_shadow: how many light samples of the vertex are exactly zero, their occlusion queries not due (the caller
// counts them).
// s_T: the Sobol nibble table in LDS (load_sobol_to_lds).
// RASTER: what the raster pass's images are used for at vertex 0 (shader/pathtracing.glsl:612-628, :641, :684)
//   0  nothing (no raster pass behind this host: depth and albedo come from the traced hit);
//   1  RasterizaionFor1st: they ARE the first hit, no primary ray was traced;
//   2  RasterizaionFor1st off with the depth and instance-index images bound: the first hit is traced, but the
//      depth output is the raster depth and the albedo is looked up through the raster instance index, as there.
// FIRST: dep == 0, known when the kernel is compiled: the launch of vertex 0 carries none of the code of later vertices
// (throughput update, escape to the sky) and the launches of later vertices none of the first-hit outputs.
// given: the closest hit of the slot's ray when the caller holds it in registers (the launch that traces the camera rays
// and shades their hits in one go); null: it is read from dp.hits[i].
// next_dir: when not null and the path continues, the direction of its next ray (what was stored in dp.rays[i]);
// sky_dir: when not null, the direction of the sky's light sample (what was stored in dp.shadow[i]).
template <int RASTER, bool FIRST>
__device__ __forceinline__ void shade_vertex(const DrawParams &dp, int dep, unsigned i, const unsigned *s_T, bool &continues,
                                             bool &has_shadow, unsigned &null_shadow, const LaneHit *given = nullptr,
                                             V3 *next_dir = nullptr, V3 *sky_dir = nullptr) {
    continues = false;
    has_shadow = false;
    null_shadow = 0u;
    const pt_params &P = dp.p;
    const bool skyL = P.SkyboxLighting != 0, skyS = P.SkyboxSampling != 0;
    unsigned px, py, sample_k;
    if(!state_pixel(dp, i, px, py, sample_k)) return;
    const size_t pix = (size_t)py * (unsigned)dp.W + px;  // index into the outputs
    // motion, depth, normal and instance index keep what the LAST sample wrote (:672-688 overwrite them per draw)
    const bool writes_gbuffer = sample_k + 1u == (unsigned)dp.batch;

    // Vertex 0 needs nothing of the path state but the hit: its ray is the pixel's camera ray, made again here with the
    // arithmetic that made it for the traversal (no rounding depends on where it runs), its throughput is 1, its sampler
    // dimension 1 (:602), its accumulators start at zero -- so the launches that make the camera rays store none of that.
    V3 ray_o, ray_d;
    float prev_cos = 0.0f;  // |wo.z| of the BSDF sample this ray follows
    if(FIRST) {
        camera_ray(P, dp.samples[sample_k].jitter[0], dp.samples[sample_k].jitter[1], dp.tan_half_fov, px, py, ray_o, ray_d);
    } else {
        const float4 *rp = reinterpret_cast<const float4 *>(dp.rays + i);
        const float4 r0 = rp[0], r1 = rp[1];
        ray_o = v3(r0.x, r0.y, r0.z);
        ray_d = v3(r1.x, r1.y, r1.z);
        prev_cos = r0.w;
    }
    // RasterizaionFor1st: the first hit comes from the raster pass's images (:606-624), no
    // primary ray was traced
    const bool raster0 = RASTER == 1;
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
    } else if(given) {
        hit_t = given->t; hit_u = given->u; hit_v = given->v;
        hit_tri = given->tri;
        hit_inst = given->inst;
        exist = hit_tri >= 0;
    } else {
        const float4 *hp = reinterpret_cast<const float4 *>(dp.hits + i);
        const float4 h0 = hp[0], h1 = hp[1];
        hit_t = h0.x; hit_u = h0.y; hit_v = h0.z;
        hit_tri = __float_as_int(h0.w);
        hit_inst = __float_as_int(h1.x);
        exist = hit_tri >= 0;
    }
    float4 *tp = reinterpret_cast<float4 *>(dp.thr + i);
    float4 t0 = make_float4(1.0f, 1.0f, 1.0f, 0.0f), t1 = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(1u));
    if(!FIRST) {
        t0 = tp[0];
        t1 = tp[1];
    }
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
    cx.rng.T = s_T;
    cx.rng.seed = pixel_seed(dp.samples[sample_k].seed_term, px, py);
    cx.rng.dim = __float_as_uint(t1.w);

    PathAccum *acc = dp.acc + i;

    float raster_depth = 0.0f;
    int raster_inst = -1;
    if(RASTER == 2 && FIRST) {
        const float suv_x0 = ((float)px + 0.5f) / (float)dp.W, suv_y0 = ((float)py + 0.5f) / (float)dp.H;
        const float wu = suv_x0 * (float)dp.W / (float)P.WINDOW_W;
        const float wv = (suv_y0 - 1.0f) * (float)dp.H / (float)P.WINDOW_H + 1.0f;
        raster_depth = texture_rgb(dp.gb_depth, wu, wv).x;
        raster_inst = float_to_int(texture_nearest_r(dp.gb_inst, wu, wv) + 0.01f);
    }

    if(FIRST) {
        if(!exist) {
            // :496-499 and the G-buffer writes of :672-688 for a miss
            const V3 bg = cx.get_background_color(ray_d);
            float4 *a = reinterpret_cast<float4 *>(acc);
            a[0] = make_float4(bg.x, bg.y, bg.z, 0.0f);
            a[1] = make_float4(bg.x, bg.y, bg.z, 0.0f);
            // motion vector of the intersection's position, :652-655: (0,0,0) of the default
            // Intersection for a traced miss, ori + dir * depth for an uncovered raster pixel
            if(!writes_gbuffer) return;
            const V3 mp = raster0 ? ray_o + ray_d * hit_t : v3(0.0f);
            const float *bp = P.backprojMat;
            const float nx = (bp[0] * mp.x + bp[4] * mp.y) + (bp[8] * mp.z + bp[12]);
            const float ny = (bp[1] * mp.x + bp[5] * mp.y) + (bp[9] * mp.z + bp[13]);
            const float nw = (bp[3] * mp.x + bp[7] * mp.y) + (bp[11] * mp.z + bp[15]);
            const float suv_x = ((float)px + 0.5f) / (float)dp.W, suv_y = ((float)py + 0.5f) / (float)dp.H;
            store_gbuffer(dp, pix, suv_x - (nx / nw + 1.0f) / 2.0f, suv_y - (ny / nw + 1.0f) / 2.0f,
                          raster0 ? hit_t : (RASTER == 2 ? raster_depth : 1000000.0f), g_normal, raster0 ? (float)hit_inst : -1.0f);
            return;
        }
    } else {
        if(!exist) {  // :569-577 escape to the sky
            if(skyL) {
                V3 recv = cx.get_background_color(ray_d) * prev_fr * prev_cos;
                if(skyS) recv = recv / (prev_pdf + cx.skybox_sample_pdf(ray_d));
                else recv = recv / prev_pdf;
                const V3 add = history * recv;
                float4 *ag = reinterpret_cast<float4 *>(acc) + 1;   // gi and its padding word: one load, one store
                float4 g = *ag;
                g.x += add.x;
                g.y += add.y;
                g.z += add.z;
                *ag = g;
            }
            return;
        }
        history = history * (prev_fr * prev_cos / prev_pdf);  // :579
    }
    if(!FIRST && dep >= P.MAX_DEPTH) return;  // loop bound of :501 (the first hit's outputs are due even at MAX_DEPTH 0)

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
    if(FIRST) {
        // first-hit outputs, :640-655 and :682-688
        BSDFQueryRecord ab;
        ab.wi = v3(0.0f); ab.wo = v3(0.0f); ab.mptr = materialPtr; ab.uvx = uvx; ab.uvy = uvy; ab.eta = 1.0f;
        V3 albedo;
        if(RASTER == 2 && raster_inst >= 0 && raster_inst < dp.n_instances && raster_inst != hit_inst) {
            // :641 indexes the instance records with the RASTER index (one that names no instance is out of
            // bounds there; the traced one stands in)
            ab.mptr = __float_as_int(reinterpret_cast<const float4 *>(dp.inst + raster_inst)[3].y);
            cx.bind_material(ab.mptr, uvx, uvy);
            albedo = cx.albedo_material(ab);
            cx.bind_material(materialPtr, uvx, uvy);
        } else {
            albedo = cx.albedo_material(ab);
        }
        {
            // (direct and accumulated light start at zero; first_exist = 1)
            float4 *a = reinterpret_cast<float4 *>(acc);
            a[0] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
            a[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            a[2] = make_float4(albedo.x, albedo.y, albedo.z, 0.0f);
        }
        if(writes_gbuffer) {
            const float *bp = P.backprojMat;
            const float nx = (bp[0] * position.x + bp[4] * position.y) + (bp[8] * position.z + bp[12]);
            const float ny = (bp[1] * position.x + bp[5] * position.y) + (bp[9] * position.z + bp[13]);
            const float nw = (bp[3] * position.x + bp[7] * position.y) + (bp[11] * position.z + bp[15]);
            const float suv_x = ((float)px + 0.5f) / (float)dp.W, suv_y = ((float)py + 0.5f) / (float)dp.H;
            store_gbuffer(dp, pix, suv_x - (nx / nw + 1.0f) / 2.0f, suv_y - (ny / nw + 1.0f) / 2.0f,
                          RASTER == 2 ? raster_depth : hit_t, normal, (float)hit_inst);
        }
        if(dep >= P.MAX_DEPTH) return;
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
    const unsigned first_bit = FIRST ? 2u : 0u;
    bool any_shadow = false, null_sample = false;
    unsigned null_count = 0u;  // light samples of this vertex that are exactly zero
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
                // A sample that is +0 in every component (a direction below the surface of a conductor: half of what
                // an importance-sampled sky offers a wall) changes nothing whether it is visible or not: x + (+-0) = x
                // for every x the accumulator can hold (it starts as +0), and the direct light of vertex 0 starts as
                // the +0 it would be set to.  Its occlusion query is counted, not traced -- unless the throughput is
                // not finite (0 * inf is NaN in the reference's sum).
                null_sample = (__float_as_uint(recv.x) | __float_as_uint(recv.y) | __float_as_uint(recv.z)) == 0u &&
                              fabsf(history.x) <= 3.4028235e38f && fabsf(history.y) <= 3.4028235e38f && fabsf(history.z) <= 3.4028235e38f;
                if(!null_sample) {
                    flags = 1u | first_bit;
                    any_shadow = true;
                }
            }
        }
        write_shadow(dp.shadow + i, gwo, S_INF, recv, flags);
        if(sky_dir) *sky_dir = gwo;
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
        // (a sample that is exactly zero -- a light behind the surface -- as for the sky's above; not at vertex 0,
        // where a visible sample, zero or not, replaces the direct light an earlier one has set, :551-552)
        const bool null_light = !FIRST && (__float_as_uint(recv.x) | __float_as_uint(recv.y) | __float_as_uint(recv.z)) == 0u &&
                                fabsf(history.x) <= 3.4028235e38f && fabsf(history.y) <= 3.4028235e38f && fabsf(history.z) <= 3.4028235e38f;
        write_shadow(dp.shadow + (size_t)(1 + li) * dp.npix + i, dir, dis2, recv, null_light ? 0u : (1u | first_bit));
        if(null_light) null_count++;
        else any_shadow = true;
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
    if(next_dir) *next_dir = gwo;
    tp[0] = make_float4(history.x, history.y, history.z, pdf);
    tp[1] = make_float4(fr.x, fr.y, fr.z, __uint_as_float(cx.rng.dim));

    has_shadow = any_shadow;
    null_shadow = null_count + (null_sample ? 1u : 0u);
    continues = !(pdf <= 0.0f);  // `if(pdf <= 0) break;` -- a NaN pdf continues, as it does there
}

}  // namespace rta
