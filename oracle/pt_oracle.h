/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's path-tracing pass (shader/pathtracing.glsl and the
 * files it includes), used as the checker for the HIP path and as the CPU baseline in
 * bench.py.  Nothing in the product library includes, links or calls this; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may.
 *
 * Pinning: the reference has no tests or golden vectors for this path and its GLSL cannot
 * be executed as GLSL in the build container (no GL context).  It is pinned by running the
 * reference's own source text, compiled here:
 *   - its C++ (oracle/_ref/ref_harness): the BVH arrays this oracle traverses, the slab /
 *     Moller-Trumbore arithmetic (against the CPU picking path BVH.cpp:89-114, which differs
 *     from the shader only in strict-vs-inclusive comparisons and visiting order),
 *     glm::inverse, RGBE and JPEG decode;
 *   - its GLSL compiled as C++ against the glm headers it vendors (oracle/glsl_kat.py,
 *     oracle/glsl_frame.py; fixtures tests/golden/glsl_bsdf_kat.npz, glsl_frame_casa.npz):
 *     both BSDFs and the Sobol sampler query by query, and the whole fragment shader one
 *     main() per pixel on eight configurations (raster images in play and RasterizaionFor1st
 *     on among them).  With the shader's sin / cos / atan / asin / pow resolved to the recipe
 *     below, every output word of those frames equals this oracle's.  The post-processing
 *     restatements (orc_post_*, orc_svgf_*, orc_taa) are pinned the same way against
 *     shader/postprocessing/*.glsl (oracle/glsl_post.py, tests/golden/glsl_post.npz).
 * Still unpinned: what GLSL itself leaves open (a driver's transcendentals and texture
 * filtering), and the inputs of the raster pass.
 *
 * Conventions chosen where GLSL leaves behaviour open (the HIP path makes the same ones):
 *   - no fused multiply-add anywhere (build with -ffp-contract=off);
 *   - a / b is IEEE division, sqrt is IEEE, normalize(v) = v * (1 / sqrt(dot(v, v)));
 *   - min / max drop a NaN operand (fminf / fmaxf);
 *   - sin, cos, atan, asin and pow (GLSL bounds their error, not their value) follow one
 *     fixed fp32 recipe (pt_oracle.cpp, "sin / cos / atan / asin / pow"), abs. error ~1e-7;
 *     tan(fov/2) is the C library's;
 *   - mat4 * vec4 associates as (m0*x + m1*y) + (m2*z + m3*w) like glm;
 *   - texture(): GL 4.6 section 8.14 bilinear filter evaluated in fp32, REPEAT wrap;
 *   - the varying screen_uv is ((x + 0.5) / W, (y + 0.5) / H);
 *   - Sobol dimensions >= 20 (out of bounds in the reference, sobol_matrix.glsl:25):
 *     dimension d uses table row 1 + (d - 1) % 19 with the index replaced by
 *     wang_hash(index + 0x9E3779B9 * ((d - 1) / 19)); rows 1..19 are untouched, so every
 *     draw the reference defines is reproduced bit for bit;
 *   - with RasterizaionFor1st off the reference still writes the RASTER depth and looks the
 *     albedo up through the RASTER instance index (pathtracing.glsl:615-616, :641, :684).  When
 *     the depth and instance-index images are given, that is reproduced (an index that names
 *     no instance, an out-of-bounds read there, falls back to the traced one); when they are
 *     not (no raster pass), `depth` is the primary hit's t (1e6 on a miss, the raster clear
 *     value) and the albedo lookup uses the traced primary hit.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_texture {
    const void *data; /* u8 (format 0) or f32 (format 1), row 0 = t 0 */
    int32_t width, height, channels, format;
} orc_texture;

typedef struct orc_scene {
    const float *materials;      /* slot 1 */
    const float *triangles;      /* slot 2: 24 floats each */
    const void *instances;       /* slot 3: 112-byte records */
    const void *mesh_nodes;      /* slot 4: 48-byte nodes */
    const void *scene_nodes;     /* slot 5 */
    const void *lights;          /* slot 6: 32-byte records */
    const orc_texture *textures; /* what slot 0's handles point at */
    int32_t n_textures;
    orc_texture skybox;            /* f32 RGB */
    orc_texture skybox_samplecache; /* f32 RGB */
    /* RasterizaionFor1st: what RasterPass draws (raster_ps.glsl:17-27), f32: distance along
     * the primary ray (R), world normal (RGB, not normalised), uv (RG), instance index (R);
     * cleared to 1e6 where nothing was drawn.  Only read when params->RasterizaionFor1st. */
    orc_texture depthGBufferTexture, normalGBufferTexture, uvGBufferTexture, instanceIndexGBufferTexture;
    int32_t n_instances; /* records in `instances`; bounds the raster instance index */
} orc_scene;

/* same field order as pt_params in include/raytracing_amd.h */
typedef struct orc_params {
    float jitter[2];
    uint32_t currentspp;
    float skybox_Light_SUM;
    float v2wMat[16];
    int32_t SCREEN_W, SCREEN_H, WINDOW_W, WINDOW_H;
    int32_t MAX_DEPTH;
    float cameraNear, cameraFar;
    uint32_t frameCounter;
    float fov;
    int32_t SKY_W, SKY_H;
    float backprojMat[16];
    int32_t BRDFSampling, SkyboxSampling, SkyboxLighting, RasterizaionFor1st;
    int32_t lightCount;
    int32_t accum_mode;
    int32_t reserved[7];
} orc_params;

typedef struct orc_hit {
    int32_t instance, triangle;
    float t, u, v;
} orc_hit;

/* Work counters of the REFERENCE traversal order (binary tree, right child first,
 * prune on tnear > t_best, one triangle per leaf, no leaf box test). */
typedef struct orc_counters {
    uint64_t rays;             /* intersect_sceneBVH calls                          */
    uint64_t scene_nodes;      /* scene-level nodes popped                          */
    uint64_t mesh_inner_nodes; /* mesh-level inner nodes popped                     */
    uint64_t leaf_tests;       /* triangle leaves popped = triangle tests           */
    uint64_t hits;             /* rays with exist = true                            */
    uint64_t max_stack;        /* deepest stack height seen                         */
    uint64_t samples;          /* pixels shaded                                     */
    uint64_t rays_extend;      /* primary + indirect                                */
    uint64_t rays_shadow;      /* sky NEE + light shadow rays                       */
    uint64_t shaded_vertices;  /* bounce-loop iterations (InstanceInfo fetches)     */
    uint64_t material_evals;   /* eval/pdf/sample/albedo dispatches                 */
    uint64_t ldr_tex_fetches;  /* bilinear fetches of 8-bit textures                */
    uint64_t sky_fetches;      /* bilinear fetches of sky or sample cache           */
    uint64_t ext_nodes;        /* nodes popped (scene + mesh inner + leaf) by extend rays only */
    uint64_t ext_leaf_tests;   /* triangle tests by extend rays only                */
} orc_counters;

/* intersect_sceneBVH (pathtracing.glsl:292-316) for n rays (6 floats each). */
void orc_intersect(const orc_scene *scene, const float *rays, size_t n, orc_hit *hits, orc_counters *counters,
                   int n_threads);

/* main() (pathtracing.glsl:585-691) for every pixel: one sample per pixel.  The three
 * accumulated outputs are read when currentspp != 1.  Any output pointer may be NULL. */
typedef struct orc_outputs {
    float *directLum, *indirectLum, *albedo; /* W*H*3 */
    float *motion;                           /* W*H*2 */
    float *depth;                            /* W*H   */
    float *normal;                           /* W*H*3 */
    float *instanceIndex;                    /* W*H   */
} orc_outputs;
void orc_render(const orc_scene *scene, const orc_params *params, const orc_outputs *out, orc_counters *counters,
                int n_threads);

/* known-answer hooks */
float orc_sobol(uint32_t d, uint32_t i);
/* which: 0 eval (3 floats), 1 pdf (1), 2 sample (3 value + pdf + wo[3] + eta = 8; consumes
 * rand draws for index `seed` starting at dimension `dim`), 3 albedo (3) */
void orc_material(const orc_scene *scene, int which, int mptr, const float wi[3], const float wo[3],
                  const float uv[2], uint32_t seed, uint32_t dim, float *out);
void orc_texture_fetch(const orc_texture *tex, float s, float t, float out[4]);
void orc_background(const orc_scene *scene, const float dir[3], float out[3]);
/* out: wi[3], pdf */
void orc_skybox_sample(const orc_scene *scene, const orc_params *params, uint32_t seed, uint32_t dim, float out[4]);
float orc_skybox_sample_pdf(const orc_scene *scene, const orc_params *params, const float dir[3]);

/* The element-wise consumers of the outputs:
 * SVGF_Merge.glsl:24-52, StaticBlender.glsl:26-61, ToneMappingGamma.glsl:17-54 and the
 * float -> byte + flip step of SSBOBuffer::save_as_image (src/SSBOBuffer.h:47-56). */
void orc_post_merge(size_t npix, const float *di, const float *idi, const float *albedo, float *color);
void orc_post_static_blend(size_t npix, uint32_t frame, const float *color, float *hist_color, float *hist_moment);
void orc_post_tonemap(size_t n, const float *in, float *out);
void orc_quantize_image(const float *data, int w, int h, int c, uint8_t *out);

/* The denoising chain, one call per shader dispatch (buffers as the shaders bind them):
 * SVGF_TemporalFilter.glsl:124-200, SVGF_VarianceFilter.glsl:30-103, SVGF_SpatialFilter.glsl:32-90,
 * TAA.glsl:124-190. */
void orc_svgf_temporal(int W, int H, int first_frame, const float *color, const float *normal, const float *inst,
                       const float *motion, float *o_color, float *o_moment, float *o_num, const float *h_color,
                       const float *h_moment, const float *h_num, const float *h_normal, const float *h_inst);
void orc_svgf_variance(int W, int H, const float *moment, const float *normal, const float *depth, const float *num,
                       float *variance);
void orc_svgf_spatial(int W, int H, int step, const float *color, const float *normal, const float *depth,
                      const float *variance, float *out);
void orc_taa(int W, int H, const float *color, const float *h_color, const float *motion, const float *normal,
             const float *inst, const float *h_normal, const float *h_inst, float *out);

#ifdef __cplusplus
}
#endif
#endif
