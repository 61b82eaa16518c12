/* raytracing_amd.h -- C ABI of the MI355X (gfx950) path-tracing pass.
 *
 * Drop-in boundary for ONE path of lxtyin/Raytracing: the per-pixel path-tracing pass
 * that the reference runs as an OpenGL fragment shader behind its
 * Renderer / ResourceManager / SSBOBuffer classes.  Every entry point below names
 * the reference interface it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - every function returns PT_OK (0) or a negative pt_status; pt_last_error()
 *     returns a thread-local message for the last failure on the calling thread.
 *     (The reference has no error channel at all: GL errors are dropped and shader
 *     failures go to std::cerr, src/renderpass/RenderPass.cpp:60-84.)
 *   - plain pointers and sizes only; handles are opaque.
 *   - one context per GPU; calls on a context are serialised by the caller, as the
 *     reference's single GL thread serialises its own (main.cpp:153-354).
 *   - draw / intersect are asynchronous.  Draws run on internal streams of the pass; every
 *     other call of the library, pt_ctx_stream included, first orders the context's main
 *     stream behind the draws issued so far.  pt_buf_download and pt_ctx_synchronize wait
 *     for everything.  Work of the caller's own (events, kernels or collectives on
 *     pt_buf_device_ptr memory) must be queued on the stream pt_ctx_stream returned AFTER
 *     the draws it depends on were issued, or follow pt_ctx_synchronize.
 *   - a traversal that runs out of stack (never with arrays made by the reference's builder,
 *     whose tree heights size the stack) is reported as PT_ERR_STATE by the next call that
 *     waits for the device: pt_ctx_synchronize, pt_buf_download, pt_pass_stats, pt_pass_profile.
 *   - the host-side helpers (pt_host_*) never touch a GPU.
 */
#ifndef RAYTRACING_AMD_H
#define RAYTRACING_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID = -1,   /* bad argument / handle / size */
    PT_ERR_HIP = -2,       /* a HIP runtime call failed; message has hipGetErrorString */
    PT_ERR_STATE = -3,     /* call sequence error, e.g. draw with a scene slot unbound */
    PT_ERR_UNSUPPORTED = -4
} pt_status;

typedef struct pt_ctx pt_ctx;
typedef struct pt_buf pt_buf;
typedef struct pt_tex pt_tex;
typedef struct pt_pass pt_pass;

const char *pt_last_error(void);
/* Library version string, e.g. "raytracing_amd 0.1 (gfx950)". */
const char *pt_version(void);

/* ---- context: replaces the GL context created in main.cpp:400-416 ------------------ */
int pt_ctx_create(int device_id, pt_ctx **out);
int pt_ctx_destroy(pt_ctx *ctx);
int pt_ctx_synchronize(pt_ctx *ctx);
/* The context's main hipStream_t, ordered behind every draw issued so far by this very call
 * (draws run on streams of their own): what the caller queues on it afterwards -- timing
 * events, copies, kernels, collectives -- runs after those draws.  Call it again after later
 * draws.  NULL on failure. */
void *pt_ctx_stream(pt_ctx *ctx);

/* ---- buffers: SSBOBuffer<T>, src/SSBOBuffer.h:19-150 ------------------------------- */
/* SSBOBuffer(uint siz) / SSBOBuffer(uint siz, T* data), :93-124.  bytes may be 0
 * (the reference creates zero-length buffers, e.g. no lights); host may be NULL. */
int pt_buf_create(pt_ctx *ctx, size_t bytes, const void *host, pt_buf **out);
/* release(), :81-85 */
int pt_buf_release(pt_buf *buf);
/* copy(const T* data), :64-70: overwrite the whole buffer from host memory */
int pt_buf_upload(pt_buf *buf, const void *host, size_t bytes);
/* copy(const SSBOBuffer<T>*), :132-143: device to device, sizes must match */
int pt_buf_copy(pt_buf *dst, const pt_buf *src);
/* fill(const T& val), :72-79: repeat one element of elem_bytes (1, 2, 4, 8, 12 or 16) */
int pt_buf_fill(pt_buf *buf, const void *elem, size_t elem_bytes);
/* the glMapBufferRange read-back of save_as_image, :45-46 (synchronises the stream) */
int pt_buf_download(const pt_buf *buf, void *host, size_t bytes);
size_t pt_buf_bytes(const pt_buf *buf);
/* raw device address, for interop (torch / RCCL); stays valid until release */
void *pt_buf_device_ptr(const pt_buf *buf);

/* ---- textures: Texture::load_to_gpu, src/texture/Texture.cpp:7-36 and the two float
 *      sky textures of Skybox::load_to_gpu, src/texture/Skybox.cpp:55-84 ------------- */
typedef enum pt_tex_format { PT_TEX_U8 = 0, PT_TEX_F32 = 1 } pt_tex_format;
/* REPEAT wrap, LINEAR filter, no mipmaps in use (Texture.cpp:24-27).  Row 0 of data is
 * texture coordinate t = 0.  channels: 1, 3 or 4. */
int pt_tex_create(pt_ctx *ctx, int width, int height, int channels, int format, const void *data, pt_tex **out);
int pt_tex_release(pt_tex *tex);
/* the 64-bit value to store in the texture-handle buffer (slot 0); plays the role of
 * glGetTextureHandleARB, Texture.cpp:31.  Never 0 for a live texture. */
uint64_t pt_tex_handle(const pt_tex *tex);

/* ---- the pass: Renderer, src/renderpass/Renderer.{h,cpp} ----------------------------- */
/* Parameters that reach the reference shader as GL uniforms (main.cpp:185-205,
 * Renderer.cpp:40; declarations shader/pathtracing.glsl:112-141).  Matrices are column
 * major like glm::value_ptr. */
typedef struct pt_params {
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
    /* Extension with no counterpart in the reference; 0 keeps reference behaviour.
     * accum_mode 0: outputs directLum/indirectLum/albedo hold the running mean
     *   mix(last, cur, 1/currentspp) when currentspp != 1 (pathtracing.glsl:657-670).
     * accum_mode 1: they hold plain sums (cur is added; currentspp == 1 still
     *   overwrites) so that sample slices rendered on several GPUs can be summed. */
    int32_t accum_mode;
    int32_t reserved[7];
} pt_params;

/* Renderer::Renderer, Renderer.cpp:19-27: allocates the seven outputs for W x H. */
int pt_pass_create(pt_ctx *ctx, int screen_w, int screen_h, pt_pass **out);
/* Renderer::~Renderer, Renderer.cpp:56-64 */
int pt_pass_destroy(pt_pass *pass);

/* input slots, Renderer.cpp:32-38 */
enum {
    PT_SLOT_TEXTURE_HANDLES = 0, /* uint64 handles from pt_tex_handle            */
    PT_SLOT_MATERIAL = 1,        /* float stream, 7 or 8 floats per record       */
    PT_SLOT_TRIANGLE = 2,        /* 96-byte triangles                            */
    PT_SLOT_INSTANCE_INFO = 3,   /* 112-byte records                             */
    PT_SLOT_MESH_BVH = 4,        /* 48-byte nodes                                */
    PT_SLOT_SCENE_BVH = 5,       /* 48-byte nodes                                */
    PT_SLOT_LIGHT = 6            /* 32-byte records                              */
};
/* SSBOBuffer::bind_current_shader(index) for the seven scene inputs.  The pass keeps
 * the pointer, not a copy; rebinding or re-uploading a bound buffer is picked up by
 * the next draw (the reference re-uploads the scene every frame, main.cpp:155-159). */
int pt_pass_bind(pt_pass *pass, int slot, pt_buf *buf);
/* RenderPass::bind_texture(name, ...), RenderPass.cpp:123-128.  Names understood:
 * "skybox", "skybox_samplecache" (main.cpp:179-180; 3-channel float), and the four images of
 * the raster pass (main.cpp:181-184, RasterPass.cpp:100-133; float textures of
 * WINDOW_W x WINDOW_H, row 0 = bottom row):
 *   "depthGBufferTexture"          1 channel: distance from the camera along the primary ray
 *                                  (raster_ps.glsl:22), 1e6 where nothing was drawn
 *   "normalGBufferTexture"         3 channels: world-space normal, any length
 *   "uvGBufferTexture"             2 channels
 *   "instanceIndexGBufferTexture"  1 channel: index into the instance buffer (sampled NEAREST,
 *                                  the others LINEAR, as RasterPass sets them up)
 * With pt_params.RasterizaionFor1st set, all four are required and no primary ray is traced:
 * the first path vertex is rebuilt from the images (pathtracing.glsl:606-624); a pixel whose
 * instance index names no instance counts as not covered (an out-of-bounds read in the
 * reference).  With RasterizaionFor1st off the first hit is traced, and -- as in the reference,
 * which samples these two images either way (:615-616) -- when "depthGBufferTexture" and
 * "instanceIndexGBufferTexture" are bound the depth output is the raster depth (:684) and the
 * albedo is looked up through the raster instance index (:641; an index that names no instance
 * falls back to the traced one).  With neither bound (a host without a raster pass) the depth
 * output is the traced distance, 1e6 on a miss, and the albedo that of the traced instance.
 * Pass NULL to unbind a name. */
int pt_pass_bind_texture(pt_pass *pass, const char *name, pt_tex *tex);
int pt_pass_set_params(pt_pass *pass, const pt_params *params);
/* Renderer::draw(), Renderer.cpp:30-54: one sample per pixel for the whole frame. */
int pt_pass_draw(pt_pass *pass);

/* The host loop `for spp in 1..Config::SPP` around Renderer::draw (main.cpp:169-208: per sample a new jitter and
 * currentspp, then one draw) as ONE call: n samples of the frame (1..16), params[k] the parameter set of sample k.
 * The sets may differ in jitter, currentspp and frameCounter only.  The outputs are, word for word, those of n
 * pt_pass_set_params + pt_pass_draw pairs in that order: the running mean folds the samples in order, motion / depth /
 * normal / instance index are those of the last sample.  What changes is the shape of the work: the n samples share
 * the queues and the launches of one draw, so every launch is n times fatter and the thin late path vertices, whose
 * launches cost the latency of their slowest ray whatever they carry, are paid once per call instead of once per
 * sample.  Path state: about 210 bytes per pixel per sample, allocated at the first call of that size (or by
 * pt_pass_prepare after pt_pass_reserve_batch).  Afterwards the pass's current parameters are params[n - 1]. */
int pt_pass_draw_batch(pt_pass *pass, const pt_params *params, int n);
/* No counterpart in the reference.  Tells pt_pass_prepare how many samples per pt_pass_draw_batch call to size the
 * path state of every draw in flight for (1..16, default 1), so that no draw allocates. */
int pt_pass_reserve_batch(pt_pass *pass, int n);

/* output members of Renderer (Renderer.h:19-25), by the SSBO slot the shader writes
 * (pathtracing.glsl:88-110) */
enum {
    PT_OUT_ALBEDO = 7,          /* W*H*3 */
    PT_OUT_MOTION = 9,          /* W*H*2 */
    PT_OUT_DEPTH = 11,          /* W*H   */
    PT_OUT_NORMAL = 12,         /* W*H*3 */
    PT_OUT_INSTANCE_INDEX = 13, /* W*H   */
    PT_OUT_DIRECT_LUM = 14,     /* W*H*3 */
    PT_OUT_INDIRECT_LUM = 15    /* W*H*3 */
};
/* Borrowed handle, owned by the pass. */
int pt_pass_output(pt_pass *pass, int which, pt_buf **out);

/* Work counters accumulated by draw / intersect since the last reset. */
typedef struct pt_stats {
    uint64_t samples;        /* pixels shaded (one per pixel per draw)            */
    uint64_t rays_extend;    /* closest-hit queries: primary + indirect           */
    uint64_t rays_shadow;    /* occlusion queries: sky NEE + light shadow rays (rays_shadow_null of them not traced) */
    uint64_t draws;
    uint64_t rays_retraced;  /* closest-hit queries the ordered traversal handed to the reference-order one */
    /* How often the pass re-laid out its traversal arrays since it was created: the mesh
     * level follows slots 2 and 4, the scene level slots 3 and 5 (re-uploaded by the reference
     * every frame while instances move); not reset. */
    uint64_t mesh_level_builds, scene_level_builds;
    uint64_t debug[3];       /* step counters of RT_TRACE_STATS builds, else 0 */
    /* 1 when the bound BVH buffers hold a child box that is not inside its parent's box (the
     * reference's builder never makes one, src/BVH.cpp:11-82): the ordered traversal's
     * equivalence to the reference's order rests on that nesting, so such buffers are traced
     * in the reference's order whatever pt_pass_set_traversal says.  Not reset. */
    uint64_t reference_order_forced;
    /* (instance, triangle) pairs in the flattened tree the ordered traversal currently walks (see
     * pt_pass_set_traversal), 0 while it walks the two levels.  Not reset. */
    uint64_t flat_leaves;
    /* Occlusion queries of the reference (counted in rays_shadow) that were decided without a traversal: light
     * samples that are +0 in every component (a direction below the surface of a conductor), which change no output
     * whether they are visible or not (shader/pathtracing.glsl:516-528 and :531-554 trace them first and evaluate
     * afterwards): the sky's at every vertex, those of analytic lights from the second vertex on. */
    uint64_t rays_shadow_null;
    /* Of rays_extend: the rays of vertex 1 that left their 8 x 8-pixel tile in nearly one direction (a polished floor, a
     * pane) and were therefore walked through the tree as one group per tile, like the camera rays before them. */
    uint64_t rays_extend_grouped;
    /* Of rays_shadow: the sky's light samples of vertex 0 that pointed one way from a tile (an importance-sampled sky sends
     * most samples to its brightest texels) and were walked as one group per tile too. */
    uint64_t rays_shadow_grouped;
} pt_stats;
int pt_pass_stats(pt_pass *pass, pt_stats *out, int reset);

/* Traversal strategy; no counterpart in the reference.
 *   0 (default) ordered traversal (nearer child first, reciprocal-direction culling,
 *     batched leaves); rays whose closest hit the reference's order-dependent pruning could
 *     have dropped (a second hit at the same distance, or nearer than the boxes that lead
 *     to the closest one) are traced again in the reference's order, so results are those
 *     of mode 1.  Once the scene buffers (slots 2-5) have stood still for one draw, or at
 *     pt_pass_prepare, a scene of up to 4 Mi (instance, triangle) pairs is also laid out as ONE
 *     tree in world space -- unless an instance is rotated off the coordinate axes (then the
 *     boxes of that tree would be loose and the two levels are faster) -- which the ordered
 *     traversal then walks instead of the scene level
 *     and the mesh level in turn (a ray is moved into an instance's space only where a triangle
 *     is tested, with the reference's arithmetic; the boxes are widened by what that move can
 *     round away, so the candidates, hence the results, stay those of mode 1).  Host work of
 *     about 10 ms per 100 k pairs and 150 bytes of device memory per pair; a scene whose
 *     instances move every frame never pays it.  Scenes with singular or badly conditioned
 *     instance matrices keep the two levels.  pt_stats.flat_leaves tells which tree is walked;
 *   1 the reference's visiting order for every ray (shader/pathtracing.glsl:254-316);
 *   2 as 0, but never the flattened tree;
 *   3 as 0, with the flattened tree built at the next draw / intersect / prepare whatever the
 *     scene's size (up to 128 Mi pairs) and however its instances are rotated. */
int pt_pass_set_traversal(pt_pass *pass, int mode);

/* Node format of the ordered traversal; no counterpart in the reference, results do not depend
 * on it.  The ordered traversal walks the reference's binary trees collapsed to four children
 * per node:  0 = 128-byte nodes with the reference's fp32 boxes,  1 = 64-byte nodes with boxes
 * on a 256-step grid rounded outwards (half the bytes and loads per visit; faster on every
 * scene measured),  -1 (default) = the library's choice, currently 1. */
int pt_pass_set_wide_nodes(pt_pass *pass, int format);

/* Does now what the next draw or intersect would do first: validates the bound scene buffers and
 * (re)builds what changed -- the traversal arrays from slots 2-5 (host work: about a second per
 * 10 M triangles on 16 threads, then an upload), the texture descriptor table from slot 0.
 * Replaces the eager half of ResourceManager::reload_meshes / reload_scene
 * (src/ResourceManager.cpp:95-224), which upload when the scene changes rather than inside
 * the frame: a loader calls this after binding, and the first draw then starts at once.
 * Waits for the draws in flight when something has to be rebuilt. */
int pt_pass_prepare(pt_pass *pass);

/* How many draws may be in flight on the device at once (1..16, default 3); no counterpart in
 * the reference, whose GL driver queues frames by itself.  pt_pass_draw returns as soon as
 * the draw is queued; consecutive draws run on separate HIP streams with path state of their
 * own, so the thin late bounces of one overlap the wide early bounces of the next.  The
 * outputs are the same as with 1: the last-write outputs and the running mean are applied in
 * call order.  Every other call on the context (uploads, downloads, pt_ctx_synchronize,
 * pt_pass_stats ...) waits for the draws issued before it.  Each draw in flight holds about
 * 210 bytes per pixel of path state.  More than three need more hardware queues than HIP's
 * default of 4 per process (GPU_MAX_HW_QUEUES, set before the process first touches the GPU).
 * While profiling is enabled (pt_pass_set_profiling) the
 * per-kernel times of overlapping draws include each other: profile with 1. */
int pt_pass_set_pipeline(pt_pass *pass, int draws_in_flight);

/* The thin tail of a draw; no counterpart in the reference, results do not depend on it.  Late
 * path vertices carry few paths, yet each costs a chain of launches that wait for their slowest
 * ray.  From the first vertex at which fewer than `threshold` paths were alive in the last
 * finished draw (the pass reads that draw's queue lengths without waiting), ONE launch carries
 * the remaining paths through all their remaining vertices.  threshold 0 turns it off; -1
 * (default) is the library's choice.  Used with the ordered traversal and the sky as the only
 * light; other draws run vertex by vertex throughout. */
int pt_pass_set_tail_threshold(pt_pass *pass, int threshold);

/* Grouping of paths inside the shade launches; no counterpart in the reference (whose fragment
 * shader branches per pixel, shader/pathtracing.glsl:569-577, shader/materials/materials.glsl:82-111),
 * results do not depend on it.  Every block of a shade launch first sorts its paths into two groups,
 * so that at most one of its four waves runs the code of both.
 * -1 (default): the library's choice: in a draw that finds no other draw of its pass on the device,
 *     from the second path vertex on, paths that escaped to the sky behind those that hit something
 *     (shade time -10 % on the demo scene, the draw -2.6 %; with draws in flight it does not pay);
 * 0: no grouping;
 * 1: at every vertex, paths that hit a rough dielectric behind the others (on a 10 M-triangle scene
 *    with half the instances dielectric it measured 1 % slower than no grouping: neighbouring pixels
 *    mostly hit one instance). */
int pt_pass_set_material_grouping(pt_pass *pass, int mode);

/* Per-kernel device time, measured with HIP events recorded on the launching stream around
 * every launch while profiling is enabled.  No counterpart in the reference (its only
 * timing is an FPS counter, main.cpp:423-435); used by bench.py for the roofline figure. */
typedef struct pt_profile {
    double extend_ms, shadow_ms, shade_ms, other_ms; /* other = generate, shadow-apply, finalize */
    double intersect_ms;                             /* pt_pass_intersect launches            */
    uint64_t extend_launches, shadow_launches, shade_launches, other_launches, intersect_launches;
    double trace_ms;         /* launches that trace the shadow rays of one vertex together with the
                                extend rays of the next (ordered traversal, no analytic lights)     */
    uint64_t trace_launches;
    double tail_ms;          /* the launch that carries the few paths still alive at a late vertex
                                through all their remaining vertices (shading and tracing)           */
    uint64_t tail_launches;
} pt_profile;
int pt_pass_set_profiling(pt_pass *pass, int enable);
/* Waits for the stream, then reports the time accumulated since the last reset. */
int pt_pass_profile(pt_pass *pass, pt_profile *out, int reset);

/* Closest hit for n rays against the scene bound to the pass: intersect_sceneBVH,
 * shader/pathtracing.glsl:292-316, exposed on its own.  rays: n * 6 floats
 * (origin xyz, direction xyz; the direction is used as given).  hits: n records. */
typedef struct pt_hit {
    int32_t instance; /* -1 on a miss                                            */
    int32_t triangle; /* index into the triangle buffer, -1 on a miss            */
    float t;          /* 1e18 on a miss (INF of shader/basic/math.glsl:1)        */
    float u, v;       /* barycentrics of vertices 1 and 2                        */
} pt_hit;
int pt_pass_intersect(pt_pass *pass, const pt_buf *rays, size_t n, pt_buf *hits);
/* pt_pass_intersect without any allocation.  No counterpart in the reference: it is for callers
 * that cannot allocate device memory where they call the library (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  Same kernels and results as
 * pt_pass_intersect; the counters, the list of rays left to the reference-order traversal and
 * the stack spill area live in a file-scope device array of 28 MiB that every GPU the library
 * runs on holds until the process ends.  Limits, reported as errors: at most 1 048 576 rays
 * per call; trees whose traversal needs more than 48 spill entries per ray are refused
 * (PT_ERR_UNSUPPORTED); the traversal arrays of the bound scene must be current, i.e.
 * pt_pass_intersect or pt_pass_draw ran since the scene buffers last changed (PT_ERR_STATE
 * otherwise: laying them out allocates).  All calls on one GPU share that GPU's array: a call
 * queues itself behind the previous call's launches when those went to another context's stream
 * (an event made when the first context of the device was created; no allocation), so calls from
 * several contexts or host threads on one GPU take turns; they do not run side by side. */
int pt_pass_intersect_noalloc(pt_pass *pass, const pt_buf *rays, size_t n, pt_buf *hits);

/* ---- multi-GPU: sample slicing over RCCL ------------------------------------------------
 * No counterpart in the reference (single GPU, no collective anywhere).  What shards is the
 * per-pixel mean over the samples of a frame (shader/pathtracing.glsl:657-670; the frame loop
 * main.cpp:169-208): every rank (one per GPU) holds the whole scene and draws its slice of the
 * global sample indices 1..SPP -- pt_params.currentspp carries the GLOBAL index, which seeds
 * the sampler -- with accum_mode = 1, after zero-filling directLum, indirectLum and albedo
 * (pt_buf_fill; the draw with currentspp == 1 overwrites instead).  pt_pass_allreduce then
 * leaves the finished frame on every rank: one all-reduce(sum) over the nine accumulated floats
 * per pixel, times 1 / total_spp, and one broadcast of the seven last-write floats per pixel
 * (motion, depth, normal, instance index) from last_rank, the rank that drew sample SPP.
 * Both run on the context's stream behind the draws issued so far; the call does not wait.
 * RCCL (librccl.so.1) is opened at the first pt_comm_* call; without it they return
 * PT_ERR_UNSUPPORTED.
 *
 * One process per GPU: rank 0 calls pt_comm_unique_id and hands the PT_COMM_ID_BYTES bytes to
 * the other ranks by any host channel (MPI, a file, torch.distributed ...); every rank calls
 * pt_comm_create (collective: returns when all ranks have called it).
 * One process driving n GPUs: pt_comm_create_all over n contexts on n different GPUs, then
 * pt_pass_allreduce_all with one pass per context, in rank order (rank k = ctxs[k]). */
typedef struct pt_comm pt_comm;
#define PT_COMM_ID_BYTES 128
int pt_comm_unique_id(void *id_out);
int pt_comm_create(pt_ctx *ctx, const void *id, int rank, int world, pt_comm **out);
int pt_comm_create_all(pt_ctx *const *ctxs, int n, pt_comm **out);
int pt_comm_destroy(pt_comm *comm);
int pt_comm_rank(const pt_comm *comm);
int pt_comm_world(const pt_comm *comm);
int pt_pass_allreduce(pt_pass *pass, pt_comm *comm, uint32_t total_spp, int last_rank);
int pt_pass_allreduce_all(pt_pass *const *passes, pt_comm *const *comms, int n, uint32_t total_spp, int last_rank);

/* ---- post-processing: the consumers of the pass's outputs that produce a viewable image ----
 * All buffers hold floats, 3 per pixel unless stated; npix is taken from the buffer sizes,
 * which must agree.  Each call is queued behind the draws issued before it.
 *
 * pt_post_merge: color = (direct + indirect) * albedo, NaN or Inf in any channel -> 10000 in
 *   all three (shader/postprocessing/SVGF_Merge.glsl:24-52; SVGF_Merge::draw,
 *   src/renderpass/SVGF_Merge.cpp).
 * pt_post_static_blend: running mean of `color` into history_color across frames with 3-sigma
 *   luminance firefly rejection; history_moment holds 2 floats per pixel (mean luminance and
 *   mean squared luminance).  frame_counter is the value the reference's StaticBlender::draw
 *   passes after its ++frameCounter: 1 for the first frame (StaticBlender.glsl:26-61,
 *   StaticBlender.cpp:17-27).
 * pt_post_tonemap: ACES curve then pow(x, 0.45) per channel (ToneMappingGamma.glsl:17-54);
 *   out may be the same buffer as color. */
int pt_post_merge(pt_ctx *ctx, const pt_buf *direct, const pt_buf *indirect, const pt_buf *albedo, pt_buf *color);
int pt_post_static_blend(pt_ctx *ctx, uint32_t frame_counter, const pt_buf *color, pt_buf *history_color,
                         pt_buf *history_moment);
int pt_post_tonemap(pt_ctx *ctx, const pt_buf *color, pt_buf *out);

/* ---- denoising chain: one call per shader dispatch of the reference's SVGF and TAA passes ----
 * width x height is the render target; every buffer holds width*height pixels of the channel
 * count given (c3 = 3 floats per pixel ...).  Outputs must not alias inputs.  The history
 * copies that the reference's pass classes make after the dispatch (SVGFTemporalFilter.cpp:47-51,
 * TAA.cpp:38-40) are plain pt_buf_copy calls and stay with the caller (include/raytracing_amd.hpp
 * has the classes).
 *
 * pt_post_svgf_temporal: SVGF_TemporalFilter.glsl:124-200.  first_frame is the shader's uniform
 *   (SVGFTemporalFilter::draw always passes false, SVGFTemporalFilter.cpp:30-31).
 *   in: color c3, normal c3, instance_index c1, motion c2; history color c3, moment c2,
 *   num_samples c1, normal c3, instance_index c1.  out: color c3, moment c2, num_samples c1.
 * pt_post_svgf_variance: SVGF_VarianceFilter.glsl:30-103.  in: moment c2, normal c3, depth c1,
 *   num_samples c1; out: variance c1.
 * pt_post_svgf_spatial: one a-trous level, SVGF_SpatialFilter.glsl:32-90; step = 1 << level.
 *   in: color c3, normal c3, depth c1, variance c1; out: color c3.
 * pt_post_taa: TAA.glsl:124-190.  in: color c3 (LDR), history color c3, motion c2, normal c3,
 *   instance_index c1, history normal c3, history instance_index c1; out: color c3. */
int pt_post_svgf_temporal(pt_ctx *ctx, int width, int height, int first_frame, const pt_buf *color, const pt_buf *normal,
                          const pt_buf *instance_index, const pt_buf *motion, pt_buf *out_color, pt_buf *out_moment,
                          pt_buf *out_num_samples, const pt_buf *history_color, const pt_buf *history_moment,
                          const pt_buf *history_num_samples, const pt_buf *history_normal,
                          const pt_buf *history_instance_index);
int pt_post_svgf_variance(pt_ctx *ctx, int width, int height, const pt_buf *moment, const pt_buf *normal, const pt_buf *depth,
                          const pt_buf *num_samples, pt_buf *out_variance);
int pt_post_svgf_spatial(pt_ctx *ctx, int width, int height, int step, const pt_buf *color, const pt_buf *normal,
                         const pt_buf *depth, const pt_buf *variance, pt_buf *out_color);
int pt_post_taa(pt_ctx *ctx, int width, int height, const pt_buf *color, const pt_buf *history_color, const pt_buf *motion,
                const pt_buf *normal, const pt_buf *instance_index, const pt_buf *history_normal,
                const pt_buf *history_instance_index, pt_buf *out_color);

/* SSBOBuffer<float>::save_as_image (src/SSBOBuffer.h:41-62): bytes = min(255, int(x * 255)),
 * rows flipped (buffers hold the bottom row first), written as an 8-bit PNG (the reference
 * lets OpenCV pick the format from the file name; only PNG is written here).  channel is 1, 3
 * or 4 and width * height * channel floats must be the buffer's size. */
int pt_buf_save_as_image(const pt_buf *buf, int width, int height, int channel, const char *path);

/* ---- host-side scene helpers (no GPU) ------------------------------------------------ */
/* Mesh::build_meshBVH + ResourceManager::reload_meshes, src/instance/Mesh.cpp:10-22,
 * src/ResourceManager.cpp:95-140.  tris: concatenated 96-byte triangles, tri_count[m]
 * per mesh.  nodes_out must hold sum(2*tri_count[m]-1) 48-byte nodes.  root_box_out:
 * 6 floats per mesh (min xyz, max xyz).  siz_out/depth_out: per mesh tree. */
int pt_host_build_mesh_bvh(const void *tris, const int32_t *tri_count, int32_t n_mesh, int32_t n_threads,
                           void *nodes_out, float *root_box_out, int32_t *siz_out, int32_t *depth_out);
/* ResourceManager::reload_scene, src/ResourceManager.cpp:200-224 (+ :146-198) for an
 * already walked instance list.  l2w: 16 floats per instance.  instance_info_out:
 * n_inst 112-byte records; scene_nodes_out: 2*n_inst-1 48-byte nodes;
 * stats_out[2] = {siz, depth} of the two-level tree as BVHNode::build books them. */
int pt_host_build_scene(const float *mesh_root_box, const int32_t *mesh_siz, const int32_t *mesh_depth,
                        int32_t n_mesh, const float *l2w, const int32_t *mesh_index, const int32_t *material_ptr,
                        int32_t n_inst, void *instance_info_out, void *scene_nodes_out, int32_t *stats_out);
/* No counterpart in the reference (test and debugging aid; no GPU).  Builds, from
 * reference-format buffers (slots 4, 2, 5, 3), the records the traversal kernels read and
 * checks them independently: every four-wide tree reaches exactly the triangles / instances
 * of its binary tree, every child box as the kernel decodes it contains the reference's box,
 * references stay inside the buffers, the stack bound holds.  format as in
 * pt_pass_set_wide_nodes.  counts_out (optional): wide mesh nodes, wide scene nodes, child
 * boxes checked, box planes that decode to "unbounded".  PT_ERR_INVALID when the buffers are
 * refused (as pt_pass_draw would), PT_ERR_STATE when a check fails. */
int pt_host_check_traversal_layout(const void *mesh_nodes, size_t n_mesh_nodes, const void *tris, size_t n_tris,
                                   const void *scene_nodes, size_t n_scene_nodes, const void *instances, size_t n_inst,
                                   int32_t format, uint64_t counts_out[4]);
/* No counterpart in the reference (test and debugging aid; no GPU).  Builds the flattened tree of
 * pt_pass_set_traversal from reference-format buffers and checks it independently: it reaches every
 * (instance, triangle) pair exactly once, every leaf carries its instance's matrix, its triangle
 * and the box of the triangle's parent in the reference's mesh BVH, every child box as the kernel
 * decodes it contains the world-space image of those parent boxes, the stack bound holds.
 * counts_out: leaves, wide nodes, child boxes checked, planes that decode to "unbounded"; all 0
 * when this scene gets no flattened tree, with the reason in why_not (optional, why_cap bytes).
 * only_axis_aligned: refuse scenes with instances rotated off the axes, as the default traversal
 * mode does.  PT_ERR_INVALID when the buffers are refused, PT_ERR_STATE when a check fails. */
int pt_host_check_flat_layout(const void *mesh_nodes, size_t n_mesh_nodes, const void *tris, size_t n_tris,
                              const void *scene_nodes, size_t n_scene_nodes, const void *instances, size_t n_inst,
                              int32_t only_axis_aligned, uint64_t counts_out[4], char *why_not, size_t why_cap);
/* Transform::matrix, src/Transform.cpp:20-53 */
int pt_host_transform_matrix(const float position[3], const float rotation_deg[3], const float scale[3],
                             float out16[16]);
/* glm mat4 * mat4 / inverse as the reference uses them for the hierarchy walk and
 * InstanceInfo (ResourceManager.cpp:75,154) */
int pt_host_mat4_mul(const float a[16], const float b[16], float out16[16]);
int pt_host_mat4_inverse(const float m[16], float out16[16]);
/* Camera::projection, src/instance/Camera.cpp:21-43 */
int pt_host_camera_projection(float fov_x, float far_plane, int32_t screen_w, int32_t screen_h, float out16[16]);
/* The jitters of one frame as update() draws them, main.cpp:56 and :169-172: std::mt19937 globalRand(0), per sample two
 * draws divided by max() in float (x first: the order of the two draws inside the vec2 constructor is unspecified in
 * C++; this is the choice made here), the last sample of a frame at the pixel centre unless Config::useStaticBlender.
 * frame: how many frames of spp_per_frame samples have been drawn before (the generator runs on from frame to frame).
 * xy: 2 * spp_per_frame floats. */
int pt_host_frame_jitters(uint64_t frame, int32_t spp_per_frame, int32_t static_blender, float *xy);
/* HDRLoader::load, include/hdrloader.cpp:29-116, from memory.  Call with rgb_out NULL
 * to get the size. */
int pt_host_rgbe_decode(const void *bytes, size_t n, int32_t *w, int32_t *h, float *rgb_out);
/* Writes a flat Radiance picture; returns the byte count (call with out NULL first). */
int64_t pt_host_rgbe_encode(const float *rgb, int32_t w, int32_t h, void *out, size_t cap);
/* Skybox::load_to_gpu's CPU half, src/texture/Skybox.cpp:12-53 */
int pt_host_sky_cache(const float *rgb, int32_t w, int32_t h, float *cache_out, double *light_sum_out);

/* ---- glTF binary (.glb) scenes -------------------------------------------------------------
 * The arrays AssimpLoader::load_model (src/tool/loader.cpp:137-158) hands to the reference's
 * scene classes, read straight from the file (the importer library is not part of this
 * build):  one mesh per glTF mesh, triangles in index order as 96-byte records, positions /
 * normals / uvs as stored (processMesh, loader.cpp:75-104; no TEXCOORD_0 -> zeros);  every
 * material a RoughConductor with albedo = baseColorFactor, roughness = max(0.001,
 * roughnessFactor), metallic = 1 (processMaterial, loader.cpp:44-73), one texture per material
 * that names a baseColorTexture;  instances = nodes that carry a mesh, in hierarchy pre-order
 * with accumulated local-to-world matrices, the root node's own transform reset to identity
 * (processNode + loader.cpp:152-156; order of ResourceManager::update_globalinstance_recursive,
 * src/ResourceManager.cpp:47-77).  What the importer would do beyond that (generated normals,
 * quaternion node rotations, several primitives per mesh, several root nodes) is refused with
 * PT_ERR_INVALID and a message.  No GPU. */
typedef struct pt_glb pt_glb;
typedef struct pt_glb_info {
    int32_t n_meshes, n_materials, n_textures, n_instances;
    int64_t n_triangles;
} pt_glb_info;
typedef struct pt_glb_material {
    float albedo[3];
    float roughness, metallic;
    int32_t texture; /* index for pt_host_glb_texture, -1 = none */
} pt_glb_material;
int pt_host_glb_open(const char *path, pt_glb **out);
int pt_host_glb_open_memory(const void *bytes, size_t n, pt_glb **out);
int pt_host_glb_close(pt_glb *glb);
int pt_host_glb_info(const pt_glb *glb, pt_glb_info *out);
/* tri_count_out: n_meshes counts; triangles_out: n_triangles 96-byte records, meshes back to
 * back (the layout pt_host_build_mesh_bvh takes).  Either may be NULL. */
int pt_host_glb_meshes(const pt_glb *glb, int32_t *tri_count_out, void *triangles_out);
int pt_host_glb_materials(const pt_glb *glb, pt_glb_material *out);
/* The ENCODED image (JPEG, PNG ...) of texture `index`, as embedded in the file; the pointer
 * stays valid until pt_host_glb_close.  Decoding is the host's (the reference uses stb_image,
 * loader.cpp:23-27, rows top to bottom, not flipped).  image_index_out: the glTF image. */
int pt_host_glb_texture(const pt_glb *glb, int32_t index, const void **bytes_out, size_t *n_out, int32_t *image_index_out);
/* Per instance: mesh index, material index (-1 = none) and local-to-world matrix (16 floats),
 * multiplied from the left by root_l2w when that is given (the transform the application puts
 * on the loaded root, main.cpp:382).  Outputs may be NULL. */
int pt_host_glb_instances(const pt_glb *glb, const float root_l2w[16], int32_t *mesh_index_out, int32_t *material_index_out,
                          float *l2w_out);
/* The material float stream with one record per INSTANCE (ResourceManager::reload_instances,
 * src/ResourceManager.cpp:146-165; RoughConductor::insert_buffer, src/material/
 * RoughConductor.cpp:26-38) and each instance's materialPtr.  Returns the number of floats
 * (call with stream_out NULL first), negative on error. */
int64_t pt_host_glb_material_stream(const pt_glb *glb, float *stream_out, size_t cap_floats, int32_t *material_ptr_out);

/* The host half of save_as_image: quantise + flip `data` (height rows of width*channel floats,
 * bottom row first) into `pixels_out` (height*width*channel bytes, top row first), and write
 * top-row-first bytes as a PNG. */
int pt_host_quantize_image(const float *data, int32_t width, int32_t height, int32_t channel, uint8_t *pixels_out);
int pt_host_write_png(const char *path, int32_t width, int32_t height, int32_t channel, const uint8_t *pixels);

#ifdef __cplusplus
}
#endif
#endif /* RAYTRACING_AMD_H */
