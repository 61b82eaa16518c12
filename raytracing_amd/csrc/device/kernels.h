// Host-callable launchers of the HIP kernels (defined in the .hip files of this directory).
#pragma once
#include <hip/hip_runtime.h>
#include "../../../include/raytracing_amd.h"
#include "../host/scene_types.h"
#include "device_types.h"
#include "trace_core.h"

namespace rta {

// Path state is kept per slot, and slots are numbered tile by tile (STATE_TILE_W x STATE_TILE_H
// pixels, 64 slots = one wavefront): the rays of one wavefront start from a square patch of the
// image rather than a 64-pixel strip, so they visit more of the same nodes.  The outputs stay in
// row-major pixel order.  Slots of a tile that hang over the right or bottom edge name no pixel.
#ifndef STATE_TILE_W
#define STATE_TILE_W 8
#endif
constexpr int STATE_TILE_H = 64 / STATE_TILE_W;
static_assert(STATE_TILE_W * STATE_TILE_H == 64, "a tile is one wavefront");

// A draw carries `batch` samples of the frame (the reference's host loop `for spp in 1..SPP`, main.cpp:169-208, issues
// one draw per sample): sample k owns the path-state slots [k * npix1, (k + 1) * npix1), queues and launches are
// shared, so every launch of the draw is `batch` times fatter.  What differs between the samples of a frame:
struct SampleParams {
    float jitter[2];        // pt_params.jitter of that sample
    unsigned currentspp;    // pt_params.currentspp: weight of the running mean
    unsigned seed_term;     // frameCounter * currentspp * 26699, the sample's part of the pixel seed (pathtracing.glsl:595-598)
};
constexpr int MAX_BATCH = 16;

// Everything one draw needs, passed to the kernels by value.
struct DrawParams {
    pt_params p;  // (jitter, currentspp, frameCounter of the draw's samples are in `samples`)
    float tan_half_fov;  // tan(fov / 2), evaluated once on the host
    int W, H;
    int npix;     // path-state slots of the draw: batch * npix1
    int npix1;    // path-state slots of one sample: whole tiles, so >= W * H (see state_pixel)
    int batch;    // samples in this draw, 1..MAX_BATCH
    SampleParams samples[MAX_BATCH];
    int tiles_x;  // tiles per row
    int n_shadow;  // shadow entries per pixel = 1 (sky) + lightCount
    // scene
    const float *materials;
    int n_material_floats;
    const TriangleRec *tris;
    const InstAccel *inst;
    const LightInfo *lights;
    const TexDesc *textures;
    int n_textures;
    TexDesc sky, sky_cache;
    // RasterizaionFor1st: the raster pass's images (distance, world normal, uv, instance index)
    TexDesc gb_depth, gb_normal, gb_uv, gb_inst;
    // RasterizaionFor1st off, but the depth and instance-index images are bound: the depth output and the albedo
    // lookup of vertex 0 go through them, as in the reference (gb_depth and gb_inst are valid then)
    int raster_outputs;
    int n_instances;
    // path state, one slot per pixel
    PathRay *rays;
    PathHit *hits;
    PathThroughput *thr;
    PathAccum *acc;
    ShadowEntry *shadow;  // [n_shadow][npix]
    // outputs
    float *o_direct, *o_indirect, *o_albedo, *o_motion, *o_depth, *o_normal, *o_inst;
    // counters: [0] samples, [1] extend rays, [2] shadow rays, [3] draws, [4] extend rays re-traced in reference order,
    // [5] light samples that needed no occlusion query, [6] extend rays of vertex 1 walked as groups, [7] occlusion
    // queries of vertex 0 walked as groups
    unsigned long long *stats;
    int redo_counter_base;  // index in `counters` of the per-vertex re-trace counts, -1 when there are none
    // per-draw queue counters, zeroed before generate: [d] = paths alive at vertex d
    unsigned *counters;
    // Where finalize leaves a copy of the queue lengths for the host (pinned host memory as the
    // device sees it; see QueueSnapshot), or null.
    unsigned *snapshot;
    int tail_from;  // vertex from which the tail launch took over (MAX_DEPTH + 1: it did not run)
};

// Queue lengths of the last finished draw, written by its finalize launch into pinned host memory:
// the host reads them (without waiting for anything) to decide from which vertex the next draw
// hands its paths to the tail launch.  seq_begin is written first and seq_end last, so a reader
// that finds them equal has a consistent copy.  Only a heuristic reads this: outputs do not depend on it.
struct QueueSnapshot {
    unsigned seq_begin;
    unsigned max_depth, tail_from, npix;
    unsigned alive[66];  // [d] = paths alive at vertex d, valid for d <= min(tail_from, max_depth)
    unsigned seq_end;
};

struct FastScene;

// closest hit for n rays given as 6 floats each; hits as pt_hit records.  With redo != nullptr
// only the rays listed there (count read from *redo_count) are traced.
void launch_intersect(hipStream_t s, const TraceScene &sc, int blocks, const float *rays, unsigned n,
                      const unsigned *redo, const unsigned *redo_count, pt_hit *hits, unsigned *next_counter,
                      unsigned long long *retraced_stat);
// ordered form: undecided rays are appended to redo instead of being written
void launch_intersect_fast(hipStream_t s, const FastScene &sc, int blocks, const float *rays, unsigned n,
                           pt_hit *hits, unsigned *next_counter, unsigned *redo, unsigned *redo_count);
// extend stage: rays[queue[q]] -> hits[queue[q]] for q < *count (queue == nullptr: q itself)
void launch_extend(hipStream_t s, const TraceScene &sc, int blocks, const unsigned *queue, const unsigned *count,
                   const PathRay *rays, PathHit *hits, unsigned *next_counter, bool second_pass);
void launch_extend_fast(hipStream_t s, const FastScene &sc, int blocks, const unsigned *queue, const unsigned *count,
                        const PathRay *rays, PathHit *hits, unsigned *next_counter, unsigned *redo,
                        unsigned *redo_count);
// shadow stage: ray (q, k) for q < *count, k < n_shadow; marks visible entries
void launch_shadow(hipStream_t s, const TraceScene &sc, int blocks, const unsigned *queue, const unsigned *count,
                   const PathRay *rays, ShadowEntry *shadow, int n_shadow, int npix, unsigned *next_counter);
// the same through the ordered traversal; work items it leaves undecided are listed in redo[0 .. *redo_count) and
// traced in the reference's order by launch_shadow_redo
void launch_shadow_fast(hipStream_t s, const FastScene &sc, int blocks, const unsigned *queue, const unsigned *count,
                        const PathRay *rays, ShadowEntry *shadow, int n_shadow, int npix, unsigned *next_counter, unsigned *redo,
                        unsigned *redo_count);
void launch_shadow_redo(hipStream_t s, const TraceScene &sc, int blocks, const unsigned *queue, const PathRay *rays,
                        ShadowEntry *shadow, int n_shadow, int npix, const unsigned *redo, const unsigned *redo_count,
                        unsigned *next_counter);
// reference-order pass over what launch_path_trace_fast left undecided: extend rays (entries are path slots) and,
// tagged with bit 31, sky occlusion queries, whose light sample is applied where the query ends
void launch_path_redo(hipStream_t s, const TraceScene &sc, int blocks, const unsigned *queue, const unsigned *count,
                      const PathRay *rays, PathHit *hits, const ShadowEntry *shadow, const PathThroughput *thr, PathAccum *acc,
                      unsigned *next_counter);

void launch_path_trace_fast(hipStream_t s, const FastScene &sc, int blocks, const unsigned *queue_e, const unsigned *count_e,
                            const unsigned *queue_s, const unsigned *count_s, const PathRay *rays, PathHit *hits,
                            ShadowEntry *shadow, int npix, unsigned *next_counter, unsigned *redo, unsigned *redo_count,
                            const PathThroughput *thr, PathAccum *acc, unsigned long long *shadow_rays_stat);
// One launch for every vertex from dep0 on: each wave carries its paths through shade / trace /
// shade ... to their ends (tail_kernels.hip).  queue_in / count_in: the paths alive at vertex dep0,
// with their hits in place; next_batch: a zeroed counter.
void launch_path_tail(hipStream_t s, const FastScene &fsc, const TraceScene &tsc, const DrawParams &dp, int dep0, int blocks,
                      const unsigned *queue_in, const unsigned *count_in, unsigned *next_batch);
int tail_paths_per_block();
void launch_generate(hipStream_t s, const DrawParams &dp);
// generate + the ordered extend of vertex 0 in one launch (camera rays made where they are traced)
// packets: over the flattened tree, one wave per tile of 64 slots walking one node sequence (trace_packet.h); the tree's
// deepest walk must fit packet_stack_entries() deferred nodes
// base, count: the slots of the draw this launch covers (multiples of 64; the whole draw: 0, dp.npix); books: it keeps the
// draw's books (one launch of a draw does)
void launch_primary_fast(hipStream_t s, const FastScene &sc, int blocks, const DrawParams &dp, unsigned *next_counter, unsigned *redo,
                         unsigned *redo_count, bool packets, unsigned base, unsigned count, bool books);
int packet_stack_entries();
// the packet form of launch_primary_fast and launch_shade for vertex 0 in one kernel (plain traced first hits only: no
// raster images in play): a wave shades the pixels of a tile where its walk ends.  Slots it cannot decide are stored and
// listed in redo as launch_primary_fast does; they want launch_extend(second_pass) and then launch_shade over that list
// with SHADE_LISTED.
// packet_queue (may be null): tiles whose paths go on in nearly one direction are listed there as groups of 64 entries
// (0xFFFFFFFF: no path) instead of in queue_out; *packet_count counts the groups, *packet_valid their paths.  Such groups
// want launch_extend_packets, and launch_shade with SHADE_GROUPS for the vertex they reach.
void launch_primary_shade_packets(hipStream_t s, const FastScene &sc, int blocks, const DrawParams &dp, unsigned *next_counter,
                                  unsigned *redo, unsigned *redo_count, unsigned *queue_out, unsigned *count_out,
                                  unsigned *shadow_queue, unsigned *shadow_count, unsigned *packet_queue, unsigned *packet_count,
                                  unsigned *packet_valid, unsigned *sgroup_queue, unsigned *sgroup_count, unsigned *sgroup_valid);
// (sgroup_*: the same for the sky's light samples of a tile that point one way -- instead of in shadow_queue; such groups
// want launch_shadow_packets)
void launch_shadow_packets(hipStream_t s, const FastScene &sc, int blocks, const unsigned *queue, const unsigned *group_count,
                           const unsigned *valid_count, const PathRay *rays, const ShadowEntry *shadow, const PathThroughput *thr,
                           PathAccum *acc, unsigned *next_counter, unsigned *redo, unsigned *redo_count, unsigned long long *stats);
// extend rays of those groups, one wave per group (hits and the list of undecided slots as launch_extend_fast)
void launch_extend_packets(hipStream_t s, const FastScene &sc, int blocks, const unsigned *queue, const unsigned *group_count,
                           const PathRay *rays, PathHit *hits, unsigned *next_counter, unsigned *redo, unsigned *redo_count);
// shade stage for path vertex `dep` (0 = primary hit).  Appends continuing paths to
// queue_out and paths with pending light samples to shadow_queue.
// group: every block first sorts its 256 paths into two groups, so that at most one of its four waves runs the code
// of both: paths that escaped (a sky lookup, shader/pathtracing.glsl:569-577) behind those that hit something, or,
// with group_materials, dielectrics behind conductors (the reference dispatches per fragment,
// shader/materials/materials.glsl:82-111).
// what: SHADE_ALL, the paths of the vertex (queue_in, *count_in);
//   SHADE_LISTED, the few slots of vertex 0 that launch_primary_shade_packets left to the reference-order pass (queue_in is
//   that list; the rays of the vertex have been counted by that launch);
//   SHADE_GROUPS, the paths of the groups in a packet queue (queue_in holds *count_in groups of 64 entries, *valid_count
//   paths in all), behind the SHADE_ALL launch of the same vertex: their rays are counted, the re-traced ones are not
//   counted a second time, and the vertex's count of live paths (dp.counters[dep]) is made whole.
enum ShadeWhat { SHADE_ALL = 0, SHADE_LISTED = 1, SHADE_GROUPS = 2 };
void launch_shade(hipStream_t s, const DrawParams &dp, int dep, const unsigned *queue_in, const unsigned *count_in,
                  unsigned *queue_out, unsigned *count_out, unsigned *shadow_queue, unsigned *shadow_count, bool group, bool group_materials,
                  ShadeWhat what = SHADE_ALL, const unsigned *valid_count = nullptr);
void launch_shadow_apply(hipStream_t s, const DrawParams &dp, const unsigned *shadow_queue,
                         const unsigned *shadow_count);
void launch_finalize(hipStream_t s, const DrawParams &dp);

// post-processing (post_kernels.hip)
void launch_post_merge(hipStream_t s, int npix, const float *di, const float *idi, const float *albedo, float *color);
void launch_post_tonemap(hipStream_t s, size_t n, const float *in, float *out);
void launch_post_static_blend(hipStream_t s, int npix, unsigned frame, const float *color, float *hist_color, float *hist_moment);

// denoising chain (denoise_kernels.hip)
void launch_svgf_temporal(hipStream_t s, int W, int H, int first_frame, const float *color, const float *normal, const float *inst,
                          const float *motion, float *o_color, float *o_moment, float *o_num, const float *h_color,
                          const float *h_moment, const float *h_num, const float *h_normal, const float *h_inst);
void launch_svgf_variance(hipStream_t s, int W, int H, const float *moment, const float *normal, const float *depth,
                          const float *num, float *variance);
void launch_svgf_spatial(hipStream_t s, int W, int H, int step, const float *color, const float *normal, const float *depth,
                         const float *variance, float *out);
void launch_taa(hipStream_t s, int W, int H, const float *color, const float *h_color, const float *motion, const float *normal,
                const float *inst, const float *h_normal, const float *h_inst, float *out);

}  // namespace rta
