// Device half of the C ABI: context, buffers, textures and the path-tracing pass.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
#include "capi_internal.h"
#include "device/kernels.h"
#include "device/trace_fast.h"
#include "host/accel_layout.h"
#include "host/png_writer.h"
#include "host/scene_types.h"

using namespace rta;

namespace {

constexpr int MAX_PATH_DEPTH = 64;
constexpr int CNT_QUEUE = 0;       // [0, 70): paths alive per vertex
constexpr int CNT_SHADOW = 70;     // [70, 140): paths with pending light samples per vertex
constexpr int CNT_NEXT = 140;      // [140, 280): next-ray counters, two per vertex
constexpr int CNT_REDO = 280;       // [280, 350): rays the ordered extend handed to the reference-order extend
constexpr int CNT_NEXT_REDO = 350;  // [350, 420): next-ray counters of those launches
constexpr int CNT_INTERSECT = 420;  // next-ray, redo count, next-ray of the redo launch
constexpr int CNT_TAIL = 424;       // batch counter of the tail launch
constexpr int CNT_TUNE = 425;       // next-ray counter of the second primary launch of a draw that times both kinds
constexpr int CNT_GROUPS = 426;     // groups in the packet queue of vertex 1, the paths in them (427), next-group counter of their launch (428)
constexpr int CNT_SGROUPS = 429;    // the same three for the groups of occlusion queries of vertex 0
constexpr int CNT_SHADOW_REDO = 432;       // [432, 502): occlusion queries the ordered traversal left undecided, per vertex
constexpr int CNT_NEXT_SHADOW_REDO = 502;  // [502, 572): next-ray counters of the launches that trace those
constexpr int CNT_TOTAL = 572;
constexpr size_t GROUPS_FROM_SLOTS = 6000000;  // path slots from which a draw walks coherent bounce rays as groups (three samples at 1920x1080: measured even at two, 2 % faster at four)

struct DeviceArray {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if(need <= bytes && p) return PT_OK;
        if(p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, need ? need : 16);
        if(e != hipSuccess) return fail(PT_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
        bytes = need ? need : 16;
        return PT_OK;
    }
    void release() {
        if(p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

const int OUT_SLOTS[7] = {PT_OUT_ALBEDO, PT_OUT_MOTION, PT_OUT_DEPTH, PT_OUT_NORMAL,
                          PT_OUT_INSTANCE_INDEX, PT_OUT_DIRECT_LUM, PT_OUT_INDIRECT_LUM};
const int OUT_CHANNELS[7] = {3, 2, 1, 3, 1, 3, 3};

}  // namespace

// ---- scratch of pt_pass_intersect_noalloc ------------------------------------------------
// One array per GPU the library runs on, held until the process ends:
//   [0, 256)                    the three counter words of a call (next ray, redo count, next redo ray)
//   [256, 256 + 4 * MAX_RAYS)   ray indices the ordered traversal leaves to the reference-order one
//   then                        stack spill area, [depth][thread] entries of 8 bytes
// `scratch` is the host-side stand-in of the symbol, not a device address: host code only ever
// uses the address hipGetSymbolAddress returned for the current device (g_scratch_ptr).
constexpr size_t NOALLOC_MAX_RAYS = 1u << 20;
constexpr int NOALLOC_BLOCKS = 256;       // persistent grid, 256 threads each (one block per CU)
constexpr int NOALLOC_SPILL_DEPTH = 48;   // spill entries per thread beyond the LDS stack (casa's trees are bounded by 29)
constexpr size_t NOALLOC_REDO_OFFSET = 256;
constexpr size_t NOALLOC_SPILL_OFFSET = NOALLOC_REDO_OFFSET + NOALLOC_MAX_RAYS * sizeof(unsigned);
constexpr size_t NOALLOC_SCRATCH_BYTES =
    NOALLOC_SPILL_OFFSET + (size_t)NOALLOC_BLOCKS * 256 * NOALLOC_SPILL_DEPTH * sizeof(uint2);  // 28 MiB + 256 B
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];
constexpr int NOALLOC_MAX_DEVICES = 64;
static void *g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};
// Every context on a GPU shares that GPU's array, and every context has a stream of its own: a call orders itself
// behind the last call that used the array (an event recorded after its launches, made once per device in
// pt_ctx_create; waiting on it allocates nothing), so calls from several contexts or host threads take turns
// instead of corrupting each other's counters and spill entries.
static std::mutex g_scratch_mutex;
static hipEvent_t g_scratch_event[NOALLOC_MAX_DEVICES] = {};
static hipStream_t g_scratch_last_stream[NOALLOC_MAX_DEVICES] = {};

struct pt_pass {
    pt_ctx *ctx = nullptr;
    int W = 0, H = 0;
    pt_buf *outputs[7] = {};
    // The outputs are views into two allocations, so that the multi-GPU combine moves each group
    // with one collective: the three accumulated ones (direct, indirect, albedo: 9 floats per
    // pixel) are summed over the ranks, the four last-write ones (motion, depth, normal,
    // instance index: 7 floats per pixel) are taken from the rank that drew the last sample.
    void *d_accum_group = nullptr, *d_lastwrite_group = nullptr;
    pt_buf *slots[7] = {};
    uint64_t seen_version[7] = {};
    const pt_buf *seen_ptr[7] = {};
    pt_tex *sky = nullptr, *sky_cache = nullptr;
    pt_tex *gb_depth = nullptr, *gb_normal = nullptr, *gb_uv = nullptr, *gb_inst = nullptr;  // RasterizaionFor1st inputs
    pt_params params;
    bool have_params = false;
    // derived scene
    // Two levels, rebuilt separately: the mesh level follows the triangle and mesh-BVH slots,
    // the (small) scene level the instance and scene-BVH slots, which the reference re-uploads
    // every frame while instances move (Config::DynamicBVH, main.cpp).
    bool mesh_level_valid = false, scene_level_valid = false;
    AccelMeshLevel mesh_level;    // device arrays are cleared after upload; the two lookup tables stay
    AccelSceneLevel scene_level;  // header + max_stack (vectors are cleared after upload)
    uint64_t mesh_level_builds = 0, scene_level_builds = 0;
    DeviceArray d_mesh_buf, d_scene_buf, d_spill, d_tex_table;  // the two traversal buffers (device_types.h)
    DeviceArray d_redo_isect;  // redo list of the ordered intersect
    int traversal_mode = 0;               // pt_pass_set_traversal
    // The flattened tree (trace_flat.h): built once the scene has stood still for a draw (or at once: mode 3,
    // pt_pass_prepare), dropped when slots 2-5 change.  flat_refused: this scene gets none (flat_reason says why).
    bool flat_valid = false, flat_refused = false, flat_forced = false;  // flat_forced: built under mode 3
    int flat_unchanged = 0;
    AccelFlat flat;  // (slots cleared after upload)
    DeviceArray d_flat_buf;
    std::string flat_reason;
    uint64_t flat_builds = 0;
    int quantise = -1;                    // wide nodes: 0 WNodes, 1 or -1 (default) QNodes
    int n_textures = 0;
    int spill_depth = 0;
    size_t n_material_floats = 0;
    // Path state.  Consecutive draws are independent until they touch the outputs, so each
    // draw runs on one of `n_lanes` streams with path state of its own: the narrow late
    // bounces of one draw overlap the wide early bounces of the next.  Events keep the two
    // places where draws do meet in call order: the last-write outputs written while shading
    // the first vertex, and the running mean in finalize.
    struct Lane {
        hipStream_t stream = nullptr;
        DeviceArray rays, hits, thr, acc, shadow, queue_a, queue_b, queue_s, queue_r, queue_sr, queue_p, queue_ps, counters, spill;
        int shadow_capacity = 0;
        int batch_capacity = 0;  // samples per draw the path state is sized for
        int spill_depth = 0;
        hipEvent_t done = nullptr, gbuf = nullptr, fin = nullptr, start = nullptr;
    };
    static constexpr int MAX_LANES = 16;
    Lane lanes[MAX_LANES];
    int n_lanes = 3;
    int next_lane = 0;
    int batch_reserve = 1;  // pt_pass_reserve_batch: samples per draw that pt_pass_prepare sizes the path state for
    // Camera rays over the flattened tree: one ray per lane (kind 0) or one tile per wave (kind 1, trace_packet.h).  Which
    // is faster depends on how many triangles a tile of 8 x 8 pixels sees (few: the tile's rays share their walk; many:
    // every lane pays for the union of 64 walks), so the pass times a launch of each kind (events on the draw's
    // stream, read when they have completed, no wait) and keeps the faster one until the flattened tree is rebuilt: on the
    // second draw that carries four samples or more, one half of the samples goes through each kind (the same image, so the
    // two figures compare); draws of fewer samples are timed one kind per draw and compared when their sizes are equal.
    // The two sorts of draw are decided apart ([0]: up to three samples, [1]: more): a launch over one sample's tiles waits
    // for its slowest walk, one over eight samples' fills the chip, and the kinds do not rank the same in both.
    // Results do not depend on the choice.
    struct PrimaryTuner {
        hipEvent_t a[2] = {nullptr, nullptr}, b[2] = {nullptr, nullptr};
        size_t slots[2] = {0, 0};
        bool pending[2] = {false, false};
        bool halves[2] = {false, false};  // the figure is of one half of a draw whose other half went through the other kind
        int draws_seen = 0;               // draws over this tree so far (the first one is not timed: caches and code are cold)
        double ms_per_slot[2] = {0.0, 0.0};
        int decided = -1;
        uint64_t flat_build = 0;  // the flattened tree the measurements belong to
    } primary_tuner[2];
    hipEvent_t last_gbuf = nullptr, last_fin = nullptr;  // of the previous draw; null before the first
    hipStream_t cur_stream = nullptr;                    // stream the ScopedTimer records on
    DeviceArray d_counters, d_stats, d_err;  // d_err: traversal counters of RT_TRACE_STATS builds (bytes 16..47)
    // Stack-overflow flag of the traversal kernels, in pinned host memory the device writes to
    // directly (it is written only when a ray runs out of stack, i.e. never in a healthy run), so
    // that every call that waits for the device can look at it without a copy.
    // Queue lengths of the last finished draw (QueueSnapshot, kernels.h), same kind of memory.
    QueueSnapshot *h_snapshot = nullptr;
    unsigned *d_snapshot = nullptr;
    int tail_threshold = -1;  // pt_pass_set_tail_threshold
    bool mixed_materials = false;  // instances with different material types: the shade launch groups paths by type
    int material_grouping = -1;    // pt_pass_set_material_grouping: -1 the library's choice (a lone draw: escaped paths apart, from vertex 1 on), 0 never, 1 by material type, always
    unsigned *h_err = nullptr;  // host address
    unsigned *d_err_flag = nullptr;  // the same word as the device sees it
    int trace_blocks = 0;   // persistent grid of a stand-alone launch (pt_pass_intersect); also sizes the spill areas
    int draw_blocks = 0;    // persistent grid of the trace launches of a draw
    int redo_blocks = 0;  // grid of the (small) reference-order launches behind the ordered ones
    // optional per-kernel timing
    bool profiling = false;
    struct Timed { int kind; hipEvent_t a, b; };
    std::vector<Timed> timed;
    std::vector<hipEvent_t> event_pool;
    pt_profile prof;
};

int ctx_check_overflow(pt_ctx *ctx) {
    bool hit = false;
    for(pt_pass *p : ctx->passes)
        if(p->h_err && *static_cast<volatile unsigned *>(p->h_err)) {
            *static_cast<volatile unsigned *>(p->h_err) = 0u;
            hit = true;
        }
    if(hit)
        return fail(PT_ERR_STATE, "traversal stack overflow: a ray needed more deferred nodes than the stack holds; "
                                  "results of the calls since the last successful synchronisation are incomplete");
    return PT_OK;
}

namespace {

void unbind_everywhere(pt_ctx *ctx, const pt_buf *buf, const pt_tex *tex) {
    for(pt_pass *p : ctx->passes) {
        if(buf)
            for(int k = 0; k < 7; k++)
                if(p->slots[k] == buf) {
                    p->slots[k] = nullptr;
                    p->seen_ptr[k] = nullptr;
                }
        if(tex) {
            for(pt_tex **t : {&p->sky, &p->sky_cache, &p->gb_depth, &p->gb_normal, &p->gb_uv, &p->gb_inst})
                if(*t == tex) *t = nullptr;
            // the descriptor table built from the handle buffer may name it: rebuild at the next draw,
            // which then reports the dead handle
            p->seen_version[PT_SLOT_TEXTURE_HANDLES] = ~0ull;
        }
    }
}

// The bytes of a buffer on the host: its upload shadow when there is one (no copy), else a
// download into `tmp`.
int host_view(pt_buf *b, std::vector<uint8_t> &tmp, const uint8_t **out) {
    if(b->shadow_valid) {
        *out = b->shadow.data();
        return PT_OK;
    }
    tmp.resize(b->bytes);
    if(b->bytes) {
        JOIN_TRY(b->ctx);
        HIP_TRY(hipStreamSynchronize(b->ctx->stream));
        HIP_TRY(hipMemcpy(tmp.data(), b->dptr, b->bytes, hipMemcpyDeviceToHost));
    }
    *out = tmp.data();
    return PT_OK;
}

// The upload shadows of the triangle and mesh-BVH buffers are read by the mesh-level re-layout only
// (the scene level works from the root-node copies the mesh level keeps): once the layout stands,
// large ones are given back (0.96 GB each for 10 M triangles); a later mesh-level re-layout of the
// same buffers (other node format, instances rooted inside a tree) reads the device copies instead.
void release_large_shadow(const pt_buf *b) {
    size_t least = (size_t)64 << 20;
    if(const char *e = getenv("RTA_SHADOW_RELEASE_MIN")) least = (size_t)strtoull(e, nullptr, 10);  // (test aid)
    if(b->shadow_valid && b->shadow.size() >= least) {
        // (the upload that reads the shadow was queued on the context's stream and may still be running)
        if(hipStreamSynchronize(b->ctx->stream) != hipSuccess) return;
        b->shadow_valid = false;
        std::vector<uint8_t>().swap(b->shadow);
    }
}

int upload_array(pt_ctx *ctx, DeviceArray &d, const void *src, size_t bytes) {
    // the array may still be read by draws in flight
    JOIN_TRY(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    int rc = d.ensure(bytes);
    if(rc) return rc;
    if(bytes) {
        HIP_TRY(hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return PT_OK;
}

bool slot_changed(pt_pass *p, int s) {
    return p->seen_ptr[s] != p->slots[s] || (p->slots[s] && p->seen_version[s] != p->slots[s]->version);
}
void slot_mark(pt_pass *p, int s) {
    p->seen_ptr[s] = p->slots[s];
    p->seen_version[s] = p->slots[s] ? p->slots[s]->version : 0;
}

// (Re)builds whatever the bound scene buffers changed: traversal layout from slots 2-5,
// texture descriptor table from slot 0, and validates material pointers against slot 1.
int prepare_scene(pt_pass *p, bool eager = false) {
    pt_ctx *ctx = p->ctx;
    for(int s = PT_SLOT_MATERIAL; s <= PT_SLOT_SCENE_BVH; s++)
        if(!p->slots[s] || p->slots[s]->bytes == 0)
            return fail(PT_ERR_STATE, "scene slot " + std::to_string(s) + " is not bound or empty");

    const bool mesh_changed = !p->mesh_level_valid || slot_changed(p, PT_SLOT_TRIANGLE) || slot_changed(p, PT_SLOT_MESH_BVH);
    // (the instance records of the scene level carry each instance's material type, so a new material
    // buffer re-lays out the scene level too: a few KB)
    const bool mat_changed = slot_changed(p, PT_SLOT_MATERIAL);
    const bool scene_changed = mesh_changed || !p->scene_level_valid || slot_changed(p, PT_SLOT_INSTANCE_INFO) ||
                               slot_changed(p, PT_SLOT_SCENE_BVH) || mat_changed;
    if(mesh_changed || !p->scene_level_valid || slot_changed(p, PT_SLOT_INSTANCE_INFO) || slot_changed(p, PT_SLOT_SCENE_BVH)) {
        // (a new material buffer alone leaves the geometry, hence the flattened tree, as it is)
        p->flat_valid = false;
        p->flat_refused = false;
        p->flat_unchanged = 0;
    }
    if(scene_changed || mat_changed) {
        std::vector<uint8_t> t_inst, t_mats, t_tris, t_mesh, t_scene;
        const uint8_t *inst = nullptr, *mats = nullptr, *tris = nullptr, *mesh = nullptr, *scene = nullptr;
        int rc;
        pt_buf *b_inst = p->slots[PT_SLOT_INSTANCE_INFO], *b_mats = p->slots[PT_SLOT_MATERIAL], *b_tris = p->slots[PT_SLOT_TRIANGLE],
               *b_mesh = p->slots[PT_SLOT_MESH_BVH], *b_scene = p->slots[PT_SLOT_SCENE_BVH];
        if((rc = host_view(b_inst, t_inst, &inst))) return rc;
        if((rc = host_view(b_mats, t_mats, &mats))) return rc;
        if(b_inst->bytes % sizeof(InstanceInfo)) return fail(PT_ERR_INVALID, "instance buffer size is not a multiple of 112");
        if(b_mats->bytes % 4) return fail(PT_ERR_INVALID, "material buffer size is not a multiple of 4");
        const size_t n_inst = b_inst->bytes / sizeof(InstanceInfo), n_mat = b_mats->bytes / 4;
        const InstanceInfo *ii = reinterpret_cast<const InstanceInfo *>(inst);
        const float *mf = reinterpret_cast<const float *>(mats);
        for(size_t i = 0; i < n_inst; i++) {
            const int mp = ii[i].materialPtr;
            if(mp < 0 || (size_t)mp + 7 > n_mat) return fail(PT_ERR_INVALID, "instance " + std::to_string(i) + ": material pointer outside the material buffer");
            if((int)(mf[mp] + 0.5f) == 2 && (size_t)mp + 8 > n_mat) return fail(PT_ERR_INVALID, "instance " + std::to_string(i) + ": dielectric record is truncated");
        }
        p->n_material_floats = n_mat;
        if(scene_changed) {
            if((rc = host_view(b_scene, t_scene, &scene))) return rc;
            if(b_tris->bytes % sizeof(TriangleRec) || b_mesh->bytes % sizeof(BVHNodeInfo) || b_scene->bytes % sizeof(BVHNodeInfo))
                return fail(PT_ERR_INVALID, "triangle / BVH buffer sizes are not multiples of 96 / 48 bytes");
            // The mesh BVH array is read by the mesh-level re-layout; the scene level needs the instances' root
            // nodes only, which the mesh level keeps copies of: moving instances does not touch the big array.
            const BVHNodeInfo *mesh_n = nullptr;
            const size_t n_mesh_nodes = b_mesh->bytes / sizeof(BVHNodeInfo);
            std::string err;
            if(mesh_changed) {
                if((rc = host_view(b_mesh, t_mesh, &mesh))) return rc;
                mesh_n = reinterpret_cast<const BVHNodeInfo *>(mesh);
                if((rc = host_view(b_tris, t_tris, &tris))) return rc;
                p->mesh_level_valid = false;
                p->scene_level_valid = false;
                AccelMeshLevel &ml = p->mesh_level;
                if(!build_accel_mesh_level(mesh_n, n_mesh_nodes, reinterpret_cast<const TriangleRec *>(tris), b_tris->bytes / sizeof(TriangleRec), {}, p->quantise, ml, err))
                    return fail(PT_ERR_INVALID, "scene buffers rejected: " + err);
                if((rc = upload_array(ctx, p->d_mesh_buf, ml.slots.data(), ml.slots.size() * sizeof(TNode)))) return rc;
                std::vector<TNode>().swap(ml.slots);
                p->mesh_level_valid = true;
                p->mesh_level_builds++;
                slot_mark(p, PT_SLOT_TRIANGLE);
                slot_mark(p, PT_SLOT_MESH_BVH);
            }
            AccelSceneLevel sl;
            bool missing_root = false;
            bool ok = build_accel_scene_level(mesh_n, n_mesh_nodes, p->mesh_level, reinterpret_cast<const BVHNodeInfo *>(scene),
                                              b_scene->bytes / sizeof(BVHNodeInfo), ii, n_inst, sl, err, &missing_root);
            if(!ok && missing_root) {
                // an instance starts inside a tree: give the mesh level wide trees for those nodes too
                std::vector<int32_t> roots;
                for(size_t i = 0; i < n_inst; i++) roots.push_back(ii[i].meshIndex);
                if(!mesh_n) {
                    if((rc = host_view(b_mesh, t_mesh, &mesh))) return rc;
                    mesh_n = reinterpret_cast<const BVHNodeInfo *>(mesh);
                }
                if((rc = host_view(b_tris, t_tris, &tris))) return rc;
                p->mesh_level_valid = false;
                AccelMeshLevel &ml = p->mesh_level;
                if(!build_accel_mesh_level(mesh_n, n_mesh_nodes, reinterpret_cast<const TriangleRec *>(tris), b_tris->bytes / sizeof(TriangleRec), roots, p->quantise, ml, err))
                    return fail(PT_ERR_INVALID, "scene buffers rejected: " + err);
                if((rc = upload_array(ctx, p->d_mesh_buf, ml.slots.data(), ml.slots.size() * sizeof(TNode)))) return rc;
                std::vector<TNode>().swap(ml.slots);
                p->mesh_level_valid = true;
                p->mesh_level_builds++;
                ok = build_accel_scene_level(mesh_n, n_mesh_nodes, p->mesh_level, reinterpret_cast<const BVHNodeInfo *>(scene),
                                             b_scene->bytes / sizeof(BVHNodeInfo), ii, n_inst, sl, err);
            }
            if(!ok) {
                p->scene_level_valid = false;
                return fail(PT_ERR_INVALID, "scene buffers rejected: " + err);
            }
            release_large_shadow(b_tris);
            release_large_shadow(b_mesh);
            {
                // material types, for the shade launch's grouping; mixed_materials: more than one type is in use
                bool seen[3] = {false, false, false};
                for(size_t i = 0; i < n_inst && i < sl.instances.size(); i++) {
                    const float tf = mf[ii[i].materialPtr];
                    const int t = (int)(tf - (0.0f < tf ? 0.0f : 1.0f) + 0.5f);  // roundint of shader/basic/math.glsl:6-9
                    sl.instances[i].material_type = (uint32_t)t;
                    seen[t == 1 ? 1 : (t == 2 ? 2 : 0)] = true;
                }
                p->mixed_materials = (int)seen[0] + (int)seen[1] + (int)seen[2] > 1;
            }
            {
                std::vector<char> host(sl.scene_nodes.size() * sizeof(TNode) + sl.instances.size() * sizeof(InstAccel) +
                                       sl.wide_nodes.size() * sizeof(TNode));
                char *w = host.data();
                memcpy(w, sl.scene_nodes.data(), sl.scene_nodes.size() * sizeof(TNode));
                w += sl.scene_nodes.size() * sizeof(TNode);
                memcpy(w, sl.instances.data(), sl.instances.size() * sizeof(InstAccel));
                w += sl.instances.size() * sizeof(InstAccel);
                if(!sl.wide_nodes.empty()) memcpy(w, sl.wide_nodes.data(), sl.wide_nodes.size() * sizeof(TNode));
                if((rc = upload_array(ctx, p->d_scene_buf, host.data(), host.size()))) return rc;
            }
            p->scene_level.header = sl.header;
            p->scene_level.max_stack = sl.max_stack;
            p->scene_level.nested = sl.nested;
            int need = sl.max_stack + 2 - (LDS_STACK < FAST_LDS_STACK ? LDS_STACK : FAST_LDS_STACK);
            if(need < 4) need = 4;
            // (test aid: a cap below what the trees need makes deep rays overflow, which must be reported)
            const char *cap_env = getenv("RTA_SPILL_DEPTH_CAP");
            const bool capped = cap_env && *cap_env && atoi(cap_env) >= 0 && atoi(cap_env) < need;
            if(capped) need = atoi(cap_env);
            if(capped ? need != p->spill_depth || !p->d_spill.p : need > p->spill_depth || !p->d_spill.p) {
                JOIN_TRY(ctx);
                HIP_TRY(hipStreamSynchronize(ctx->stream));
                p->spill_depth = need;
                const size_t nthreads = (size_t)p->trace_blocks * TRACE_BLOCK;
                if((rc = p->d_spill.ensure(nthreads * (size_t)p->spill_depth * sizeof(uint2)))) return rc;
            }
            p->scene_level_valid = true;
            p->scene_level_builds++;
            slot_mark(p, PT_SLOT_SCENE_BVH);
        }
        slot_mark(p, PT_SLOT_INSTANCE_INFO);
        slot_mark(p, PT_SLOT_MATERIAL);
    }
    // ---- the flattened tree
    const bool flat_wanted = (p->traversal_mode == 0 || p->traversal_mode == 3) && p->quantise != 0 && p->mesh_level.nested && p->scene_level.nested;
    if(flat_wanted && !p->flat_valid && !p->flat_refused) {
        if(!eager && p->traversal_mode != 3 && p->flat_unchanged < 1) {
            p->flat_unchanged++;  // (instances that move every frame never get here twice in a row)
        } else {
            std::vector<uint8_t> t_inst, t_tris, t_mesh, t_scene;
            const uint8_t *inst = nullptr, *tris = nullptr, *mesh = nullptr, *scene = nullptr;
            pt_buf *b_inst = p->slots[PT_SLOT_INSTANCE_INFO], *b_tris = p->slots[PT_SLOT_TRIANGLE], *b_mesh = p->slots[PT_SLOT_MESH_BVH],
                   *b_scene = p->slots[PT_SLOT_SCENE_BVH];
            int rc;
            size_t cap = p->traversal_mode == 3 ? (size_t)1 << 27 : (size_t)1 << 22;
            if(const char *e = getenv("RTA_FLAT_MAX_LEAVES")) cap = (size_t)strtoull(e, nullptr, 10);
            // (the triangle count bounds the leaves from below: do not read the big arrays back for nothing)
            if(b_tris->bytes / sizeof(TriangleRec) > cap) {
                p->flat_refused = true;
                p->flat_reason = "more triangles than the flattened tree is built for";
            } else {
                if((rc = host_view(b_inst, t_inst, &inst))) return rc;
                if((rc = host_view(b_tris, t_tris, &tris))) return rc;
                if((rc = host_view(b_mesh, t_mesh, &mesh))) return rc;
                if((rc = host_view(b_scene, t_scene, &scene))) return rc;
                AccelFlat &fl = p->flat;
                std::string why;
                if(build_accel_flat(reinterpret_cast<const BVHNodeInfo *>(mesh), b_mesh->bytes / sizeof(BVHNodeInfo),
                                    reinterpret_cast<const TriangleRec *>(tris), b_tris->bytes / sizeof(TriangleRec),
                                    reinterpret_cast<const BVHNodeInfo *>(scene), b_scene->bytes / sizeof(BVHNodeInfo),
                                    reinterpret_cast<const InstanceInfo *>(inst), b_inst->bytes / sizeof(InstanceInfo), cap, p->traversal_mode != 3, 0, fl, why)) {
                    if((rc = upload_array(ctx, p->d_flat_buf, fl.slots.data(), fl.slots.size() * sizeof(TNode)))) return rc;
                    std::vector<TNode>().swap(fl.slots);
                    const int need = fl.max_stack + 2 - FAST_LDS_STACK;
                    const char *cap_env = getenv("RTA_SPILL_DEPTH_CAP");
                    if(need > p->spill_depth && !(cap_env && *cap_env)) {
                        JOIN_TRY(ctx);
                        HIP_TRY(hipStreamSynchronize(ctx->stream));
                        p->spill_depth = need;
                        const size_t nthreads = (size_t)p->trace_blocks * TRACE_BLOCK;
                        if((rc = p->d_spill.ensure(nthreads * (size_t)p->spill_depth * sizeof(uint2)))) return rc;
                    }
                    p->flat_valid = true;
                    p->flat_forced = p->traversal_mode == 3;
                    p->flat_builds++;
                    p->flat_reason.clear();
                } else {
                    p->flat_refused = true;
                    p->flat_reason = why;
                }
                release_large_shadow(b_tris);
                release_large_shadow(b_mesh);
            }
        }
    }
    if(slot_changed(p, PT_SLOT_TEXTURE_HANDLES) || !p->d_tex_table.p) {
        std::vector<TexDesc> table;
        pt_buf *hb = p->slots[PT_SLOT_TEXTURE_HANDLES];
        if(hb && hb->bytes) {
            std::vector<uint8_t> tmp;
            const uint8_t *raw = nullptr;
            int rc = host_view(hb, tmp, &raw);
            if(rc) return rc;
            if(hb->bytes % 8) return fail(PT_ERR_INVALID, "texture handle buffer size is not a multiple of 8");
            const uint64_t *h = reinterpret_cast<const uint64_t *>(raw);
            for(size_t k = 0; k < hb->bytes / 8; k++) {
                auto it = ctx->textures.find(h[k]);
                if(it == ctx->textures.end()) return fail(PT_ERR_INVALID, "texture handle " + std::to_string(k) + " does not name a live texture of this context");
                TexDesc d;
                d.data = it->second->dptr;
                d.width = it->second->width; d.height = it->second->height;
                d.channels = it->second->channels; d.format = it->second->format;
                table.push_back(d);
            }
        }
        p->n_textures = (int)table.size();
        TexDesc dummy;
        memset(&dummy, 0, sizeof dummy);
        int rc = upload_array(ctx, p->d_tex_table, table.empty() ? &dummy : table.data(),
                              (table.empty() ? 1 : table.size()) * sizeof(TexDesc));
        if(rc) return rc;
        slot_mark(p, PT_SLOT_TEXTURE_HANDLES);
    }
    return PT_OK;
}

// Persistent grid of a draw's trace launches.  A stand-alone launch wants every wave slot
// (8 blocks of 4 waves per CU, two rounds).  With several draws in flight a grid of 3 blocks
// per CU leaves a wave slot per SIMD to the kernels of the other draws, so that a latency-bound
// shade kernel runs beside a traversal kernel, which is bound by vector-ALU issue: measured
// +12 % (casa, 6 draws in flight) and +3 % (10 M triangles) over full grids.
void set_draw_grid(pt_pass *p) {
    const char *e = getenv("RTA_DRAW_BLOCKS_PER_CU");  // (tuning aid)
    // (blocks per CU of a traversal launch that shares the device with other draws.  Measured, casa 1080p / 10 M
    // triangles at 4K, Mrays/s with 1 | 2 | 3 blocks: 3 draws in flight 2064 | 2482 | 2472; 6: 2792 | 2826 | 2780 and
    // 1809 | 1835 | 1876; 10: 2990 | 2932 | 2891 and 1888 | 1908 | 1881; 14: 3014 | 2974 | 2910 -- the more draws
    // share the device, the smaller each one's persistent grid should be)
    const int shared = e && atoi(e) > 0 && atoi(e) <= 8 ? atoi(e) : (p->n_lanes >= 8 ? 1 : 3);
    p->draw_blocks = p->ctx->num_cus * (p->n_lanes > 1 ? shared : 8);
}

// The ordered traversal is equivalent to the reference's order only when child boxes lie
// inside their parents' boxes; other buffers are traced in the reference's order.
bool use_ordered(const pt_pass *p) { return p->traversal_mode != 1 && p->mesh_level.nested && p->scene_level.nested; }

// spill == nullptr: the pass's own spill area (launches on the main stream)
TraceScene make_trace_scene(pt_pass *p, void *spill = nullptr) {
    TraceScene sc;
    sc.mesh_buf = static_cast<const char *>(p->d_mesh_buf.p);
    sc.scene_buf = static_cast<const char *>(p->d_scene_buf.p);
    sc.hdr = p->scene_level.header;
    sc.spill = static_cast<uint2 *>(spill ? spill : p->d_spill.p);
    sc.spill_depth = p->spill_depth;
    sc.error_flag = p->d_err_flag;
    sc.stats_dbg = static_cast<unsigned *>(p->d_err.p);
    return sc;
}

enum { K_EXTEND = 0, K_SHADOW = 1, K_SHADE = 2, K_OTHER = 3, K_INTERSECT = 4, K_TRACE = 5, K_TAIL = 6 };

// Brackets one launch with two events when profiling is on.
struct ScopedTimer {
    pt_pass *p;
    hipEvent_t a = nullptr, b = nullptr;
    int kind;
    static hipEvent_t take(pt_pass *p) {
        if(!p->event_pool.empty()) {
            hipEvent_t e = p->event_pool.back();
            p->event_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    ScopedTimer(pt_pass *pass, int k) : p(pass), kind(k) {
        if(!p->profiling) return;
        a = take(p);
        b = take(p);
        if(a) (void)hipEventRecord(a, p->cur_stream);
    }
    ~ScopedTimer() {
        if(!p->profiling || !a || !b) return;
        (void)hipEventRecord(b, p->cur_stream);
        p->timed.push_back({kind, a, b});
    }
};

// Folds finished event pairs into the profile (the stream must be idle).
void collect_profile(pt_pass *p) {
    for(auto &t : p->timed) {
        float ms = 0.0f;
        if(hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            switch(t.kind) {
                case K_EXTEND: p->prof.extend_ms += ms; p->prof.extend_launches++; break;
                case K_SHADOW: p->prof.shadow_ms += ms; p->prof.shadow_launches++; break;
                case K_SHADE: p->prof.shade_ms += ms; p->prof.shade_launches++; break;
                case K_INTERSECT: p->prof.intersect_ms += ms; p->prof.intersect_launches++; break;
                case K_TRACE: p->prof.trace_ms += ms; p->prof.trace_launches++; break;
                case K_TAIL: p->prof.tail_ms += ms; p->prof.tail_launches++; break;
                default: p->prof.other_ms += ms; p->prof.other_launches++; break;
            }
        }
        p->event_pool.push_back(t.a);
        p->event_pool.push_back(t.b);
    }
    p->timed.clear();
}

FastScene make_fast_scene(pt_pass *p, void *spill = nullptr) {
    FastScene sc;
    sc.mesh_buf = static_cast<const char *>(p->d_mesh_buf.p);
    sc.scene_buf = static_cast<const char *>(p->d_scene_buf.p);
    sc.hdr = p->scene_level.header;
    const bool flat = p->flat_valid && (p->traversal_mode == 0 || p->traversal_mode == 3);
    sc.flat_buf = flat ? static_cast<const char *>(p->d_flat_buf.p) : nullptr;
    sc.flat_root_ref = p->flat.root_ref;
    sc.flat_pad = p->flat.pad;
    sc.flat_reach = p->flat.reach;
    sc.spill = static_cast<uint2 *>(spill ? spill : p->d_spill.p);
    sc.spill_depth = p->spill_depth;
    sc.error_flag = p->d_err_flag;
    sc.stats_dbg = static_cast<unsigned *>(p->d_err.p);
    return sc;
}

TexDesc desc_of(const pt_tex *t) {
    TexDesc d;
    d.data = t->dptr;
    d.width = t->width; d.height = t->height; d.channels = t->channels; d.format = t->format;
    return d;
}

void release_lane(pt_pass::Lane &l) {
    if(l.stream) {
        (void)hipStreamSynchronize(l.stream);
        (void)hipStreamDestroy(l.stream);
    }
    for(hipEvent_t *e : {&l.done, &l.gbuf, &l.fin, &l.start})
        if(*e) {
            (void)hipEventDestroy(*e);
            *e = nullptr;
        }
    DeviceArray *arrs[] = {&l.rays, &l.hits, &l.thr, &l.acc, &l.shadow, &l.queue_a, &l.queue_b, &l.queue_s, &l.queue_r, &l.queue_sr, &l.queue_p, &l.queue_ps, &l.counters, &l.spill};
    for(DeviceArray *a : arrs) a->release();
    l = pt_pass::Lane();
}

// Stream, events and path state of one lane, sized for the pass and the scene bound now.
// path-state slots of a w x h image: whole tiles of 64 pixels
size_t state_slots(int w, int h) {
    return (size_t)((w + STATE_TILE_W - 1) / STATE_TILE_W) * (size_t)((h + STATE_TILE_H - 1) / STATE_TILE_H) * 64u;
}

int prepare_lane(pt_pass *p, pt_pass::Lane &l, int n_shadow, int batch) {
    if(!l.stream) {
        HIP_TRY(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        for(hipEvent_t *e : {&l.done, &l.gbuf, &l.fin, &l.start}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    if(batch > l.batch_capacity || n_shadow > l.shadow_capacity) {
        if(batch < l.batch_capacity) batch = l.batch_capacity;
        if(n_shadow < l.shadow_capacity) n_shadow = l.shadow_capacity;
        const size_t npix = state_slots(p->W, p->H) * (size_t)batch;
        HIP_TRY(hipStreamSynchronize(l.stream));
        int rc = PT_OK;
        if(!rc) rc = l.rays.ensure(npix * sizeof(PathRay));
        if(!rc) rc = l.hits.ensure(npix * sizeof(PathHit));
        if(!rc) rc = l.thr.ensure(npix * sizeof(PathThroughput));
        if(!rc) rc = l.acc.ensure(npix * sizeof(PathAccum));
        if(!rc) rc = l.queue_a.ensure(npix * 4);
        if(!rc) rc = l.queue_b.ensure(npix * 4);
        if(!rc) rc = l.queue_s.ensure(npix * 4);
        if(!rc) rc = l.queue_r.ensure(2 * npix * 4);  // undecided extend rays and given-up occlusion queries of one vertex
        if(!rc) rc = l.queue_p.ensure(npix * 4);      // groups of 64 paths that leave a tile in nearly one direction (vertex 1)
        if(!rc) rc = l.queue_ps.ensure(npix * 4);     // groups of a tile's light samples that point one way (vertex 0)
        if(!rc) rc = l.counters.ensure(CNT_TOTAL * 4);
        if(!rc) rc = l.shadow.ensure(npix * n_shadow * sizeof(ShadowEntry));
        // (with analytic lights: the occlusion queries of one vertex that the ordered traversal leaves undecided)
        if(!rc && n_shadow > 1) rc = l.queue_sr.ensure(npix * (size_t)n_shadow * 4);
        if(rc) return rc;
        l.batch_capacity = batch;
        l.shadow_capacity = n_shadow;
    }
    if(l.spill_depth < p->spill_depth || !l.spill.p) {
        HIP_TRY(hipStreamSynchronize(l.stream));
        const size_t nthreads = (size_t)p->trace_blocks * TRACE_BLOCK;
        int rc = l.spill.ensure(nthreads * (size_t)p->spill_depth * sizeof(uint2));
        if(rc) return rc;
        l.spill_depth = p->spill_depth;
    }
    return PT_OK;
}

// From which vertex the draw hands its paths to the tail launch (MAX_DEPTH + 1: not at all),
// given the queue lengths of the last finished draw.  A heuristic over possibly stale numbers:
// whatever it answers, the outputs are the same.
int choose_tail_vertex(const pt_pass *p, const pt_params &P, size_t npix, bool eligible) {
    const int none = P.MAX_DEPTH + 1;
    int threshold = p->tail_threshold;
    if(threshold < 0) {
        const char *e = getenv("RTA_TAIL_THRESHOLD");  // (tuning aid)
        threshold = e && *e ? atoi(e) : 131072;
    }
    if(!eligible || threshold <= 0 || !p->h_snapshot || P.MAX_DEPTH < 3) return none;
    const volatile QueueSnapshot *q = p->h_snapshot;
    QueueSnapshot snap;
    snap.seq_end = q->seq_end;
    snap.max_depth = q->max_depth; snap.tail_from = q->tail_from; snap.npix = q->npix;
    for(int d = 0; d < 66; d++) snap.alive[d] = q->alive[d];
    snap.seq_begin = q->seq_begin;
    if(snap.seq_end == 0u || snap.seq_begin != snap.seq_end) return none;             // nothing finished yet, or caught mid-write
    if((int)snap.max_depth != P.MAX_DEPTH || snap.npix != (unsigned)npix) return none;   // another kind of frame: look first
    const int known = (int)snap.tail_from < P.MAX_DEPTH ? (int)snap.tail_from : P.MAX_DEPTH;  // alive[d] is valid up to here
    for(int d = 1; d <= known && d <= P.MAX_DEPTH - 2; d++)   // (handing over the last two vertices saves nothing)
        if(snap.alive[d] < (unsigned)threshold) return d;
    // nothing below the threshold among the vertices seen: if the last draw's tail started early, look one vertex further
    return known < P.MAX_DEPTH - 2 ? known + 1 : none;
}

}  // namespace

// For the multi-GPU combine (capi_comm.hip): the two output groups of a pass.  Marks the outputs as changed.
int pass_output_groups(pt_pass *pass, pt_ctx **ctx, float **accum, float **lastwrite, size_t *npix) {
    if(!pass) return fail(PT_ERR_INVALID, "null pass");
    *ctx = pass->ctx;
    *accum = static_cast<float *>(pass->d_accum_group);
    *lastwrite = static_cast<float *>(pass->d_lastwrite_group);
    *npix = (size_t)pass->W * pass->H;
    for(int k = 0; k < 7; k++) {
        pass->outputs[k]->version++;
        pass->outputs[k]->shadow_valid = false;
    }
    return PT_OK;
}

extern "C" {

// ------------------------------------------------------------------------------- context
int pt_ctx_create(int device_id, pt_ctx **out) {
    if(!out) return fail(PT_ERR_INVALID, "pt_ctx_create: out is null");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if(device_id < 0 || device_id >= n) return fail(PT_ERR_INVALID, "pt_ctx_create: no such device " + std::to_string(device_id) + " (" + std::to_string(n) + " visible)");
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if(std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(PT_ERR_UNSUPPORTED, std::string("pt_ctx_create: this library is built for gfx950 only, device is ") + prop.gcnArchName);
    // each GPU holds its own copy of the scratch array at its own address: ask once per device, with it current
    if(device_id >= NOALLOC_MAX_DEVICES) return fail(PT_ERR_UNSUPPORTED, "pt_ctx_create: device index beyond " + std::to_string(NOALLOC_MAX_DEVICES - 1));
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if(!g_scratch_ptr[device_id]) {
            void *sp = nullptr;
            HIP_TRY(hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)));
            if(!sp) return fail(PT_ERR_HIP, "pt_ctx_create: hipGetSymbolAddress returned no address for the scratch array");
            HIP_TRY(hipEventCreateWithFlags(&g_scratch_event[device_id], hipEventDisableTiming));
            g_scratch_ptr[device_id] = sp;
        }
    }
    pt_ctx *c = new pt_ctx;
    c->device = device_id;
    c->num_cus = prop.multiProcessorCount;
    c->scratch = g_scratch_ptr[device_id];
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if(e != hipSuccess) {
        delete c;
        return fail(PT_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    *out = c;
    return PT_OK;
}

int pt_ctx_destroy(pt_ctx *ctx) {
    if(!ctx) return PT_OK;
    (void)hipSetDevice(ctx->device);
    (void)ctx_join(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PT_OK;
}

int pt_ctx_synchronize(pt_ctx *ctx) {
    if(!ctx) return fail(PT_ERR_INVALID, "pt_ctx_synchronize: null context");
    HIP_TRY(hipSetDevice(ctx->device));
    JOIN_TRY(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ctx_check_overflow(ctx);
}

// Draws run on streams of their own; handing the main stream out orders it behind every draw
// issued so far, so that work the caller queues on it (events, copies, kernels reading
// pt_buf_device_ptr memory) sees finished frames.
void *pt_ctx_stream(pt_ctx *ctx) {
    if(!ctx) return nullptr;
    if(hipSetDevice(ctx->device) != hipSuccess || ctx_join(ctx) != PT_OK) return nullptr;
    return (void *)ctx->stream;
}

// ------------------------------------------------------------------------------- buffers
int pt_buf_create(pt_ctx *ctx, size_t bytes, const void *host, pt_buf **out) {
    if(!ctx || !out) return fail(PT_ERR_INVALID, "pt_buf_create: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    pt_buf *b = new pt_buf;
    b->ctx = ctx;
    b->bytes = bytes;
    hipError_t e = hipMalloc(&b->dptr, bytes ? bytes : 16);
    if(e != hipSuccess) {
        delete b;
        return fail(PT_ERR_HIP, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    }
    *out = b;
    if(host && bytes) {
        int rc = pt_buf_upload(b, host, bytes);
        if(rc) {
            (void)hipFree(b->dptr);
            delete b;
            *out = nullptr;
            return rc;
        }
    }
    return PT_OK;
}

int pt_buf_release(pt_buf *buf) {
    if(!buf) return PT_OK;
    if(!buf->owned) return fail(PT_ERR_INVALID, "pt_buf_release: this buffer is an output of a pass (a borrowed handle); pt_pass_destroy frees it");
    (void)hipSetDevice(buf->ctx->device);
    (void)ctx_join(buf->ctx);
    (void)hipStreamSynchronize(buf->ctx->stream);
    unbind_everywhere(buf->ctx, buf, nullptr);
    if(buf->dptr) (void)hipFree(buf->dptr);
    delete buf;
    return PT_OK;
}

int pt_buf_upload(pt_buf *buf, const void *host, size_t bytes) {
    if(!buf || (!host && bytes)) return fail(PT_ERR_INVALID, "pt_buf_upload: null argument");
    if(bytes != buf->bytes) return fail(PT_ERR_INVALID, "pt_buf_upload: size " + std::to_string(bytes) + " does not match buffer size " + std::to_string(buf->bytes));
    HIP_TRY(hipSetDevice(buf->ctx->device));
    buf->version++;
    if(!bytes) return PT_OK;
    buf->shadow.assign(static_cast<const uint8_t *>(host), static_cast<const uint8_t *>(host) + bytes);
    buf->shadow_valid = true;
    JOIN_TRY(buf->ctx);
    HIP_TRY(hipMemcpyAsync(buf->dptr, buf->shadow.data(), bytes, hipMemcpyHostToDevice, buf->ctx->stream));
    HIP_TRY(hipStreamSynchronize(buf->ctx->stream));
    return PT_OK;
}

int pt_buf_copy(pt_buf *dst, const pt_buf *src) {
    if(!dst || !src) return fail(PT_ERR_INVALID, "pt_buf_copy: null argument");
    if(dst->bytes != src->bytes) return fail(PT_ERR_INVALID, "pt_buf_copy: sizes differ");
    if(dst->ctx != src->ctx) return fail(PT_ERR_INVALID, "pt_buf_copy: buffers belong to different contexts");
    HIP_TRY(hipSetDevice(dst->ctx->device));
    dst->version++;
    dst->shadow_valid = src->shadow_valid;
    if(src->shadow_valid) dst->shadow = src->shadow;
    JOIN_TRY(dst->ctx);
    if(dst->bytes) HIP_TRY(hipMemcpyAsync(dst->dptr, src->dptr, dst->bytes, hipMemcpyDeviceToDevice, dst->ctx->stream));
    return PT_OK;
}

int pt_buf_fill(pt_buf *buf, const void *elem, size_t elem_bytes) {
    if(!buf || !elem) return fail(PT_ERR_INVALID, "pt_buf_fill: null argument");
    if(elem_bytes == 0 || buf->bytes % elem_bytes) return fail(PT_ERR_INVALID, "pt_buf_fill: buffer size is not a multiple of the element size");
    HIP_TRY(hipSetDevice(buf->ctx->device));
    buf->version++;
    buf->shadow_valid = false;
    buf->shadow.clear();
    if(!buf->bytes) return PT_OK;
    JOIN_TRY(buf->ctx);
    if(elem_bytes == 4) {
        uint32_t v;
        memcpy(&v, elem, 4);
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)buf->dptr, (int)v, buf->bytes / 4, buf->ctx->stream));
        return PT_OK;
    }
    if(elem_bytes == 1) {
        HIP_TRY(hipMemsetAsync(buf->dptr, *static_cast<const uint8_t *>(elem), buf->bytes, buf->ctx->stream));
        return PT_OK;
    }
    // other widths: replicate on the host, as the reference's fill does (SSBOBuffer.h:75)
    std::vector<uint8_t> tmp(buf->bytes);
    for(size_t o = 0; o < buf->bytes; o += elem_bytes) memcpy(tmp.data() + o, elem, elem_bytes);
    HIP_TRY(hipMemcpyAsync(buf->dptr, tmp.data(), buf->bytes, hipMemcpyHostToDevice, buf->ctx->stream));
    HIP_TRY(hipStreamSynchronize(buf->ctx->stream));
    return PT_OK;
}

int pt_buf_download(const pt_buf *buf, void *host, size_t bytes) {
    if(!buf || (!host && bytes)) return fail(PT_ERR_INVALID, "pt_buf_download: null argument");
    if(bytes > buf->bytes) return fail(PT_ERR_INVALID, "pt_buf_download: more bytes requested than the buffer holds");
    HIP_TRY(hipSetDevice(buf->ctx->device));
    JOIN_TRY(buf->ctx);
    HIP_TRY(hipStreamSynchronize(buf->ctx->stream));
    if(bytes) HIP_TRY(hipMemcpy(host, buf->dptr, bytes, hipMemcpyDeviceToHost));
    return ctx_check_overflow(buf->ctx);
}

size_t pt_buf_bytes(const pt_buf *buf) { return buf ? buf->bytes : 0; }
void *pt_buf_device_ptr(const pt_buf *buf) { return buf ? buf->dptr : nullptr; }

// ------------------------------------------------------------------------------- textures
int pt_tex_create(pt_ctx *ctx, int width, int height, int channels, int format, const void *data, pt_tex **out) {
    if(!ctx || !out || !data) return fail(PT_ERR_INVALID, "pt_tex_create: null argument");
    if(width <= 0 || height <= 0 || channels < 1 || channels > 4 || (channels == 2 && format != PT_TEX_F32))
        return fail(PT_ERR_INVALID, "pt_tex_create: width/height must be positive and channels 1, 3 or 4 (2 for float images)");
    if(format != PT_TEX_U8 && format != PT_TEX_F32) return fail(PT_ERR_INVALID, "pt_tex_create: unknown format");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)width * height * channels * (format == PT_TEX_F32 ? 4 : 1);
    pt_tex *t = new pt_tex;
    t->ctx = ctx;
    t->width = width; t->height = height; t->channels = channels; t->format = format;
    // (16 bytes behind the last texel: the kernels read a three-byte texel as one word)
    hipError_t e = hipMalloc(&t->dptr, bytes + 16);
    if(e == hipSuccess) e = hipMemsetAsync(static_cast<char *>(t->dptr) + bytes, 0, 16, ctx->stream);
    if(e == hipSuccess) e = hipMemcpyAsync(t->dptr, data, bytes, hipMemcpyHostToDevice, ctx->stream);
    if(e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if(e != hipSuccess) {
        if(t->dptr) (void)hipFree(t->dptr);
        delete t;
        return fail(PT_ERR_HIP, std::string("pt_tex_create: ") + hipGetErrorString(e));
    }
    t->handle = ctx->next_handle++;
    ctx->textures[t->handle] = t;
    *out = t;
    return PT_OK;
}

int pt_tex_release(pt_tex *tex) {
    if(!tex) return PT_OK;
    (void)hipSetDevice(tex->ctx->device);
    (void)ctx_join(tex->ctx);
    (void)hipStreamSynchronize(tex->ctx->stream);
    unbind_everywhere(tex->ctx, nullptr, tex);
    tex->ctx->textures.erase(tex->handle);
    if(tex->dptr) (void)hipFree(tex->dptr);
    delete tex;
    return PT_OK;
}

uint64_t pt_tex_handle(const pt_tex *tex) { return tex ? tex->handle : 0; }

// ------------------------------------------------------------------------------- pass
int pt_pass_create(pt_ctx *ctx, int screen_w, int screen_h, pt_pass **out) {
    if(!ctx || !out) return fail(PT_ERR_INVALID, "pt_pass_create: null argument");
    if(screen_w <= 0 || screen_h <= 0 || (long long)screen_w * screen_h > (1ll << 28))
        return fail(PT_ERR_INVALID, "pt_pass_create: unsupported resolution");
    HIP_TRY(hipSetDevice(ctx->device));
    pt_pass *p = new pt_pass;
    p->ctx = ctx;
    p->W = screen_w;
    p->H = screen_h;
    memset(&p->params, 0, sizeof p->params);
    memset(&p->prof, 0, sizeof p->prof);
    const size_t npix = (size_t)screen_w * screen_h;
    int rc = PT_OK;
    {
        // order inside the groups: [direct | indirect | albedo], [motion | depth | normal | instance index]
        hipError_t e = hipMalloc(&p->d_accum_group, npix * 9 * sizeof(float));
        if(e == hipSuccess) e = hipMalloc(&p->d_lastwrite_group, npix * 7 * sizeof(float));
        if(e != hipSuccess) rc = fail(PT_ERR_HIP, std::string("hipMalloc (outputs): ") + hipGetErrorString(e));
        const int accum_order[3] = {5, 6, 0}, last_order[4] = {1, 2, 3, 4};  // indices into OUT_SLOTS
        size_t off = 0;
        for(int j = 0; j < 3 && !rc; j++) {
            const int k = accum_order[j];
            pt_buf *b = new pt_buf;
            b->ctx = ctx; b->owned = false;
            b->dptr = static_cast<float *>(p->d_accum_group) + off;
            b->bytes = npix * OUT_CHANNELS[k] * sizeof(float);
            off += npix * OUT_CHANNELS[k];
            p->outputs[k] = b;
        }
        off = 0;
        for(int j = 0; j < 4 && !rc; j++) {
            const int k = last_order[j];
            pt_buf *b = new pt_buf;
            b->ctx = ctx; b->owned = false;
            b->dptr = static_cast<float *>(p->d_lastwrite_group) + off;
            b->bytes = npix * OUT_CHANNELS[k] * sizeof(float);
            off += npix * OUT_CHANNELS[k];
            p->outputs[k] = b;
        }
    }
    // 6 blocks of 256 threads per CU: 24 KiB of LDS stack each, 24 waves per CU
    p->trace_blocks = ctx->num_cus * 8;
    set_draw_grid(p);
    p->redo_blocks = ctx->num_cus;
    if(!rc) rc = p->d_counters.ensure(CNT_TOTAL * 4);
    if(!rc) rc = p->d_stats.ensure(8 * 8);
    if(!rc) rc = p->d_err.ensure(64);  // bytes 16..47: traversal counters of RT_TRACE_STATS builds
    if(!rc) {
        void *h = nullptr, *d = nullptr;
        hipError_t e = hipHostMalloc(&h, 64, hipHostMallocMapped);
        if(e == hipSuccess) {
            memset(h, 0, 64);
            e = hipHostGetDevicePointer(&d, h, 0);
        }
        if(e != hipSuccess) rc = fail(PT_ERR_HIP, std::string("hipHostMalloc (overflow flag): ") + hipGetErrorString(e));
        p->h_err = static_cast<unsigned *>(h);
        p->d_err_flag = static_cast<unsigned *>(d);
    }
    if(!rc) {
        void *h = nullptr, *d = nullptr;
        hipError_t e = hipHostMalloc(&h, sizeof(QueueSnapshot), hipHostMallocMapped);
        if(e == hipSuccess) {
            memset(h, 0, sizeof(QueueSnapshot));
            e = hipHostGetDevicePointer(&d, h, 0);
        }
        if(e != hipSuccess) rc = fail(PT_ERR_HIP, std::string("hipHostMalloc (queue snapshot): ") + hipGetErrorString(e));
        p->h_snapshot = static_cast<QueueSnapshot *>(h);
        p->d_snapshot = static_cast<unsigned *>(d);
    }
    if(!rc) {
        hipError_t e = hipMemsetAsync(p->d_stats.p, 0, 64, ctx->stream);
        if(e == hipSuccess) e = hipMemsetAsync(p->d_err.p, 0, 64, ctx->stream);
        if(e == hipSuccess) e = hipMemsetAsync(p->d_counters.p, 0, CNT_TOTAL * 4, ctx->stream);
        if(e != hipSuccess) rc = fail(PT_ERR_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
    }
    if(rc) {
        pt_pass_destroy(p);
        return rc;
    }
    ctx->passes.push_back(p);
    *out = p;
    return PT_OK;
}

int pt_pass_destroy(pt_pass *p) {
    if(!p) return PT_OK;
    for(size_t k = 0; k < p->ctx->passes.size(); k++)
        if(p->ctx->passes[k] == p) {
            p->ctx->passes.erase(p->ctx->passes.begin() + (long)k);
            break;
        }
    (void)hipSetDevice(p->ctx->device);
    (void)ctx_join(p->ctx);
    (void)hipStreamSynchronize(p->ctx->stream);
    for(int k = 0; k < 7; k++)
        if(p->outputs[k]) {
            unbind_everywhere(p->ctx, p->outputs[k], nullptr);
            delete p->outputs[k];
        }
    if(p->d_accum_group) (void)hipFree(p->d_accum_group);
    if(p->d_lastwrite_group) (void)hipFree(p->d_lastwrite_group);
    collect_profile(p);
    for(pt_pass::PrimaryTuner &tu : p->primary_tuner)
        for(hipEvent_t e : {tu.a[0], tu.a[1], tu.b[0], tu.b[1]})
            if(e) (void)hipEventDestroy(e);
    for(hipEvent_t e : p->event_pool) (void)hipEventDestroy(e);
    for(pt_pass::Lane &l : p->lanes) release_lane(l);
    DeviceArray *arrs[] = {&p->d_mesh_buf, &p->d_scene_buf, &p->d_flat_buf, &p->d_redo_isect,
                           &p->d_spill, &p->d_tex_table, &p->d_counters, &p->d_stats, &p->d_err};
    for(DeviceArray *a : arrs) a->release();
    if(p->h_err) (void)hipHostFree(p->h_err);
    if(p->h_snapshot) (void)hipHostFree(p->h_snapshot);
    delete p;
    return PT_OK;
}

int pt_pass_bind(pt_pass *pass, int slot, pt_buf *buf) {
    if(!pass) return fail(PT_ERR_INVALID, "pt_pass_bind: null pass");
    if(slot < 0 || slot > 6) return fail(PT_ERR_INVALID, "pt_pass_bind: slot must be 0..6");
    if(buf && buf->ctx != pass->ctx) return fail(PT_ERR_INVALID, "pt_pass_bind: buffer belongs to another context");
    pass->slots[slot] = buf;
    return PT_OK;
}

int pt_pass_bind_texture(pt_pass *pass, const char *name, pt_tex *tex) {
    if(!pass || !name) return fail(PT_ERR_INVALID, "pt_pass_bind_texture: null argument");
    if(tex && tex->ctx != pass->ctx) return fail(PT_ERR_INVALID, "pt_pass_bind_texture: texture belongs to another context");
    struct Slot { const char *name; pt_tex **where; int min_channels; bool exact; };
    const Slot slots[] = {{"skybox", &pass->sky, 3, true},
                          {"skybox_samplecache", &pass->sky_cache, 3, true},
                          {"depthGBufferTexture", &pass->gb_depth, 1, false},
                          {"normalGBufferTexture", &pass->gb_normal, 3, false},
                          {"uvGBufferTexture", &pass->gb_uv, 2, false},
                          {"instanceIndexGBufferTexture", &pass->gb_inst, 1, false}};
    for(const Slot &sl : slots)
        if(!strcmp(name, sl.name)) {
            if(tex && (tex->format != PT_TEX_F32 || tex->channels < sl.min_channels || (sl.exact && tex->channels != sl.min_channels)))
                return fail(PT_ERR_INVALID, std::string("pt_pass_bind_texture: ") + name + " must be a float texture with " +
                                                (sl.exact ? "" : "at least ") + std::to_string(sl.min_channels) + " channels");
            *sl.where = tex;
            return PT_OK;
        }
    return fail(PT_ERR_INVALID, std::string("pt_pass_bind_texture: unknown sampler name ") + name);
}

int pt_pass_set_params(pt_pass *pass, const pt_params *params) {
    if(!pass || !params) return fail(PT_ERR_INVALID, "pt_pass_set_params: null argument");
    if(params->SCREEN_W != pass->W || params->SCREEN_H != pass->H)
        return fail(PT_ERR_INVALID, "pt_pass_set_params: SCREEN_W/H differ from the size the pass was created with");
    if(params->MAX_DEPTH < 0 || params->MAX_DEPTH > MAX_PATH_DEPTH)
        return fail(PT_ERR_INVALID, "pt_pass_set_params: MAX_DEPTH must be 0.." + std::to_string(MAX_PATH_DEPTH));
    if(params->lightCount < 0) return fail(PT_ERR_INVALID, "pt_pass_set_params: negative lightCount");
    if(params->accum_mode != 0 && params->accum_mode != 1) return fail(PT_ERR_INVALID, "pt_pass_set_params: accum_mode must be 0 or 1");
    if(params->RasterizaionFor1st && (params->WINDOW_W <= 0 || params->WINDOW_H <= 0))
        return fail(PT_ERR_INVALID, "pt_pass_set_params: RasterizaionFor1st needs positive WINDOW_W / WINDOW_H");
    pass->params = *params;
    pass->have_params = true;
    return PT_OK;
}

}  // extern "C"

namespace {
// One draw of `n` samples of the frame: prms[k] are the parameters of sample k (they differ in jitter, currentspp and
// frameCounter only).  The queues and launches are shared by the samples; finalize folds them in order.
int draw_samples(pt_pass *p, const pt_params *prms, int n) {
    pt_ctx *ctx = p->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = prepare_scene(p);
    if(rc) return rc;
    const pt_params &P = prms[n - 1];
    if(!p->sky) return fail(PT_ERR_STATE, "pt_pass_draw: texture \"skybox\" is not bound");
    if(P.SkyboxLighting && P.SkyboxSampling && !p->sky_cache)
        return fail(PT_ERR_STATE, "pt_pass_draw: texture \"skybox_samplecache\" is not bound");
    if(P.SKY_W <= 0 || P.SKY_H <= 0) return fail(PT_ERR_INVALID, "pt_pass_draw: SKY_W / SKY_H must be positive");
    const bool raster_first = P.RasterizaionFor1st != 0;
    if(raster_first && (!p->gb_depth || !p->gb_normal || !p->gb_uv || !p->gb_inst))
        return fail(PT_ERR_STATE, "pt_pass_draw: RasterizaionFor1st needs the textures depthGBufferTexture, normalGBufferTexture, "
                                  "uvGBufferTexture and instanceIndexGBufferTexture");
    if(P.lightCount > 0) {
        pt_buf *lb = p->slots[PT_SLOT_LIGHT];
        if(!lb || lb->bytes < (size_t)P.lightCount * sizeof(LightInfo))
            return fail(PT_ERR_STATE, "pt_pass_draw: lightCount exceeds the bound light buffer");
    }
    const size_t npix1 = state_slots(p->W, p->H);
    const size_t npix = npix1 * (size_t)n;  // (slot numbers fit the 28 bits the tail launch keeps them in: checked by the callers)
    const int n_shadow = 1 + P.lightCount;
    pt_pass::Lane &L = p->lanes[p->next_lane];
    p->next_lane = (p->next_lane + 1) % p->n_lanes;
    if((rc = prepare_lane(p, L, n_shadow, n))) return rc;
    // A draw that finds no other draw of this pass on the device takes the full grid (a caller
    // that waits for every frame should not pay for leaving room to draws that never come).
    int grid = p->draw_blocks;
    bool alone = true;
    if(p->n_lanes > 1) {
        for(int k = 0; k < p->n_lanes && alone; k++)
            if(p->lanes[k].done && hipEventQuery(p->lanes[k].done) == hipErrorNotReady) alone = false;
        if(alone) grid = p->trace_blocks;
    }

    DrawParams dp;
    memset(&dp, 0, sizeof dp);
    dp.p = P;
    dp.tan_half_fov = std::tan(P.fov / 2.0f);
    dp.W = p->W; dp.H = p->H; dp.npix = (int)npix;
    dp.npix1 = (int)npix1;
    dp.batch = n;
    for(int k = 0; k < n; k++) {
        dp.samples[k].jitter[0] = prms[k].jitter[0];
        dp.samples[k].jitter[1] = prms[k].jitter[1];
        dp.samples[k].currentspp = prms[k].currentspp;
        dp.samples[k].seed_term = prms[k].frameCounter * prms[k].currentspp * 26699u;
    }
    dp.tiles_x = (p->W + STATE_TILE_W - 1) / STATE_TILE_W;
    dp.n_shadow = n_shadow;
    dp.materials = static_cast<const float *>(p->slots[PT_SLOT_MATERIAL]->dptr);
    dp.n_material_floats = (int)p->n_material_floats;
    dp.tris = static_cast<const TriangleRec *>(p->slots[PT_SLOT_TRIANGLE]->dptr);
    dp.inst = reinterpret_cast<const InstAccel *>(static_cast<const char *>(p->d_scene_buf.p) + (size_t)p->scene_level.header.inst_base * 64);
    dp.lights = p->slots[PT_SLOT_LIGHT] ? static_cast<const LightInfo *>(p->slots[PT_SLOT_LIGHT]->dptr) : nullptr;
    dp.textures = static_cast<const TexDesc *>(p->d_tex_table.p);
    dp.n_textures = p->n_textures;
    dp.sky = desc_of(p->sky);
    dp.sky_cache = p->sky_cache ? desc_of(p->sky_cache) : desc_of(p->sky);
    if(raster_first) {
        dp.gb_depth = desc_of(p->gb_depth);
        dp.gb_normal = desc_of(p->gb_normal);
        dp.gb_uv = desc_of(p->gb_uv);
        dp.gb_inst = desc_of(p->gb_inst);
    } else if(p->gb_depth && p->gb_inst) {
        // the reference reads these two even when it traces the first hit (shader/pathtracing.glsl:615-616)
        dp.gb_depth = desc_of(p->gb_depth);
        dp.gb_inst = desc_of(p->gb_inst);
        dp.raster_outputs = 1;
    }
    dp.n_instances = (int)p->scene_level.header.n_instances;
    dp.rays = static_cast<PathRay *>(L.rays.p);
    dp.hits = static_cast<PathHit *>(L.hits.p);
    dp.thr = static_cast<PathThroughput *>(L.thr.p);
    dp.acc = static_cast<PathAccum *>(L.acc.p);
    dp.shadow = static_cast<ShadowEntry *>(L.shadow.p);
    float *outs[7];
    for(int k = 0; k < 7; k++) {
        outs[k] = static_cast<float *>(p->outputs[k]->dptr);
        p->outputs[k]->version++;
        p->outputs[k]->shadow_valid = false;
    }
    dp.o_albedo = outs[0]; dp.o_motion = outs[1]; dp.o_depth = outs[2]; dp.o_normal = outs[3];
    dp.o_inst = outs[4]; dp.o_direct = outs[5]; dp.o_indirect = outs[6];
    dp.stats = static_cast<unsigned long long *>(p->d_stats.p);
    unsigned *cnt = static_cast<unsigned *>(L.counters.p);
    dp.counters = cnt;
    dp.redo_counter_base = use_ordered(p) ? CNT_REDO : -1;

    hipStream_t s = L.stream;
    p->cur_stream = s;
    // behind everything issued on the main stream so far (uploads, fills of the outputs)
    HIP_TRY(hipEventRecord(L.start, ctx->stream));
    HIP_TRY(hipStreamWaitEvent(s, L.start, 0));
    const TraceScene sc = make_trace_scene(p, L.spill.p);
    const FastScene fsc = make_fast_scene(p, L.spill.p);
    const bool ordered = use_ordered(p);
    unsigned *qr = static_cast<unsigned *>(L.queue_r.p);
    HIP_TRY(hipMemsetAsync(cnt, 0, CNT_TOTAL * 4, s));
    // camera rays: a launch of their own, unless the ordered extend of vertex 0 makes them where it traces them
    const bool fused_primary = ordered && !raster_first;
    // camera rays of a tile walk the flattened tree together when its deepest walk fits the wave's stack (RTA_PRIMARY_PACKETS=0:
    // one ray per lane, a tuning aid)
    static const int packets_forced = getenv("RTA_PRIMARY_PACKETS") ? atoi(getenv("RTA_PRIMARY_PACKETS")) : -1;  // 0 / 1: a tuning aid
    const bool packets_possible = fused_primary && fsc.flat_buf != nullptr && p->flat.max_stack + 2 <= packet_stack_entries();
    bool packets = packets_possible && packets_forced != 0;
    int tuner_kind = -1;       // >= 0: this draw's primary launch is timed as a measurement of that kind
    bool tuner_halves = false;  // this draw's samples go half through each kind, both timed
    if(packets_possible && packets_forced < 0) {
        pt_pass::PrimaryTuner &tu = p->primary_tuner[n >= 4 ? 1 : 0];
        if(tu.flat_build != p->flat_builds) {  // another tree: measure again
            tu.flat_build = p->flat_builds;
            tu.decided = -1;
            tu.pending[0] = tu.pending[1] = false;
            tu.ms_per_slot[0] = tu.ms_per_slot[1] = 0.0;
            tu.draws_seen = 0;
        }
        const bool cold = tu.draws_seen++ == 0;
        for(int k = 0; k < 2; k++)
            if(tu.pending[k] && hipEventQuery(tu.b[k]) == hipSuccess) {
                float ms = 0.0f;
                if(hipEventElapsedTime(&ms, tu.a[k], tu.b[k]) == hipSuccess && tu.slots[k]) tu.ms_per_slot[k] = (double)ms / (double)tu.slots[k];
                tu.pending[k] = false;
            }
        if(tu.decided < 0 && tu.ms_per_slot[0] > 0.0 && tu.ms_per_slot[1] > 0.0) {
            // (launches of different sizes do not compare: a thin one waits for its slowest walk, a fat one fills the chip)
            if((tu.halves[0] && tu.halves[1]) || tu.slots[0] == tu.slots[1]) tu.decided = tu.ms_per_slot[1] <= tu.ms_per_slot[0] ? 1 : 0;
            else tu.ms_per_slot[tu.slots[0] == npix ? 1 : 0] = 0.0;  // keep the one that was taken at this draw's size
            static const bool log = getenv("RTA_TUNER_LOG") != nullptr;  // (a tuning aid)
            if(log && tu.decided >= 0)
                fprintf(stderr, "primary launch: %.3f ns per slot one ray per lane (%zu slots), %.3f as tiles (%zu slots): %s\n",
                        tu.ms_per_slot[0] * 1e6, tu.slots[0], tu.ms_per_slot[1] * 1e6, tu.slots[1], tu.decided ? "tiles" : "per lane");
        }
        // Nothing to time where a tile of 8 x 8 pixels sees a triangle or two: fewer than one (instance, triangle) pair per
        // four pixels (the demo scene at 1920x1080: one per sixty, tiles 42 % faster; ten million triangles at 3840x2160: 1.2
        // per pixel, tiles 2.5 % slower -- that side of the line is measured).  A timed draw costs: its vertex 0 runs as three
        // launches instead of one.
        const bool sparse = (double)p->flat.n_leaves * 4.0 < (double)npix1;
        if(tu.decided >= 0) {
            packets = tu.decided == 1;
        } else if(sparse || cold || p->profiling || tu.pending[0] || tu.pending[1]) {
            packets = true;  // (not to be timed, a measurement is under way, or the per-kernel profile is being taken: the likelier kind)
        } else if(n >= 4) {
            for(int k = 0; k < 2; k++)
                for(hipEvent_t *e : {&tu.a[k], &tu.b[k]})
                    if(!*e) HIP_TRY(hipEventCreate(e));
            tuner_halves = true;
            if(getenv("RTA_TUNER_LOG")) fprintf(stderr, "primary launch: timing both kinds on a draw of %d samples\n", n);
        } else {
            // a thin draw: the kind that has no figure yet (packets first)
            const int k = tu.ms_per_slot[1] == 0.0 ? 1 : 0;
            for(hipEvent_t *e : {&tu.a[k], &tu.b[k]})
                if(!*e) HIP_TRY(hipEventCreate(e));
            tuner_kind = k;
            packets = k == 1;
        }
    }
    if(!fused_primary) {
        ScopedTimer t(p, K_OTHER);
        launch_generate(s, dp);
    }
    unsigned *qa = static_cast<unsigned *>(L.queue_a.p), *qb = static_cast<unsigned *>(L.queue_b.p);
    unsigned *qs = static_cast<unsigned *>(L.queue_s.p);
    const unsigned *q_in = nullptr;  // vertex 0: every pixel, in pixel order
    // In the ordered mode without analytic lights, the shadow rays of vertex d and the extend
    // rays of vertex d+1 (both produced by shade(d)) share one launch.
    const bool merged = ordered && n_shadow == 1;
    // The tail launch pays when the chip would otherwise idle through the thin vertices: a draw on its own.
    // With other draws in flight those vertices hide behind the others' wide ones (measured: 2-5 % slower with it).
    const int tail_from = choose_tail_vertex(p, P, npix, merged && alone);
    dp.tail_from = tail_from;
    dp.snapshot = p->d_snapshot;
    auto extend = [&](int dep, const unsigned *q) {
        ScopedTimer t(p, K_EXTEND);
        if(ordered) {
            if(dep == 0 && fused_primary && tuner_halves) {
                // the first samples as packets, the others one ray per lane, each launch between two events
                pt_pass::PrimaryTuner &tu = p->primary_tuner[1];
                const unsigned first = (unsigned)(n / 2) * (unsigned)npix1, rest = (unsigned)npix - first;
                (void)hipEventRecord(tu.a[1], s);
                launch_primary_fast(s, fsc, grid, dp, cnt + CNT_NEXT, qr, cnt + CNT_REDO, true, 0u, first, true);
                (void)hipEventRecord(tu.b[1], s);
                (void)hipEventRecord(tu.a[0], s);
                launch_primary_fast(s, fsc, grid, dp, cnt + CNT_TUNE, qr, cnt + CNT_REDO, false, first, rest, false);
                (void)hipEventRecord(tu.b[0], s);
                tu.slots[1] = first;
                tu.slots[0] = rest;
                tu.pending[0] = tu.pending[1] = true;
                tu.halves[0] = tu.halves[1] = true;
            } else if(dep == 0 && fused_primary) {
                pt_pass::PrimaryTuner &tu = p->primary_tuner[0];  // (draws of up to three samples are timed whole)
                if(tuner_kind >= 0) (void)hipEventRecord(tu.a[tuner_kind], s);
                launch_primary_fast(s, fsc, grid, dp, cnt + CNT_NEXT + 2 * dep, qr, cnt + CNT_REDO + dep, packets, 0u, (unsigned)npix, true);
                if(tuner_kind >= 0) {
                    (void)hipEventRecord(tu.b[tuner_kind], s);
                    tu.slots[tuner_kind] = npix;
                    tu.pending[tuner_kind] = true;
                    tu.halves[tuner_kind] = false;
                }
            }
            else
                launch_extend_fast(s, fsc, grid, q, cnt + CNT_QUEUE + dep, dp.rays, dp.hits, cnt + CNT_NEXT + 2 * dep, qr,
                                   cnt + CNT_REDO + dep);
            // near-ties in t: decided by the reference-order traversal
            launch_extend(s, sc, p->redo_blocks, qr, cnt + CNT_REDO + dep, dp.rays, dp.hits, cnt + CNT_NEXT_REDO + dep, true);
        } else {
            launch_extend(s, sc, grid, q, cnt + CNT_QUEUE + dep, dp.rays, dp.hits, cnt + CNT_NEXT + 2 * dep, false);
        }
    };
    // Camera rays walked as packets and vertex 0 shaded in ONE launch (trace_kernels.hip), when the first hit is a plain
    // traced one and each stage need not be timed apart; the slots that launch leaves to the reference-order pass are
    // traced by it and shaded by a small launch over their list.  RTA_FUSED_PRIMARY=0: two launches, a tuning aid.
    static const bool fusion_off = getenv("RTA_FUSED_PRIMARY") && atoi(getenv("RTA_FUSED_PRIMARY")) == 0;
    const bool fuse_first = !fusion_off && fused_primary && packets && tuner_kind < 0 && !tuner_halves && !p->profiling && !dp.raster_outputs &&
                            tail_from != 0 && p->material_grouping <= 0;
    // Tiles whose paths leave in nearly one direction (a polished floor, a pane) are walked as groups once more at vertex 1
    // (trace_kernels.hip: extend_packet_kernel) -- in a draw fat enough for a launch over some of its tiles to fill the chip:
    // the walk of a group is one dependent chain per wave, and a thin launch of such walks waits for its slowest
    // (single 1920x1080 samples: 2.91 -> 3.44 ms with groups, sixteen samples in a draw: 1.47 -> 1.40 ms per sample).
    // RTA_BOUNCE_GROUPS=0: never, 2: whenever possible (tuning and test aid, read at every draw).
    const char *groups_env = getenv("RTA_BOUNCE_GROUPS");
    const int groups_mode = groups_env ? atoi(groups_env) : 1;
    const bool bounce_groups = fuse_first && groups_mode != 0 && merged && tail_from != 1 && P.MAX_DEPTH >= 1 &&
                               (groups_mode == 2 || npix >= GROUPS_FROM_SLOTS);
    unsigned *qp = bounce_groups ? static_cast<unsigned *>(L.queue_p.p) : nullptr;
    static const bool sgroups_off = getenv("RTA_SHADOW_GROUPS") && atoi(getenv("RTA_SHADOW_GROUPS")) == 0;
    unsigned *qps = bounce_groups && !sgroups_off ? static_cast<unsigned *>(L.queue_ps.p) : nullptr;
    if(fuse_first) {
        // (this launch writes the last-write outputs: after the previous draw's)
        if(p->last_gbuf && p->last_gbuf != L.gbuf) HIP_TRY(hipStreamWaitEvent(s, p->last_gbuf, 0));
        launch_primary_shade_packets(s, fsc, grid, dp, cnt + CNT_NEXT, qr, cnt + CNT_REDO, qa, cnt + CNT_QUEUE + 1, qs, cnt + CNT_SHADOW, qp,
                                     cnt + CNT_GROUPS, cnt + CNT_GROUPS + 1, qps, cnt + CNT_SGROUPS, cnt + CNT_SGROUPS + 1);
        launch_extend(s, sc, p->redo_blocks, qr, cnt + CNT_REDO, dp.rays, dp.hits, cnt + CNT_NEXT_REDO, true);
        launch_shade(s, dp, 0, qr, cnt + CNT_REDO, qa, cnt + CNT_QUEUE + 1, qs, cnt + CNT_SHADOW, false, false, SHADE_LISTED);
        HIP_TRY(hipEventRecord(L.gbuf, s));
        p->last_gbuf = L.gbuf;
    } else if(!raster_first) {
        extend(0, q_in);  // with RasterizaionFor1st the first hit comes from the raster images
    }
    for(int dep = 0; dep <= P.MAX_DEPTH; dep++) {
        if(dep == tail_from) {
            // few paths are left: one launch carries them through all their remaining vertices
            ScopedTimer t(p, K_TAIL);
            const unsigned expect = p->h_snapshot->alive[dep];  // (stale at worst; the launch is persistent, any grid is correct)
            long long blocks = ((long long)expect + tail_paths_per_block() - 1) / tail_paths_per_block() + 8;
            if(blocks > p->trace_blocks) blocks = p->trace_blocks;
            launch_path_tail(s, fsc, sc, dp, dep, (int)blocks, q_in, cnt + CNT_QUEUE + dep, cnt + CNT_TAIL);
            break;
        }
        unsigned *q_out = (dep & 1) ? qb : qa;
        // shading the first vertex writes the last-write outputs: after the previous draw's
        if(dep == 0 && !fuse_first && p->last_gbuf && p->last_gbuf != L.gbuf) HIP_TRY(hipStreamWaitEvent(s, p->last_gbuf, 0));
        if(!(dep == 0 && fuse_first)) {
            ScopedTimer t(p, K_SHADE);
            // The library's choice, for a draw that is alone on the device: from vertex 1 on, every block moves the
            // paths that escaped behind the ones that hit something (at vertex 0 the sky fills whole tiles: nothing to
            // sort).  Shade time -10 % (demo scene) / -11 % (10 M triangles), a lone draw -2.6 %; with draws in flight
            // the frame does not gain (0 .. +2.5 % time: the larger blocks and their barriers pack less well beside the
            // traversal launches of the other draws), so those launches stay ungrouped.  By material type only when asked:
            // on a 10 M-triangle scene with 50 % dielectric instances, 16 albedo maps, MAX_DEPTH 16 at 3840x2160 that
            // measured shade time +1.4 %, frame -1 % -- neighbouring pixels mostly hit the same instance already.
            const bool by_material = p->material_grouping > 0;
            const bool group = by_material || (p->material_grouping < 0 && dep >= 1 && alone);
            launch_shade(s, dp, dep, q_in, cnt + CNT_QUEUE + dep, q_out, cnt + CNT_QUEUE + dep + 1, qs, cnt + CNT_SHADOW + dep, group, by_material);
            if(dep == 1 && bounce_groups)  // (the paths that reached this vertex as groups)
                launch_shade(s, dp, dep, qp, cnt + CNT_GROUPS, q_out, cnt + CNT_QUEUE + dep + 1, qs, cnt + CNT_SHADOW + dep, false, false,
                             SHADE_GROUPS, cnt + CNT_GROUPS + 1);
        }
        if(dep == 0 && !fuse_first) {
            HIP_TRY(hipEventRecord(L.gbuf, s));
            p->last_gbuf = L.gbuf;
        }
        if(dep < P.MAX_DEPTH) {
            if(merged) {
                ScopedTimer t(p, K_TRACE);
                launch_path_trace_fast(s, fsc, grid, q_out, cnt + CNT_QUEUE + dep + 1, qs, cnt + CNT_SHADOW + dep, dp.rays,
                                       dp.hits, dp.shadow, (int)npix, cnt + CNT_NEXT + 2 * dep + 1, qr, cnt + CNT_REDO + dep + 1, dp.thr,
                                       dp.acc, dp.stats + 2);
                if(dep == 0 && bounce_groups)
                    launch_extend_packets(s, fsc, grid, qp, cnt + CNT_GROUPS, dp.rays, dp.hits, cnt + CNT_GROUPS + 2, qr, cnt + CNT_REDO + 1);
                if(dep == 0 && qps)
                    launch_shadow_packets(s, fsc, grid, qps, cnt + CNT_SGROUPS, cnt + CNT_SGROUPS + 1, dp.rays, dp.shadow, dp.thr, dp.acc,
                                          cnt + CNT_SGROUPS + 2, qr, cnt + CNT_REDO + 1, dp.stats);
                launch_path_redo(s, sc, p->redo_blocks, qr, cnt + CNT_REDO + dep + 1, dp.rays, dp.hits, dp.shadow, dp.thr, dp.acc,
                                 cnt + CNT_NEXT_REDO + dep + 1);
            } else {
                {
                    // (analytic lights: the ordered traversal, what it leaves undecided in the reference's order behind it;
                    // the reference's order asked for: all of them in that order)
                    ScopedTimer t(p, K_SHADOW);
                    if(ordered && n_shadow > 1) {
                        unsigned *qsr = static_cast<unsigned *>(L.queue_sr.p);
                        launch_shadow_fast(s, fsc, grid, qs, cnt + CNT_SHADOW + dep, dp.rays, dp.shadow, n_shadow, (int)npix,
                                           cnt + CNT_NEXT + 2 * dep + 1, qsr, cnt + CNT_SHADOW_REDO + dep);
                        launch_shadow_redo(s, sc, p->redo_blocks, qs, dp.rays, dp.shadow, n_shadow, (int)npix, qsr,
                                           cnt + CNT_SHADOW_REDO + dep, cnt + CNT_NEXT_SHADOW_REDO + dep);
                    } else {
                        launch_shadow(s, sc, grid, qs, cnt + CNT_SHADOW + dep, dp.rays, dp.shadow, n_shadow, (int)npix,
                                      cnt + CNT_NEXT + 2 * dep + 1);
                    }
                }
                extend(dep + 1, q_out);
            }
            if(!merged) {  // (the merged launch applies each light sample where its occlusion query ends)
                ScopedTimer t(p, K_OTHER);
                launch_shadow_apply(s, dp, qs, cnt + CNT_SHADOW + dep);
            }
        }
        q_in = q_out;
    }
    // the running mean folds samples in call order
    if(p->last_fin && p->last_fin != L.fin) HIP_TRY(hipStreamWaitEvent(s, p->last_fin, 0));
    {
        ScopedTimer t(p, K_OTHER);
        launch_finalize(s, dp);
    }
    HIP_TRY(hipEventRecord(L.fin, s));
    p->last_fin = L.fin;
    HIP_TRY(hipEventRecord(L.done, s));
    bool listed = false;
    for(hipEvent_t e : ctx->pending) listed = listed || e == L.done;
    if(!listed) ctx->pending.push_back(L.done);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int check_params(const pt_pass *pass, const pt_params *params, const char *fn) {
    if(params->SCREEN_W != pass->W || params->SCREEN_H != pass->H)
        return fail(PT_ERR_INVALID, std::string(fn) + ": SCREEN_W/H differ from the size the pass was created with");
    if(params->MAX_DEPTH < 0 || params->MAX_DEPTH > MAX_PATH_DEPTH)
        return fail(PT_ERR_INVALID, std::string(fn) + ": MAX_DEPTH must be 0.." + std::to_string(MAX_PATH_DEPTH));
    if(params->lightCount < 0) return fail(PT_ERR_INVALID, std::string(fn) + ": negative lightCount");
    if(params->accum_mode != 0 && params->accum_mode != 1) return fail(PT_ERR_INVALID, std::string(fn) + ": accum_mode must be 0 or 1");
    if(params->RasterizaionFor1st && (params->WINDOW_W <= 0 || params->WINDOW_H <= 0))
        return fail(PT_ERR_INVALID, std::string(fn) + ": RasterizaionFor1st needs positive WINDOW_W / WINDOW_H");
    return PT_OK;
}
}  // namespace

extern "C" {

int pt_pass_draw(pt_pass *p) {
    if(!p) return fail(PT_ERR_INVALID, "pt_pass_draw: null pass");
    if(!p->have_params) return fail(PT_ERR_STATE, "pt_pass_draw: pt_pass_set_params has not been called");
    return draw_samples(p, &p->params, 1);
}

int pt_pass_draw_batch(pt_pass *p, const pt_params *params, int n) {
    if(!p || !params) return fail(PT_ERR_INVALID, "pt_pass_draw_batch: null argument");
    if(n < 1 || n > MAX_BATCH) return fail(PT_ERR_INVALID, "pt_pass_draw_batch: n must be 1.." + std::to_string(MAX_BATCH));
    if(state_slots(p->W, p->H) * (size_t)n >= ((size_t)1 << 28))
        return fail(PT_ERR_INVALID, "pt_pass_draw_batch: " + std::to_string(n) + " samples of this resolution exceed 2^28 path-state slots; use fewer per call");
    for(int k = 0; k < n; k++) {
        int rc = check_params(p, params + k, "pt_pass_draw_batch");
        if(rc) return rc;
        // the samples of one frame: everything but jitter, currentspp and frameCounter is the frame's
        pt_params a = params[k], b = params[0];
        a.jitter[0] = b.jitter[0]; a.jitter[1] = b.jitter[1];
        a.currentspp = b.currentspp;
        a.frameCounter = b.frameCounter;
        if(memcmp(&a, &b, sizeof a) != 0)
            return fail(PT_ERR_INVALID, "pt_pass_draw_batch: the parameter sets of a batch may differ in jitter, currentspp and frameCounter only (sample " +
                                            std::to_string(k) + " differs from sample 0 elsewhere)");
    }
    p->params = params[n - 1];
    p->have_params = true;
    return draw_samples(p, params, n);
}

int pt_pass_reserve_batch(pt_pass *pass, int n) {
    if(!pass) return fail(PT_ERR_INVALID, "pt_pass_reserve_batch: null pass");
    if(n < 1 || n > MAX_BATCH) return fail(PT_ERR_INVALID, "pt_pass_reserve_batch: n must be 1.." + std::to_string(MAX_BATCH));
    pass->batch_reserve = n;
    return PT_OK;
}

int pt_pass_set_pipeline(pt_pass *pass, int draws_in_flight) {
    if(!pass) return fail(PT_ERR_INVALID, "pt_pass_set_pipeline: null pass");
    if(draws_in_flight < 1 || draws_in_flight > pt_pass::MAX_LANES)
        return fail(PT_ERR_INVALID, "pt_pass_set_pipeline: draws in flight must be 1.." + std::to_string(pt_pass::MAX_LANES));
    HIP_TRY(hipSetDevice(pass->ctx->device));
    JOIN_TRY(pass->ctx);
    HIP_TRY(hipStreamSynchronize(pass->ctx->stream));
    for(int k = draws_in_flight; k < pt_pass::MAX_LANES; k++) {
        if(pass->last_gbuf == pass->lanes[k].gbuf) pass->last_gbuf = nullptr;
        if(pass->last_fin == pass->lanes[k].fin) pass->last_fin = nullptr;
        release_lane(pass->lanes[k]);
    }
    pass->n_lanes = draws_in_flight;
    set_draw_grid(pass);
    pass->next_lane = 0;
    return PT_OK;
}

int pt_pass_set_tail_threshold(pt_pass *pass, int threshold) {
    if(!pass) return fail(PT_ERR_INVALID, "pt_pass_set_tail_threshold: null pass");
    if(threshold < -1) return fail(PT_ERR_INVALID, "pt_pass_set_tail_threshold: threshold must be -1 (library's choice), 0 (off) or a path count");
    pass->tail_threshold = threshold;
    return PT_OK;
}

int pt_pass_set_material_grouping(pt_pass *pass, int mode) {
    if(!pass) return fail(PT_ERR_INVALID, "pt_pass_set_material_grouping: null pass");
    if(mode < -1 || mode > 1) return fail(PT_ERR_INVALID, "pt_pass_set_material_grouping: mode must be -1 (library's choice), 0 (never) or 1 (always)");
    pass->material_grouping = mode;
    return PT_OK;
}

int pt_pass_output(pt_pass *pass, int which, pt_buf **out) {
    if(!pass || !out) return fail(PT_ERR_INVALID, "pt_pass_output: null argument");
    for(int k = 0; k < 7; k++)
        if(OUT_SLOTS[k] == which) {
            *out = pass->outputs[k];
            return PT_OK;
        }
    return fail(PT_ERR_INVALID, "pt_pass_output: unknown output " + std::to_string(which));
}

int pt_pass_stats(pt_pass *pass, pt_stats *out, int reset) {
    if(!pass || !out) return fail(PT_ERR_INVALID, "pt_pass_stats: null argument");
    HIP_TRY(hipSetDevice(pass->ctx->device));
    JOIN_TRY(pass->ctx);
    HIP_TRY(hipStreamSynchronize(pass->ctx->stream));
    unsigned long long raw[8];
    HIP_TRY(hipMemcpy(raw, pass->d_stats.p, sizeof raw, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    out->samples = raw[0];
    out->rays_extend = raw[1];
    out->rays_shadow = raw[2] + raw[5];  // (traced ones + those decided without a traversal)
    out->draws = raw[3];
    out->rays_retraced = raw[4];
    out->rays_shadow_null = raw[5];
    out->rays_extend_grouped = raw[6];
    out->rays_shadow_grouped = raw[7];
    out->mesh_level_builds = pass->mesh_level_builds;
    out->flat_leaves = pass->flat_valid ? (uint64_t)pass->flat.n_leaves : 0;
    out->reference_order_forced = pass->mesh_level_valid && pass->scene_level_valid && !(pass->mesh_level.nested && pass->scene_level.nested) ? 1 : 0;
    out->scene_level_builds = pass->scene_level_builds;
    HIP_TRY(hipMemcpy(out->debug, static_cast<char *>(pass->d_err.p) + 16, 24, hipMemcpyDeviceToHost));
    if(reset) {
        HIP_TRY(hipMemsetAsync(pass->d_stats.p, 0, 64, pass->ctx->stream));
        HIP_TRY(hipMemsetAsync(static_cast<char *>(pass->d_err.p) + 16, 0, 32, pass->ctx->stream));
    }
    return ctx_check_overflow(pass->ctx);
}

int pt_pass_set_traversal(pt_pass *pass, int mode) {
    if(!pass) return fail(PT_ERR_INVALID, "pt_pass_set_traversal: null pass");
    if(mode < 0 || mode > 3)
        return fail(PT_ERR_INVALID, "pt_pass_set_traversal: mode must be 0 (ordered), 1 (reference order), 2 (ordered, two levels only) or 3 (ordered, flattened tree at once)");
    if(mode != pass->traversal_mode) {
        pass->flat_refused = false;  // (mode 3 builds what mode 0 refuses)
        if(pass->flat_valid && mode == 0 && pass->flat_forced) pass->flat_valid = false;  // mode 0 decides for itself
    }
    pass->traversal_mode = mode;
    return PT_OK;
}

int pt_pass_prepare(pt_pass *pass) {
    if(!pass) return fail(PT_ERR_INVALID, "pt_pass_prepare: null pass");
    HIP_TRY(hipSetDevice(pass->ctx->device));
    int rc = prepare_scene(pass, true);
    if(rc) return rc;
    // the path state of every draw that may be in flight (a draw allocates its lane's at first use otherwise:
    // eleven device allocations, a millisecond or two inside somebody's frame)
    const int n_shadow = pass->have_params ? 1 + pass->params.lightCount : 1;
    for(int k = 0; k < pass->n_lanes; k++)
        if((rc = prepare_lane(pass, pass->lanes[k], n_shadow, pass->batch_reserve))) return rc;
    return PT_OK;
}

int pt_pass_set_wide_nodes(pt_pass *pass, int format) {
    if(!pass) return fail(PT_ERR_INVALID, "pt_pass_set_wide_nodes: null pass");
    if(format < -1 || format > 1) return fail(PT_ERR_INVALID, "pt_pass_set_wide_nodes: format must be -1 (by scene size), 0 (full precision) or 1 (quantised)");
    if(pass->quantise != format) {
        pass->quantise = format;
        pass->mesh_level_valid = false;  // rebuilt by the next draw / intersect
    }
    return PT_OK;
}

int pt_pass_set_profiling(pt_pass *pass, int enable) {
    if(!pass) return fail(PT_ERR_INVALID, "pt_pass_set_profiling: null pass");
    pass->profiling = enable != 0;
    return PT_OK;
}

int pt_pass_profile(pt_pass *pass, pt_profile *out, int reset) {
    if(!pass || !out) return fail(PT_ERR_INVALID, "pt_pass_profile: null argument");
    HIP_TRY(hipSetDevice(pass->ctx->device));
    JOIN_TRY(pass->ctx);
    HIP_TRY(hipStreamSynchronize(pass->ctx->stream));
    collect_profile(pass);
    *out = pass->prof;
    if(reset) memset(&pass->prof, 0, sizeof pass->prof);
    return ctx_check_overflow(pass->ctx);
}

int pt_pass_intersect(pt_pass *p, const pt_buf *rays, size_t n, pt_buf *hits) {
    if(!p || !rays || !hits) return fail(PT_ERR_INVALID, "pt_pass_intersect: null argument");
    if(rays->ctx != p->ctx || hits->ctx != p->ctx) return fail(PT_ERR_INVALID, "pt_pass_intersect: buffers belong to another context");
    if(n > 0xFFFFFF00ull) return fail(PT_ERR_INVALID, "pt_pass_intersect: too many rays for one call");
    if(rays->bytes < n * 6 * sizeof(float)) return fail(PT_ERR_INVALID, "pt_pass_intersect: ray buffer holds fewer than n rays");
    if(hits->bytes < n * sizeof(pt_hit)) return fail(PT_ERR_INVALID, "pt_pass_intersect: hit buffer holds fewer than n records");
    HIP_TRY(hipSetDevice(p->ctx->device));
    int rc = prepare_scene(p);
    if(rc) return rc;
    hits->version++;
    hits->shadow_valid = false;
    release_large_shadow(rays);  // (nothing on the host reads a ray buffer)
    if(n == 0) return PT_OK;
    JOIN_TRY(p->ctx);
    unsigned *cnt = static_cast<unsigned *>(p->d_counters.p);
    hipStream_t s = p->ctx->stream;
    p->cur_stream = s;
    HIP_TRY(hipMemsetAsync(cnt + CNT_INTERSECT, 0, 12, s));
    const TraceScene sc = make_trace_scene(p);
    if(use_ordered(p) && (rc = p->d_redo_isect.ensure(n * 4))) return rc;
    long long blocks = ((long long)n + TRACE_BLOCK - 1) / TRACE_BLOCK;
    if(blocks > p->trace_blocks) blocks = p->trace_blocks;
    {
        ScopedTimer t(p, K_INTERSECT);
        if(use_ordered(p)) {
            unsigned *redo = static_cast<unsigned *>(p->d_redo_isect.p);
            launch_intersect_fast(s, make_fast_scene(p), (int)blocks, static_cast<const float *>(rays->dptr), (unsigned)n,
                                  static_cast<pt_hit *>(hits->dptr), cnt + CNT_INTERSECT, redo, cnt + CNT_INTERSECT + 1);
            launch_intersect(s, sc, p->redo_blocks, static_cast<const float *>(rays->dptr), (unsigned)n, redo,
                             cnt + CNT_INTERSECT + 1, static_cast<pt_hit *>(hits->dptr), cnt + CNT_INTERSECT + 2,
                             static_cast<unsigned long long *>(p->d_stats.p) + 4);
        } else {
            launch_intersect(s, sc, (int)blocks, static_cast<const float *>(rays->dptr), (unsigned)n, nullptr, nullptr,
                             static_cast<pt_hit *>(hits->dptr), cnt + CNT_INTERSECT, nullptr);
        }
    }
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

// pt_pass_intersect for callers that must not allocate: every byte of device memory beyond the
// caller's rays and hits comes from this GPU's copy of the file-scope scratch array; no
// hipMalloc / hipFree, no event creation.  Same kernels, same results.
int pt_pass_intersect_noalloc(pt_pass *p, const pt_buf *rays, size_t n, pt_buf *hits) {
    if(!p || !rays || !hits) return fail(PT_ERR_INVALID, "pt_pass_intersect_noalloc: null argument");
    if(rays->ctx != p->ctx || hits->ctx != p->ctx) return fail(PT_ERR_INVALID, "pt_pass_intersect_noalloc: buffers belong to another context");
    if(n > NOALLOC_MAX_RAYS)
        return fail(PT_ERR_INVALID, "pt_pass_intersect_noalloc: " + std::to_string(n) + " rays exceed the limit of " +
                                        std::to_string(NOALLOC_MAX_RAYS) + " per call (use pt_pass_intersect, or several calls)");
    if(rays->bytes < n * 6 * sizeof(float)) return fail(PT_ERR_INVALID, "pt_pass_intersect_noalloc: ray buffer holds fewer than n rays");
    if(hits->bytes < n * sizeof(pt_hit)) return fail(PT_ERR_INVALID, "pt_pass_intersect_noalloc: hit buffer holds fewer than n records");
    pt_ctx *ctx = p->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    // the traversal arrays must be current: building them allocates
    if(!p->mesh_level_valid || !p->scene_level_valid || !p->d_mesh_buf.p || !p->d_scene_buf.p || slot_changed(p, PT_SLOT_TRIANGLE) ||
       slot_changed(p, PT_SLOT_MESH_BVH) || slot_changed(p, PT_SLOT_INSTANCE_INFO) || slot_changed(p, PT_SLOT_SCENE_BVH))
        return fail(PT_ERR_STATE, "pt_pass_intersect_noalloc: the bound scene has not been laid out for traversal yet (or has changed); "
                                  "call pt_pass_intersect or pt_pass_draw once first");
    if(p->spill_depth > NOALLOC_SPILL_DEPTH)
        return fail(PT_ERR_UNSUPPORTED, "pt_pass_intersect_noalloc: this scene's trees need " + std::to_string(p->spill_depth) +
                                            " spill entries per ray, the scratch array holds " + std::to_string(NOALLOC_SPILL_DEPTH));
    if(!ctx->scratch) return fail(PT_ERR_STATE, "pt_pass_intersect_noalloc: no scratch address for this device");
    hits->version++;
    hits->shadow_valid = false;
    if(n == 0) return PT_OK;
    JOIN_TRY(ctx);
    char *base = static_cast<char *>(ctx->scratch);  // the device address, never the symbol's name
    unsigned *cnt = reinterpret_cast<unsigned *>(base);
    unsigned *redo = reinterpret_cast<unsigned *>(base + NOALLOC_REDO_OFFSET);
    void *spill = base + NOALLOC_SPILL_OFFSET;
    hipStream_t s = ctx->stream;
    // one user of this GPU's array at a time: behind the last call's launches when they went to another stream
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if(g_scratch_last_stream[ctx->device] && g_scratch_last_stream[ctx->device] != s)
        HIP_TRY(hipStreamWaitEvent(s, g_scratch_event[ctx->device], 0));
    // the array keeps the last call's contents: the counters must start at zero
    HIP_TRY(hipMemsetAsync(cnt, 0, NOALLOC_REDO_OFFSET, s));
    long long blocks = ((long long)n + TRACE_BLOCK - 1) / TRACE_BLOCK;
    if(blocks > NOALLOC_BLOCKS) blocks = NOALLOC_BLOCKS;
    const int redo_blocks = p->redo_blocks < NOALLOC_BLOCKS ? p->redo_blocks : NOALLOC_BLOCKS;
    const TraceScene sc = make_trace_scene(p, spill);
    if(use_ordered(p)) {
        launch_intersect_fast(s, make_fast_scene(p, spill), (int)blocks, static_cast<const float *>(rays->dptr), (unsigned)n,
                              static_cast<pt_hit *>(hits->dptr), cnt, redo, cnt + 1);
        launch_intersect(s, sc, redo_blocks, static_cast<const float *>(rays->dptr), (unsigned)n, redo, cnt + 1,
                         static_cast<pt_hit *>(hits->dptr), cnt + 2, static_cast<unsigned long long *>(p->d_stats.p) + 4);
    } else {
        launch_intersect(s, sc, (int)blocks, static_cast<const float *>(rays->dptr), (unsigned)n, nullptr, nullptr,
                         static_cast<pt_hit *>(hits->dptr), cnt, nullptr);
    }
    HIP_TRY(hipEventRecord(g_scratch_event[ctx->device], s));
    g_scratch_last_stream[ctx->device] = s;
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

// ------------------------------------------------------------------------------- post-processing
int pt_post_merge(pt_ctx *ctx, const pt_buf *direct, const pt_buf *indirect, const pt_buf *albedo, pt_buf *color) {
    if(!ctx || !direct || !indirect || !albedo || !color) return fail(PT_ERR_INVALID, "pt_post_merge: null argument");
    if(direct->ctx != ctx || indirect->ctx != ctx || albedo->ctx != ctx || color->ctx != ctx)
        return fail(PT_ERR_INVALID, "pt_post_merge: buffers belong to another context");
    if(direct->bytes != color->bytes || indirect->bytes != color->bytes || albedo->bytes != color->bytes || color->bytes % 12)
        return fail(PT_ERR_INVALID, "pt_post_merge: the four buffers must have the same size, 3 floats per pixel");
    if(color->bytes / 12 > 0x7FFFFFFFull) return fail(PT_ERR_INVALID, "pt_post_merge: too many pixels");
    HIP_TRY(hipSetDevice(ctx->device));
    JOIN_TRY(ctx);
    color->version++;
    color->shadow_valid = false;
    if(color->bytes)
        launch_post_merge(ctx->stream, (int)(color->bytes / 12), static_cast<const float *>(direct->dptr),
                          static_cast<const float *>(indirect->dptr), static_cast<const float *>(albedo->dptr),
                          static_cast<float *>(color->dptr));
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_post_static_blend(pt_ctx *ctx, uint32_t frame_counter, const pt_buf *color, pt_buf *history_color, pt_buf *history_moment) {
    if(!ctx || !color || !history_color || !history_moment) return fail(PT_ERR_INVALID, "pt_post_static_blend: null argument");
    if(color->ctx != ctx || history_color->ctx != ctx || history_moment->ctx != ctx)
        return fail(PT_ERR_INVALID, "pt_post_static_blend: buffers belong to another context");
    if(frame_counter == 0) return fail(PT_ERR_INVALID, "pt_post_static_blend: frame_counter starts at 1");
    if(color->bytes % 12 || history_color->bytes != color->bytes || history_moment->bytes != color->bytes / 12 * 8)
        return fail(PT_ERR_INVALID, "pt_post_static_blend: color and history_color hold 3 floats per pixel, history_moment 2");
    if(color->bytes / 12 > 0x7FFFFFFFull) return fail(PT_ERR_INVALID, "pt_post_static_blend: too many pixels");
    HIP_TRY(hipSetDevice(ctx->device));
    JOIN_TRY(ctx);
    for(pt_buf *b : {history_color, history_moment}) {
        b->version++;
        b->shadow_valid = false;
    }
    if(color->bytes)
        launch_post_static_blend(ctx->stream, (int)(color->bytes / 12), frame_counter, static_cast<const float *>(color->dptr),
                                 static_cast<float *>(history_color->dptr), static_cast<float *>(history_moment->dptr));
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_post_tonemap(pt_ctx *ctx, const pt_buf *color, pt_buf *out) {
    if(!ctx || !color || !out) return fail(PT_ERR_INVALID, "pt_post_tonemap: null argument");
    if(color->ctx != ctx || out->ctx != ctx) return fail(PT_ERR_INVALID, "pt_post_tonemap: buffers belong to another context");
    if(color->bytes != out->bytes || color->bytes % 4) return fail(PT_ERR_INVALID, "pt_post_tonemap: buffers must have the same size");
    HIP_TRY(hipSetDevice(ctx->device));
    JOIN_TRY(ctx);
    out->version++;
    out->shadow_valid = false;
    if(out->bytes) launch_post_tonemap(ctx->stream, out->bytes / 4, static_cast<const float *>(color->dptr), static_cast<float *>(out->dptr));
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

// ------------------------------------------------------------------------------- denoising chain
namespace {
struct BufSpec { const pt_buf *b; int channels; const char *name; };
// Common checks of the stencil passes: context, sizes, no output aliasing an input.
int check_images(pt_ctx *ctx, const char *fn, int w, int h, std::initializer_list<BufSpec> in, std::initializer_list<BufSpec> out) {
    if(!ctx) return fail(PT_ERR_INVALID, std::string(fn) + ": null context");
    if(w <= 0 || h <= 0 || (long long)w * h > (1ll << 28)) return fail(PT_ERR_INVALID, std::string(fn) + ": unsupported resolution");
    const size_t npix = (size_t)w * h;
    for(const auto *list : {&in, &out})
        for(const BufSpec &s : *list) {
            if(!s.b) return fail(PT_ERR_INVALID, std::string(fn) + ": " + s.name + " is null");
            if(s.b->ctx != ctx) return fail(PT_ERR_INVALID, std::string(fn) + ": " + s.name + " belongs to another context");
            if(s.b->bytes != npix * s.channels * sizeof(float))
                return fail(PT_ERR_INVALID, std::string(fn) + ": " + s.name + " must hold width*height*" + std::to_string(s.channels) + " floats");
        }
    for(const BufSpec &o : out)
        for(const BufSpec &i : in)
            if(o.b == i.b) return fail(PT_ERR_INVALID, std::string(fn) + ": " + o.name + " must not be the same buffer as " + i.name);
    return PT_OK;
}
const float *fptr(const pt_buf *b) { return static_cast<const float *>(b->dptr); }
float *fptr_out(pt_buf *b) {
    b->version++;
    b->shadow_valid = false;
    return static_cast<float *>(b->dptr);
}
}  // namespace

int pt_post_svgf_temporal(pt_ctx *ctx, int width, int height, int first_frame, const pt_buf *color, const pt_buf *normal,
                          const pt_buf *instance_index, const pt_buf *motion, pt_buf *out_color, pt_buf *out_moment,
                          pt_buf *out_num_samples, const pt_buf *history_color, const pt_buf *history_moment,
                          const pt_buf *history_num_samples, const pt_buf *history_normal,
                          const pt_buf *history_instance_index) {
    int rc = check_images(ctx, "pt_post_svgf_temporal", width, height,
                          {{color, 3, "color"}, {normal, 3, "normal"}, {instance_index, 1, "instance_index"}, {motion, 2, "motion"},
                           {history_color, 3, "history_color"}, {history_moment, 2, "history_moment"},
                           {history_num_samples, 1, "history_num_samples"}, {history_normal, 3, "history_normal"},
                           {history_instance_index, 1, "history_instance_index"}},
                          {{out_color, 3, "out_color"}, {out_moment, 2, "out_moment"}, {out_num_samples, 1, "out_num_samples"}});
    if(rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    JOIN_TRY(ctx);
    launch_svgf_temporal(ctx->stream, width, height, first_frame != 0, fptr(color), fptr(normal), fptr(instance_index), fptr(motion),
                         fptr_out(out_color), fptr_out(out_moment), fptr_out(out_num_samples), fptr(history_color),
                         fptr(history_moment), fptr(history_num_samples), fptr(history_normal), fptr(history_instance_index));
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_post_svgf_variance(pt_ctx *ctx, int width, int height, const pt_buf *moment, const pt_buf *normal, const pt_buf *depth,
                          const pt_buf *num_samples, pt_buf *out_variance) {
    int rc = check_images(ctx, "pt_post_svgf_variance", width, height,
                          {{moment, 2, "moment"}, {normal, 3, "normal"}, {depth, 1, "depth"}, {num_samples, 1, "num_samples"}},
                          {{out_variance, 1, "out_variance"}});
    if(rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    JOIN_TRY(ctx);
    launch_svgf_variance(ctx->stream, width, height, fptr(moment), fptr(normal), fptr(depth), fptr(num_samples), fptr_out(out_variance));
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_post_svgf_spatial(pt_ctx *ctx, int width, int height, int step, const pt_buf *color, const pt_buf *normal,
                         const pt_buf *depth, const pt_buf *variance, pt_buf *out_color) {
    int rc = check_images(ctx, "pt_post_svgf_spatial", width, height,
                          {{color, 3, "color"}, {normal, 3, "normal"}, {depth, 1, "depth"}, {variance, 1, "variance"}},
                          {{out_color, 3, "out_color"}});
    if(rc) return rc;
    if(step < 1) return fail(PT_ERR_INVALID, "pt_post_svgf_spatial: step must be at least 1");
    HIP_TRY(hipSetDevice(ctx->device));
    JOIN_TRY(ctx);
    launch_svgf_spatial(ctx->stream, width, height, step, fptr(color), fptr(normal), fptr(depth), fptr(variance), fptr_out(out_color));
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_post_taa(pt_ctx *ctx, int width, int height, const pt_buf *color, const pt_buf *history_color, const pt_buf *motion,
                const pt_buf *normal, const pt_buf *instance_index, const pt_buf *history_normal,
                const pt_buf *history_instance_index, pt_buf *out_color) {
    int rc = check_images(ctx, "pt_post_taa", width, height,
                          {{color, 3, "color"}, {history_color, 3, "history_color"}, {motion, 2, "motion"}, {normal, 3, "normal"},
                           {instance_index, 1, "instance_index"}, {history_normal, 3, "history_normal"},
                           {history_instance_index, 1, "history_instance_index"}},
                          {{out_color, 3, "out_color"}});
    if(rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    JOIN_TRY(ctx);
    launch_taa(ctx->stream, width, height, fptr(color), fptr(history_color), fptr(motion), fptr(normal), fptr(instance_index),
               fptr(history_normal), fptr(history_instance_index), fptr_out(out_color));
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_buf_save_as_image(const pt_buf *buf, int width, int height, int channel, const char *path) {
    if(!buf || !path) return fail(PT_ERR_INVALID, "pt_buf_save_as_image: null argument");
    if(width <= 0 || height <= 0 || (channel != 1 && channel != 3 && channel != 4))
        return fail(PT_ERR_INVALID, "pt_buf_save_as_image: width/height must be positive and channel 1, 3 or 4");
    if((size_t)width * height * channel * sizeof(float) != buf->bytes)
        return fail(PT_ERR_INVALID, "pt_buf_save_as_image: width * height * channel floats is not the buffer's size");
    std::vector<float> host((size_t)width * height * channel);
    int rc = pt_buf_download(buf, host.data(), buf->bytes);
    if(rc) return rc;
    std::vector<uint8_t> px;
    quantize_flip(host.data(), width, height, channel, px);
    std::string err;
    if(!write_png(path, width, height, channel, px.data(), err)) return fail(PT_ERR_INVALID, err);
    return PT_OK;
}

}  // extern "C"
