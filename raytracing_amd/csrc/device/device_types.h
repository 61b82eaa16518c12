// Records shared by the host re-layout code and the HIP kernels.
#pragma once
#include <cstdint>

namespace rta {

// ---- traversal layout ---------------------------------------------------------------
// The reference's 48-byte node holds its OWN box and two child indices, so visiting a
// node costs one dependent fetch per child before anything can be culled.  Here an
// inner node carries the boxes of BOTH children in one 64-byte, 64-byte-aligned record
// (half a 128-byte L2 line, four 16-byte loads per lane), and a child that is a leaf is
// referenced directly, so leaves need no node fetch at all.
//
// All traversal records are 64 bytes (an instance takes two) and live in two device buffers:
//   mesh buffer   one slot per node of the reference's mesh BVHs, TNode (inner) or TLeaf (leaf),
//                 each tree in depth-first order; then the wide nodes        follows slots 2 and 4
//   scene buffer  [ TNode x scene inner nodes | InstAccel x instances | wide nodes ]  follows slots 3 and 5
// The TNodes serve the reference-order traversal, the wide nodes (the same trees, four-wide:
// QNodes, or WNodes on request) the ordered one; both share the leaves and the
// instance records.
// so a step fetches base + 64 * slot whatever it is about to visit.
// Child reference: bits 31..30 = kind, bits 29..0 = 64-byte slot in the kind's buffer.
enum : uint32_t {
    REF_MESH_INNER = 0u,   // mesh buffer, a TNode
    REF_TRIANGLE = 1u,     // mesh buffer, a TLeaf (which names its triangle)
    REF_SCENE_INNER = 2u,  // scene buffer, a TNode
    REF_INSTANCE = 3u,     // scene buffer, an InstAccel: instance index = (slot - inst_base) / 2
    REF_INDEX_MASK = 0x3FFFFFFFu,
    // In a wide node, a triangle reference with this bit names TWO leaves, the TLeaf at the slot
    // and the one behind it: the two children of one reference node (they share its box and take
    // one child slot; that node itself never becomes a wide node).  The second is tested in the
    // step after the first.  Mesh-buffer slots therefore stay below 2^29.
    REF_PAIR = 0x20000000u,
    REF_SLOT_MASK = 0x1FFFFFFFu
};
inline uint32_t make_ref(uint32_t kind, uint32_t index) { return (kind << 30) | index; }

struct alignas(64) TNode {
    float lo0[3], hi0[3];  // box of child 0 (the reference's ls)
    float lo1[3], hi1[3];  // box of child 1 (the reference's rs)
    uint32_t c0, c1;       // child references
    uint32_t pad[2];
};
static_assert(sizeof(TNode) == 64, "traversal node is 64 bytes");

// Four-wide node of the ordered traversal, 128 bytes, 128-byte aligned (an even slot): up to
// four subtrees of the reference's binary tree, gathered by dissolving inner nodes below it
// (which ones: planned on the host so that a ray visits as few wide nodes as possible).  The first half reads like a TNode with two more
// references.  A child that is a triangle carries the box of its PARENT in the reference tree
// (the reference tests no box of the leaf itself); an absent child is REF_NONE.
constexpr uint32_t REF_NONE = 0xFFFFFFFFu;
struct alignas(64) WNode {  // (placed at even slots, so 128-byte aligned in the buffers)
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    uint32_t c[4];
    float lo2[3], hi2[3];
    float lo3[3], hi3[3];
    uint32_t pad[4];
};
static_assert(sizeof(WNode) == 128, "wide traversal node is 128 bytes");

// The same four children in 64 bytes (the default: half the loads per visit outweighs the
// decoding): child boxes on a grid of 256 steps per axis laid over the node's box (lo <= hi on
// every axis), rounded outwards, so that decoding (fma(step, scale, origin), one rounding)
// never yields a box smaller than the exact one.  scale is a power of two; an infinite scale
// makes every plane of that axis decode to NaN, which the slab test ignores: every ray passes
// (used when a coordinate is not finite or the extent exceeds the grid).
struct alignas(64) QNode {
    float origin[3];
    float scale[3];
    uint32_t c[4];
    uint8_t qlo[3][4];  // [axis][child]
    uint8_t qhi[3][4];
};
static_assert(sizeof(QNode) == 64, "quantised wide node is 64 bytes");

// Triangle leaf, 64 bytes: the exact fp32 box of the leaf's PARENT node in the reference tree
// (the last box the reference tests before it tests the triangle; +-inf when the triangle is
// the whole mesh; read by the ordered traversal only) followed by vertex 0 and the two edges
// v1 - v0, v2 - v0: the same fp32 differences the intersection routine would form itself.
struct alignas(64) TLeaf {
    float parent_lo[3], parent_hi[3];
    float v0[3];
    float e1[3];
    float e2[3];
    uint32_t triangle;  // index into the triangle buffer
};
static_assert(sizeof(TLeaf) == 64, "triangle leaf record is 64 bytes");

// One triangle of one instance in the flattened tree (trace_flat.h), 128 bytes, at an even slot: the TLeaf of
// the triangle (box of its reference parent and the triangle, both in the instance's space), then the rows of
// the instance's world-to-local matrix as in InstAccel, then the instance index -- everything a ray needs to
// run the reference's test on this triangle arrives with one fetch.
struct alignas(64) FlatLeaf {
    float parent_lo[3], parent_hi[3];
    float v0[3];
    float e1[3];
    float e2[3];
    uint32_t triangle;
    float row[3][4];
    uint32_t instance;
    uint32_t pad[3];
};
static_assert(sizeof(FlatLeaf) == 128, "flattened leaf record is 128 bytes");

// Per-instance record, 128 bytes.  row[r] = (m[0][r], m[1][r], m[2][r], m[3][r]) of the
// column-major world-to-local matrix, i.e. local = row . (world, 1).  The first 64 bytes are
// all a ray needs to enter the instance.
struct alignas(64) InstAccel {
    float row[3][4];
    uint32_t root_ref;               // mesh root: inner node (a TNode) or, for a 1-triangle mesh, the triangle
    int32_t material_ptr;
    uint32_t wide_root_ref;          // the same root for the ordered traversal (a WNode or the triangle)
    uint32_t material_type;          // the type word of its material record, rounded (1 conductor, 2 dielectric): the
                                     // shade launch groups paths by it before it assigns them to lanes
    float root_lo[3], root_hi[3];    // box of the mesh root (tested in local space)
    float world_lo[3], world_hi[3];  // box of the instance's scene-level leaf (tested in world space)
    uint32_t pad1[4];
};
static_assert(sizeof(InstAccel) == 128, "instance record is 128 bytes");

struct SceneHeader {
    float root_lo[3], root_hi[3];  // box of scene node 0
    uint32_t root_ref;             // scene inner node 0, or the only instance
    uint32_t wide_root_ref;        // the same for the ordered traversal (a WNode / QNode of the scene buffer)
    uint32_t quantised;            // 1: the wide nodes of both buffers are QNodes
    uint32_t n_instances;
    float extent;                  // |x| + |y| + |z| extent of the scene box (scales absolute tolerances)
    uint32_t inst_base;            // first InstAccel slot of the scene buffer
};

// ---- closest-hit record (matches pt_hit in the public header, padded to 32 B in the
//      path-state arrays) ---------------------------------------------------------------
struct HitRec {
    int32_t instance;
    int32_t triangle;
    float t, u, v;
};

struct alignas(16) PathHit {
    float t, u, v;
    int32_t triangle;
    int32_t instance;
    int32_t pad[3];
};
static_assert(sizeof(PathHit) == 32, "path hit record is 32 bytes");

struct alignas(16) PathRay {
    float o[3];
    float tmax2;  // shadow rays towards point lights: squared distance limit; unused otherwise
    float d[3];
    uint32_t flags;
};
static_assert(sizeof(PathRay) == 32, "path ray record is 32 bytes");

// throughput, pending BSDF-sample weight and sampler state of one path
struct alignas(16) PathThroughput {
    float history[3];
    float pdf;        // pdf of the BSDF-sampled direction the current ray follows
    float weight[3];  // fr * |cos| of that sample
    uint32_t dim;     // next Sobol dimension
};
static_assert(sizeof(PathThroughput) == 32, "path throughput record is 32 bytes");

struct alignas(16) PathAccum {
    float di[3];
    float first_exist;  // 1 when the primary ray hit something
    float gi[3];
    float pad0;
    float albedo[3];
    float pad1;
};
static_assert(sizeof(PathAccum) == 48, "path accumulator record is 48 bytes");

// One pending light sample (sky NEE or an analytic light) of one path vertex.
struct alignas(16) ShadowEntry {
    float d[3];
    float dis2;      // occluders beyond sqrt(dis2) do not count (1e18 for sky / directional)
    float recv[3];   // radiance * fr * |cos| / pdf, before the path throughput
    uint32_t flags;  // bit0 valid, bit1 path vertex is the first one (also sets direct light),
                     // bit2 visible (written by the shadow stage)
};
static_assert(sizeof(ShadowEntry) == 32, "shadow entry is 32 bytes");

struct TexDesc {
    const void *data;
    int32_t width, height, channels, format;  // format 0 = u8 unorm, 1 = f32
};

}  // namespace rta
