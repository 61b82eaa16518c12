// glTF 2.0 binary (.glb) reader producing what the reference's AssimpLoader::load_model
// (src/tool/loader.cpp:44-158) hands to its scene classes, as plain arrays:
//   - one mesh per glTF mesh (the reference asserts one mesh per node, loader.cpp:116):
//     triangles in index order, 96-byte records (positions, normals, uvs as stored: assimp's
//     own V flip and aiProcess_FlipUVs cancel, loader.cpp:139-142; no TEXCOORD_0 -> zeros,
//     loader.cpp:83-87);
//   - every material becomes a RoughConductor (loader.cpp:46): albedo = baseColorFactor rgb,
//     roughness = max(0.001, roughnessFactor), metallic = 1.0 because a glTF import always
//     carries the metallic key (loader.cpp:53-61), albedo map = the baseColorTexture's image,
//     one texture object per material that names one (loader.cpp:63-65);
//   - instances = the nodes that carry a mesh, in hierarchy pre-order
//     (ResourceManager.cpp:47-77), with their accumulated local-to-world matrix; the root
//     node's own transform is reset to identity (loader.cpp:152-156).
// The importer library itself (assimp) is not available; what it would do beyond the above
// (normal generation, quaternion node rotations, multi-primitive meshes, multi-root scenes)
// is refused with a message rather than guessed.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "scene_types.h"
#include "vecmath.h"

namespace rta {

struct GlbMaterial {
    float albedo[3];
    float roughness;
    float metallic;
    int32_t texture;  // index into GlbScene::textures, -1 = none
};

struct GlbTexture {
    size_t offset = 0, bytes = 0;  // the encoded image (JPEG / PNG ...) inside the file's binary chunk
    std::string mime;
    int32_t image = -1;            // glTF image index
};

struct GlbInstance {
    int32_t mesh;
    int32_t material;  // -1 = the node's mesh names no material
    Mat4 l2w;
};

struct GlbScene {
    std::vector<uint8_t> file;            // the whole file (textures point into it)
    size_t bin_offset = 0, bin_bytes = 0;
    std::vector<int32_t> tri_count;       // per mesh
    std::vector<TriangleRec> triangles;   // meshes back to back
    std::vector<GlbMaterial> materials;
    std::vector<GlbTexture> textures;
    std::vector<GlbInstance> instances;
};

// false + err on anything malformed or unsupported; every index and byte range is checked.
bool load_glb(const uint8_t *data, size_t n, GlbScene &out, std::string &err);

}  // namespace rta
