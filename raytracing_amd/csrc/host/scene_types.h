// Plain-old-data records of the seven scene buffers, byte-compatible with what the
// reference uploads as SSBOs (slot numbers from src/renderpass/Renderer.cpp:32-38,
// std430 layouts from shader/pathtracing.glsl:18-84).
#pragma once
#include <cstdint>

namespace rta {

// slot 2 -- src/instance/Triangle.h:15-27 (three vec3 vertices, three vec3 normals,
// three vec2 uvs; the shader reads it as float ver[9], normal[9], uv[6]).
struct TriangleRec {
    float ver[9];
    float normal[9];
    float uv[6];
};
static_assert(sizeof(TriangleRec) == 96, "triangle record is 96 bytes");

// slots 4 and 5 -- src/ResourceManager.h:29-35.
struct BVHNodeInfo {
    float aa[4];
    float bb[4];
    int32_t lsIndex;
    int32_t rsIndex;
    int32_t instanceIndex;
    int32_t triangleIndex;
};
static_assert(sizeof(BVHNodeInfo) == 48, "BVH node record is 48 bytes");

// slot 3 -- src/ResourceManager.h:23-28.
struct InstanceInfo {
    float world2local[16];  // column major
    int32_t materialPtr;
    int32_t meshIndex;
    int32_t emptyblock[10];
};
static_assert(sizeof(InstanceInfo) == 112, "instance record is 112 bytes");

// slot 6 -- src/ResourceManager.h:36-48.  type 0 = point (x,y,z = world position),
// type 1 = directional (x,y,z = direction the light's +Z axis points to).
struct LightInfo {
    int32_t type;
    float x, y, z;
    float r, g, b;
    float emptyblock;
};
static_assert(sizeof(LightInfo) == 32, "light record is 32 bytes");

}  // namespace rta
