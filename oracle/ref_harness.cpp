// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the
// product library.  Only tests/, bench.py's cpu_baseline leg and fixture generators
// may run the binary this file builds.
//
// Driver around the reference's OWN compiled sources.  The Makefile next to this
// file compiles, from where they lie under /root/reference (nothing is copied):
//   src/BVH.cpp  src/AABB.cpp  src/Intersection.cpp
//   src/Transform.cpp  src/instance/Instance.cpp  src/instance/Camera.cpp  (through ref_transform_unit.cpp)
//   include/hdrloader.cpp  include/stb_image.cpp
// and links them with this driver into oracle/_ref/ref_harness.  Everything
// numerical (SAH build, std::sort tie order, slab / Moller-Trumbore tests, RGBE
// component conversion, JPEG decode, glm::inverse) is executed by reference code; this file only
// moves bytes in and out and restates the two GL-entangled loops that cannot be
// compiled headless:
//   - Mesh::build_meshBVH            (src/instance/Mesh.cpp:10-22)
//   - ResourceManager::reload_meshes (src/ResourceManager.cpp:116-140, BFS flatten)
//   - ResourceManager::reload_scene / reload_sceneBVH / reload_instances
//     (src/ResourceManager.cpp:146-224: world boxes from eight transformed corners,
//     BFS flatten of the scene tree, world2local = glm::inverse)
//
// File protocol (little endian, tightly packed):
//   meshes file : i32 nmesh ; per mesh { i32 ntri ; f32 pos[ntri*9] }
//   bvh out     : i32 nnodes ; node[nnodes]{f32 aa[4]; f32 bb[4]; i32 ls, rs, inst, tri}
//                 ; per mesh { i32 siz ; i32 depth }
//   scene file  : i32 ninst ; per inst { f32 l2w[16] (column major) ; i32 mesh }
//   rays file   : i32 nrays ; f32 ray[nrays*6] (ox oy oz dx dy dz)
//   hits out    : per ray { i32 inst ; i32 tri (global) ; f32 t }   (-1,-1,1e18 = miss)
//   scene out   : i32 nnodes ; node[nnodes] (as above, inst = instance index at leaves) ;
//                 per inst { f32 world2local[16] } ; i32 siz ; i32 depth
#include "BVH.h"
#include "instance/Instance.h"
#include "instance/Mesh.h"
#include "hdrloader.h"
#include "stb_image.h"
#include "glm/glm.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <queue>
#include <map>
#include <string>
#include <vector>

struct FlatNode { float aa[4], bb[4]; int ls, rs, inst, tri; };

static std::vector<char> slurp(const char *path) {
    FILE *f = fopen(path, "rb");
    if(!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> b(n);
    if(n && fread(b.data(), 1, n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f);
    return b;
}

struct MeshSet {
    std::vector<std::vector<Triangle>> tris;
    std::vector<BVHNode*> roots;
    std::vector<int> triBase;
};

static void load_meshes(const char *path, MeshSet &ms) {
    std::vector<char> b = slurp(path);
    const char *p = b.data();
    int nmesh; memcpy(&nmesh, p, 4); p += 4;
    ms.tris.resize(nmesh);
    int base = 0;
    for(int m = 0; m < nmesh; m++) {
        int ntri; memcpy(&ntri, p, 4); p += 4;
        ms.tris[m].resize(ntri);
        for(int i = 0; i < ntri; i++) {
            float v[9]; memcpy(v, p, 36); p += 36;
            for(int k = 0; k < 3; k++) ms.tris[m][i].vertex[k] = vec3(v[k*3], v[k*3+1], v[k*3+2]);
        }
        ms.triBase.push_back(base);
        base += ntri;
    }
    for(int m = 0; m < nmesh; m++) {
        // Mesh::build_meshBVH, src/instance/Mesh.cpp:10-22
        std::vector<BVHPrimitive> primitives;
        for(Triangle &t: ms.tris[m]) {
            BVHPrimitive tmp;
            tmp.trianglePtr = &t;
            for(auto &v: t.vertex) tmp.aabb.addPoint(v);
            primitives.push_back(tmp);
        }
        ms.roots.push_back(BVHNode::build(primitives));   // reference code
    }
}

static int cmd_bvh(const char *in, const char *out) {
    MeshSet ms; load_meshes(in, ms);
    // ResourceManager::reload_meshes, src/ResourceManager.cpp:116-140
    std::vector<FlatNode> flat;
    std::queue<BVHNode*> q;
    for(auto r: ms.roots) q.push(r);
    int N = (int)ms.roots.size();
    // triangle pointer -> global index (triangleIndexMap)
    std::map<Triangle*, int> triIndex;
    for(size_t m = 0; m < ms.tris.size(); m++)
        for(size_t i = 0; i < ms.tris[m].size(); i++) triIndex[&ms.tris[m][i]] = ms.triBase[m] + (int)i;
    while(!q.empty()) {
        BVHNode *p = q.front(); q.pop();
        int il = -1, ir = -1;
        if(p->ls) il = N++, q.emplace(p->ls);
        if(p->rs) ir = N++, q.emplace(p->rs);
        FlatNode y;
        y.aa[0] = p->aabb.mi.x; y.aa[1] = p->aabb.mi.y; y.aa[2] = p->aabb.mi.z; y.aa[3] = 0;
        y.bb[0] = p->aabb.mx.x; y.bb[1] = p->aabb.mx.y; y.bb[2] = p->aabb.mx.z; y.bb[3] = 0;
        y.ls = il; y.rs = ir; y.inst = -1; y.tri = -1;
        if(p->trianglePtr) y.tri = triIndex[p->trianglePtr];
        flat.push_back(y);
    }
    FILE *f = fopen(out, "wb");
    int n = (int)flat.size();
    fwrite(&n, 4, 1, f);
    fwrite(flat.data(), sizeof(FlatNode), flat.size(), f);
    for(auto r: ms.roots) { fwrite(&r->siz, 4, 1, f); fwrite(&r->depth, 4, 1, f); }
    fclose(f);
    return 0;
}

// CPU picking path at mesh level: BVHNode::rayIntersect (src/BVH.cpp:97-114) entered at every mesh root in
// turn, with the three statements of its scene-leaf branch (BVH.cpp:91-93: w2l = inverse(l2w), ray.ori =
// w2l * vec4(ori, 1), ray.dir = w2l * vec4(dir, 0)) executed here on a matrix read from the scene file.
// This form takes ANY local-to-world matrix (sheared, singular ...); cmd_isect_scene below runs the whole
// routine from the scene root, scene-leaf branch included, for instances whose matrices come from the
// reference's own Transform / Instance objects.
static int cmd_isect(const char *meshes, const char *scene, const char *rays, const char *out) {
    MeshSet ms; load_meshes(meshes, ms);
    std::map<Triangle*, int> triIndex;
    for(size_t m = 0; m < ms.tris.size(); m++)
        for(size_t i = 0; i < ms.tris[m].size(); i++) triIndex[&ms.tris[m][i]] = ms.triBase[m] + (int)i;
    std::vector<char> sb = slurp(scene);
    const char *p = sb.data();
    int ninst; memcpy(&ninst, p, 4); p += 4;
    std::vector<glm::mat4> w2l(ninst); std::vector<int> meshOf(ninst);
    for(int i = 0; i < ninst; i++) {
        glm::mat4 l2w; memcpy(&l2w[0][0], p, 64); p += 64;
        memcpy(&meshOf[i], p, 4); p += 4;
        w2l[i] = glm::inverse(l2w);
    }
    std::vector<char> rb = slurp(rays);
    int nrays; memcpy(&nrays, rb.data(), 4);
    const float *r = (const float*)(rb.data() + 4);
    FILE *f = fopen(out, "wb");
    for(int k = 0; k < nrays; k++) {
        vec3 ori(r[k*6], r[k*6+1], r[k*6+2]), dir(r[k*6+3], r[k*6+4], r[k*6+5]);
        Intersection isect;
        int hitInst = -1;
        for(int i = 0; i < ninst; i++) {
            vec3 o = w2l[i] * vec4(ori, 1);
            vec3 d = w2l[i] * vec4(dir, 0);
            Ray ray(o, d);
            Triangle *before = isect.trianglePtr;
            float tb = isect.t;
            ms.roots[meshOf[i]]->rayIntersect(ray, isect, nullptr);
            if(isect.trianglePtr != before || isect.t != tb) hitInst = i;
        }
        int tri = isect.exist ? triIndex[isect.trianglePtr] : -1;
        if(!isect.exist) hitInst = -1;
        // the same Triangle object is shared by every instance of a mesh: report the
        // global index of the triangle inside the mesh array, as the flat buffers do
        float t = isect.t;
        fwrite(&hitInst, 4, 1, f); fwrite(&tri, 4, 1, f); fwrite(&t, 4, 1, f);
    }
    fclose(f);
    return 0;
}

// The CPU picking path as the reference runs it (mouse_clickcalback, main.cpp:125-146): a hierarchy of real
// Instance objects with real Transforms (src/instance/Instance.cpp, src/Transform.cpp, compiled in
// ref_transform_unit.cpp), the scene BVH made by the statements of ResourceManager::reload_scene /
// update_globalinstance_recursive (src/ResourceManager.cpp:47-77, 200-221) over BVHNode::build, and
// BVHNode::rayIntersect (src/BVH.cpp:89-114) entered at the SCENE root: its scene-leaf branch
// (inverse(instancePtr->matrix_to_global()), the descent into the mesh tree) runs as written.  A Mesh cannot be
// constructed headless (its constructor creates GL objects): rayIntersect and build read meshBVHRoot only, so
// a Mesh is zeroed storage of the reference's class layout with that member set.
//   nodes file : i32 nnodes ; per node { f32 position[3], rotation_deg[3], scale[3] ; i32 mesh (-1: none) ;
//                i32 parent (-1: child of the scene root, else the index of an EARLIER node) }
//                instances = the nodes that carry a mesh, in hierarchy pre-order (children in file order)
//   out        : i32 ninst ; per ray { i32 inst ; i32 tri (global) ; f32 t } ;
//                per instance { f32 transform2world[16] ; i32 mesh }   (the matrix reload_instances inverts)
static int cmd_isect_scene(const char *meshes, const char *nodes_path, const char *rays, const char *out) {
    MeshSet ms; load_meshes(meshes, ms);
    std::map<Triangle*, int> triIndex;
    for(size_t m = 0; m < ms.tris.size(); m++)
        for(size_t i = 0; i < ms.tris[m].size(); i++) triIndex[&ms.tris[m][i]] = ms.triBase[m] + (int)i;
    std::vector<char> nb = slurp(nodes_path);
    const char *p = nb.data();
    int nnodes; memcpy(&nnodes, p, 4); p += 4;
    Instance *scene = new Instance("scene");
    std::vector<Instance*> node(nnodes);
    std::map<Mesh*, int> meshIndexOf;
    for(int k = 0; k < nnodes; k++) {
        float prs[9]; int mesh, parent;
        memcpy(prs, p, 36); p += 36;
        memcpy(&mesh, p, 4); p += 4;
        memcpy(&parent, p, 4); p += 4;
        if(parent >= k || mesh >= (int)ms.roots.size()) { fprintf(stderr, "bad nodes file\n"); return 2; }
        Instance *ins = new Instance("node");
        ins->transform = Transform(vec3(prs[0], prs[1], prs[2]), vec3(prs[3], prs[4], prs[5]), vec3(prs[6], prs[7], prs[8]));
        if(mesh >= 0) {
            Mesh *m = static_cast<Mesh*>(calloc(1, sizeof(Mesh)));
            m->meshBVHRoot = ms.roots[mesh];
            meshIndexOf[m] = mesh;
            ins->mesh = m;
        }
        (parent < 0 ? scene : node[parent])->add_child(ins);
        node[k] = ins;
    }
    // ResourceManager::update_globalinstance_recursive, src/ResourceManager.cpp:58-77 (without the lights)
    std::vector<std::pair<Instance*, mat4>> globalInstances;
    std::map<Instance*, int> instanceIndexMap;
    struct Walk {
        std::vector<std::pair<Instance*, mat4>> &gi;
        std::map<Instance*, int> &im;
        void run(Instance *cur, mat4 transform2world) {
            if(cur->mesh) {
                im[cur] = (int)gi.size();
                gi.emplace_back(cur, transform2world);
            }
            Instance *child;
            for(int i = 0; (child = cur->get_child(i)) != nullptr; i++) run(child, transform2world * child->transform.matrix());
        }
    } walk{globalInstances, instanceIndexMap};
    walk.run(scene, mat4(1.0));
    // ResourceManager::reload_scene, src/ResourceManager.cpp:203-220
    std::vector<BVHPrimitive> primitives;
    for(auto &[u, mat]: globalInstances) {
        BVHPrimitive prim;
        prim.instancePtr = u;
        AABB cube = u->mesh->meshBVHRoot->aabb;
        for(int i = 0;i < 8;i++) {
            vec3 point = {
                    (i & 1) ? cube.mi[0] : cube.mx[0],
                    (i & 2) ? cube.mi[1] : cube.mx[1],
                    (i & 4) ? cube.mi[2] : cube.mx[2]
            };
            point = mat * vec4(point, 1);
            prim.aabb.addPoint(point);
        }
        primitives.push_back(prim);
    }
    BVHNode *root = BVHNode::build(primitives);   // reference code
    std::vector<char> rb = slurp(rays);
    int nrays; memcpy(&nrays, rb.data(), 4);
    const float *r = (const float*)(rb.data() + 4);
    FILE *f = fopen(out, "wb");
    int ninst = (int)globalInstances.size();
    fwrite(&ninst, 4, 1, f);
    for(int k = 0; k < nrays; k++) {
        vec3 ori(r[k*6], r[k*6+1], r[k*6+2]), dir(r[k*6+3], r[k*6+4], r[k*6+5]);
        Ray ray(ori, dir);
        Intersection isect;
        root->rayIntersect(ray, isect, nullptr);   // reference code, from the scene root (main.cpp:135-137)
        int inst = isect.exist ? instanceIndexMap[isect.instancePtr] : -1;
        int tri = isect.exist ? triIndex[isect.trianglePtr] : -1;
        float t = isect.t;
        fwrite(&inst, 4, 1, f); fwrite(&tri, 4, 1, f); fwrite(&t, 4, 1, f);
    }
    for(auto &[u, mat]: globalInstances) {
        fwrite(&mat[0][0], 4, 16, f);
        int mi = meshIndexOf[u->mesh];
        fwrite(&mi, 4, 1, f);
    }
    fclose(f);
    return 0;
}

// Scene level: the statements of ResourceManager::reload_scene (src/ResourceManager.cpp:200-221) over the
// reference's own AABB, glm and BVHNode::build, then the flatten of reload_sceneBVH (:166-198) and the
// world2local of reload_instances (:154).  BVHNode::build reads instancePtr->mesh->meshBVHRoot of a
// one-primitive node (src/BVH.cpp:19-24) and nothing else of either object; neither can be constructed
// headless (their constructors live in GL-entangled translation units), so both are zeroed storage of the
// reference's own class layouts (instance/Instance.h, instance/Mesh.h) with those two members set.
static int cmd_scene(const char *meshes, const char *scene, const char *out) {
    MeshSet ms; load_meshes(meshes, ms);
    std::vector<char> sb = slurp(scene);
    const char *p = sb.data();
    int ninst; memcpy(&ninst, p, 4); p += 4;
    std::vector<glm::mat4> mats(ninst); std::vector<int> meshOf(ninst);
    std::vector<BVHPrimitive> primitives;
    std::map<Instance*, int> instanceIndex;
    for(int k = 0; k < ninst; k++) {
        memcpy(&mats[k][0][0], p, 64); p += 64;
        memcpy(&meshOf[k], p, 4); p += 4;
        const glm::mat4 &mat = mats[k];
        BVHPrimitive prim;
        Mesh *mesh = static_cast<Mesh*>(calloc(1, sizeof(Mesh)));
        mesh->meshBVHRoot = ms.roots[meshOf[k]];
        Instance *ins = static_cast<Instance*>(calloc(1, sizeof(Instance)));
        ins->mesh = mesh;
        instanceIndex[ins] = k;
        prim.instancePtr = ins;
        AABB cube = ms.roots[meshOf[k]]->aabb;
        for(int i = 0;i < 8;i++) {
            vec3 point = {
                    (i & 1) ? cube.mi[0] : cube.mx[0],
                    (i & 2) ? cube.mi[1] : cube.mx[1],
                    (i & 4) ? cube.mi[2] : cube.mx[2]
            };
            point = mat * vec4(point, 1);
            prim.aabb.addPoint(point);
        }
        primitives.push_back(prim);
    }
    BVHNode *root = BVHNode::build(primitives);   // reference code
    std::vector<FlatNode> flat;
    std::queue<BVHNode*> q;
    q.emplace(root);
    int N = 0;
    while(!q.empty()) {
        BVHNode *n = q.front(); q.pop();
        FlatNode y;
        y.aa[0] = n->aabb.mi.x; y.aa[1] = n->aabb.mi.y; y.aa[2] = n->aabb.mi.z; y.aa[3] = 0;
        y.bb[0] = n->aabb.mx.x; y.bb[1] = n->aabb.mx.y; y.bb[2] = n->aabb.mx.z; y.bb[3] = 0;
        y.ls = -1; y.rs = -1; y.inst = -1; y.tri = -1;   // BVHNodeInfo's defaults, src/ResourceManager.h
        if(n->instancePtr) {
            y.inst = instanceIndex[n->instancePtr];
            flat.push_back(y);
            continue;
        }
        if(n->ls) y.ls = ++N, q.emplace(n->ls);
        if(n->rs) y.rs = ++N, q.emplace(n->rs);
        flat.push_back(y);
    }
    FILE *f = fopen(out, "wb");
    int n = (int)flat.size();
    fwrite(&n, 4, 1, f);
    fwrite(flat.data(), sizeof(FlatNode), flat.size(), f);
    for(int k = 0; k < ninst; k++) {
        glm::mat4 w2l = glm::inverse(mats[k]);
        fwrite(&w2l[0][0], 4, 16, f);
    }
    fwrite(&root->siz, 4, 1, f); fwrite(&root->depth, 4, 1, f);
    fclose(f);
    return 0;
}

// glm::inverse / glm mat4*mat4 as compiled from the vendored headers.
//   in : i32 n ; f32 m[n*16]      out: f32 inv[n*16] ; f32 prod[16] (m0*m1*...*m(n-1))
static int cmd_mat(const char *in, const char *out) {
    std::vector<char> b = slurp(in);
    int n; memcpy(&n, b.data(), 4);
    const float *m = (const float*)(b.data() + 4);
    FILE *f = fopen(out, "wb");
    glm::mat4 prod(1.0f);
    for(int i = 0; i < n; i++) {
        glm::mat4 a; memcpy(&a[0][0], m + i*16, 64);
        glm::mat4 inv = glm::inverse(a);
        fwrite(&inv[0][0], 4, 16, f);
        prod = prod * a;
    }
    fwrite(&prod[0][0], 4, 16, f);
    fclose(f);
    return 0;
}

// HDRLoader::load itself cannot be driven on LP64: it reads the resolution with
// sscanf("%ld") into two ints (include/hdrloader.cpp:67-68), which overwrites one of them.
// Its arithmetic lives in convertComponent(expo, val) (hdrloader.cpp:99-104), which has
// external linkage; dump it for every exponent byte and mantissa byte.
//   out: f32 table[256 exponent bytes][256 mantissa bytes]  (expo = byte - 128)
float convertComponent(int expo, int val);
static int cmd_hdr(const char *out) {
    FILE *f = fopen(out, "wb");
    for(int e = 0; e < 256; e++)
        for(int v = 0; v < 256; v++) {
            float x = convertComponent(e - 128, v);
            fwrite(&x, 4, 1, f);
        }
    fclose(f);
    return 0;
}

// stbi_load_from_memory exactly as loader.cpp:23-27 calls it (no vertical flip,
// channels as stored).   out: i32 w, h, c ; u8 data[w*h*c]
static int cmd_stb(const char *in, const char *out) {
    std::vector<char> b = slurp(in);
    int w, h, c;
    unsigned char *d = stbi_load_from_memory((unsigned char*)b.data(), (int)b.size(), &w, &h, &c, 0);
    if(!d) { fprintf(stderr, "stbi failed\n"); return 1; }
    FILE *f = fopen(out, "wb");
    fwrite(&w, 4, 1, f); fwrite(&h, 4, 1, f); fwrite(&c, 4, 1, f);
    fwrite(d, 1, (size_t)w * h * c, f);
    fclose(f);
    stbi_image_free(d);
    return 0;
}

// Defined in ref_transform_unit.cpp, which compiles src/Transform.cpp, src/instance/Instance.cpp
// and src/instance/Camera.cpp.
void ref_transform_matrix(const float *prs9, float *out16);
void ref_camera_matrices(float fov_x, float far_plane, const float *parent9, const float *cam9, float *out64,
                         int *screen_wh_out);

// transforms file : i32 n ; per item { f32 fov_x, far ; f32 parent[9] ; f32 cam[9] }   (9 = position, rotation in
//                   degrees, scale)
// out             : i32 screen_w, screen_h ; per item { f32 parent_matrix[16] ; f32 cam_matrix[16] ;
//                   f32 cam_to_global[16] ; f32 w2v[16] ; f32 v2w[16] ; f32 projection[16] }
static int cmd_transform(const char *in, const char *out) {
    FILE *f = fopen(in, "rb");
    if(!f) return 2;
    int n = 0;
    if(fread(&n, 4, 1, f) != 1) return 2;
    std::vector<float> rec((size_t)n * 20);
    if(fread(rec.data(), 4, rec.size(), f) != rec.size()) return 2;
    fclose(f);
    std::vector<float> res((size_t)n * 96);
    int wh[2] = {0, 0};
    for(int i = 0; i < n; i++) {
        const float *r = &rec[(size_t)i * 20];
        float *o = &res[(size_t)i * 96];
        ref_transform_matrix(r + 2, o);
        ref_transform_matrix(r + 11, o + 16);
        ref_camera_matrices(r[0], r[1], r + 2, r + 11, o + 32, wh);
    }
    f = fopen(out, "wb");
    fwrite(wh, 4, 2, f);
    fwrite(res.data(), 4, res.size(), f);
    fclose(f);
    return 0;
}

int main(int argc, char **argv) {
    std::string c = argc > 1 ? argv[1] : "";
    if(c == "transform" && argc == 4) return cmd_transform(argv[2], argv[3]);
    if(c == "bvh" && argc == 4) return cmd_bvh(argv[2], argv[3]);
    if(c == "isect" && argc == 6) return cmd_isect(argv[2], argv[3], argv[4], argv[5]);
    if(c == "scene" && argc == 5) return cmd_scene(argv[2], argv[3], argv[4]);
    if(c == "isect_scene" && argc == 6) return cmd_isect_scene(argv[2], argv[3], argv[4], argv[5]);
    if(c == "mat" && argc == 4) return cmd_mat(argv[2], argv[3]);
    if(c == "hdr" && argc == 3) return cmd_hdr(argv[2]);
    if(c == "stb" && argc == 4) return cmd_stb(argv[2], argv[3]);
    fprintf(stderr, "usage: ref_harness bvh <meshes> <out> | isect <meshes> <scene> <rays> <out> | isect_scene <meshes> <nodes> <rays> <out> | "
                    "scene <meshes> <scene> <out> | "
                    "mat <in> <out> | hdr <out> | stb <image> <out>\n");
    return 2;
}
