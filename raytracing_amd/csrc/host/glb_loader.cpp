// See glb_loader.h.  A small JSON reader (values kept as a tree) and the glTF walk.
#include "glb_loader.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include "scene_host.h"

namespace rta {
namespace {

// ---- JSON ------------------------------------------------------------------------------
struct JVal {
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    double num = 0.0;
    bool b = false;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal *get(const char *key) const {
        if(kind != OBJ) return nullptr;
        for(const auto &kv : obj)
            if(kv.first == key) return &kv.second;
        return nullptr;
    }
    bool has(const char *key) const { return get(key) != nullptr; }
    size_t size() const { return kind == ARR ? arr.size() : 0; }
};

struct JParser {
    const char *p, *end;
    std::string err;
    int depth = 0;
    void ws() {
        while(p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
    }
    bool fail(const char *m) {
        if(err.empty()) err = m;
        return false;
    }
    bool string(std::string &out) {
        if(p >= end || *p != '"') return fail("JSON: string expected");
        p++;
        out.clear();
        while(p < end && *p != '"') {
            if(*p == '\\') {
                if(++p >= end) return fail("JSON: truncated escape");
                switch(*p) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if(end - p < 5) return fail("JSON: truncated \\u escape");
                        unsigned cp = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                        p += 4;
                        // (names only; encoded as UTF-8 without surrogate pairing)
                        if(cp < 0x80) out += (char)cp;
                        else if(cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += *p;
                }
                p++;
            } else {
                out += *p++;
            }
        }
        if(p >= end) return fail("JSON: unterminated string");
        p++;
        return true;
    }
    bool value(JVal &v) {
        if(++depth > 64) return fail("JSON: nested too deeply");
        ws();
        if(p >= end) return fail("JSON: unexpected end");
        bool ok = true;
        if(*p == '{') {
            v.kind = JVal::OBJ;
            p++;
            ws();
            if(p < end && *p == '}') {
                p++;
            } else {
                for(;;) {
                    ws();
                    std::string key;
                    if(!string(key)) return false;
                    ws();
                    if(p >= end || *p != ':') return fail("JSON: ':' expected");
                    p++;
                    v.obj.emplace_back(key, JVal());
                    if(!value(v.obj.back().second)) return false;
                    ws();
                    if(p < end && *p == ',') { p++; continue; }
                    if(p < end && *p == '}') { p++; break; }
                    return fail("JSON: ',' or '}' expected");
                }
            }
        } else if(*p == '[') {
            v.kind = JVal::ARR;
            p++;
            ws();
            if(p < end && *p == ']') {
                p++;
            } else {
                for(;;) {
                    v.arr.emplace_back();
                    if(!value(v.arr.back())) return false;
                    ws();
                    if(p < end && *p == ',') { p++; continue; }
                    if(p < end && *p == ']') { p++; break; }
                    return fail("JSON: ',' or ']' expected");
                }
            }
        } else if(*p == '"') {
            v.kind = JVal::STR;
            ok = string(v.str);
        } else if(end - p >= 4 && !strncmp(p, "true", 4)) {
            v.kind = JVal::BOOL; v.b = true; p += 4;
        } else if(end - p >= 5 && !strncmp(p, "false", 5)) {
            v.kind = JVal::BOOL; v.b = false; p += 5;
        } else if(end - p >= 4 && !strncmp(p, "null", 4)) {
            v.kind = JVal::NUL; p += 4;
        } else {
            // a number: strtod on a bounded, terminated copy
            const char *q = p;
            while(q < end && (strchr("+-.eE", *q) || (*q >= '0' && *q <= '9'))) q++;
            if(q == p) return fail("JSON: value expected");
            std::string tmp(p, q);
            char *e = nullptr;
            v.num = strtod(tmp.c_str(), &e);
            if(!e || *e) return fail("JSON: malformed number");
            v.kind = JVal::NUM;
            p = q;
        }
        depth--;
        return ok;
    }
};

// ---- glTF ------------------------------------------------------------------------------
struct Ctx {
    const JVal *js;
    const uint8_t *bin;
    size_t bin_bytes;
    std::string *err;
    bool fail(const std::string &m) {
        if(err->empty()) *err = "glb: " + m;
        return false;
    }
};

bool index_of(const JVal *v, size_t limit, size_t &out) {
    if(!v || v->kind != JVal::NUM || v->num < 0 || v->num != (double)(long long)v->num || (size_t)v->num >= limit) return false;
    out = (size_t)v->num;
    return true;
}

int comp_size(int t) {
    switch(t) {
        case 5120: case 5121: return 1;
        case 5122: case 5123: return 2;
        case 5125: case 5126: return 4;
    }
    return 0;
}

// One accessor as rows of `ncomp` doubles-free typed reads: out gets count * ncomp floats
// (as_index: values converted from the integer types instead).
bool read_accessor(Ctx &c, size_t idx, int want_ncomp, bool as_index, std::vector<float> *fout, std::vector<uint32_t> *iout,
                   size_t &count) {
    const JVal *accs = c.js->get("accessors");
    if(!accs || idx >= accs->size()) return c.fail("accessor index out of range");
    const JVal &a = accs->arr[idx];
    if(a.has("sparse")) return c.fail("sparse accessors are not supported");
    const JVal *type = a.get("type"), *ct = a.get("componentType"), *cnt = a.get("count");
    if(!type || type->kind != JVal::STR || !ct || ct->kind != JVal::NUM || !cnt || cnt->kind != JVal::NUM) return c.fail("malformed accessor");
    int ncomp = type->str == "SCALAR" ? 1 : type->str == "VEC2" ? 2 : type->str == "VEC3" ? 3 : type->str == "VEC4" ? 4 : 0;
    if(ncomp != want_ncomp) return c.fail("accessor has type " + type->str + ", expected " + std::to_string(want_ncomp) + " components");
    const int ctype = (int)ct->num, csz = comp_size(ctype);
    if(!csz) return c.fail("unknown accessor component type");
    if(as_index ? (ctype == 5126 || ctype == 5120 || ctype == 5122) : ctype != 5126)
        return c.fail(as_index ? "index accessor must hold unsigned integers" : "vertex attribute accessor must hold floats");
    if(cnt->num < 0 || cnt->num > 4e9) return c.fail("accessor count out of range");
    count = (size_t)cnt->num;
    const JVal *bvs = c.js->get("bufferViews");
    size_t bvi;
    if(!bvs || !index_of(a.get("bufferView"), bvs->size(), bvi)) return c.fail("accessor without a valid bufferView");
    const JVal &bv = bvs->arr[bvi];
    const JVal *bo = bv.get("byteOffset"), *bl = bv.get("byteLength"), *bs = bv.get("byteStride"), *ao = a.get("byteOffset"), *bb = bv.get("buffer");
    if(bb && (bb->kind != JVal::NUM || bb->num != 0)) return c.fail("only the embedded binary chunk (buffer 0) is supported");
    if(!bl || bl->kind != JVal::NUM) return c.fail("bufferView without byteLength");
    // Every quantity of the file is checked as the number it is in the JSON (a double) before it becomes a size_t:
    // whole, non-negative, and small enough that no product below can wrap.
    auto whole = [&](const JVal *v, double dflt, double &out_d) {
        if(!v) { out_d = dflt; return true; }
        if(v->kind != JVal::NUM || !(v->num >= 0) || !(v->num <= (double)c.bin_bytes) || v->num != std::floor(v->num)) return false;
        out_d = v->num;
        return true;
    };
    double bv_off_d, bv_len_d, acc_off_d, stride_d;
    if(!whole(bo, 0.0, bv_off_d) || !whole(bl, 0.0, bv_len_d) || !whole(ao, 0.0, acc_off_d) || !whole(bs, 0.0, stride_d))
        return c.fail("bufferView / accessor offset, length or stride is not a whole number inside the binary chunk");
    if(cnt->num != std::floor(cnt->num)) return c.fail("accessor count is not a whole number");
    const size_t bv_off = (size_t)bv_off_d, bv_len = (size_t)bv_len_d, acc_off = (size_t)acc_off_d;
    if(bv_off > c.bin_bytes || bv_len > c.bin_bytes - bv_off) return c.fail("bufferView reaches beyond the binary chunk");
    const size_t elem = (size_t)csz * ncomp;
    const size_t stride = stride_d > 0 ? (size_t)stride_d : elem;
    if(stride < elem) return c.fail("bufferView stride smaller than the element");
    if(count) {
        // inside the bufferView: acc_off + stride * (count - 1) + elem <= bv_len, with no product that can wrap
        if(acc_off > bv_len || elem > bv_len - acc_off) return c.fail("accessor reaches beyond its bufferView");
        if((count - 1) > (bv_len - acc_off - elem) / stride) return c.fail("accessor reaches beyond its bufferView");
    }
    const size_t off = bv_off + acc_off;
    if(fout) fout->resize(count * ncomp);
    if(iout) iout->resize(count * ncomp);
    for(size_t i = 0; i < count; i++) {
        const uint8_t *src = c.bin + off + i * stride;
        for(int k = 0; k < ncomp; k++) {
            const uint8_t *q = src + (size_t)k * csz;
            if(as_index) {
                uint32_t v = 0;
                if(csz == 1) v = *q;
                else if(csz == 2) { uint16_t t; memcpy(&t, q, 2); v = t; }
                else memcpy(&v, q, 4);
                (*iout)[i * ncomp + k] = v;
            } else {
                float f;
                memcpy(&f, q, 4);
                (*fout)[i * ncomp + k] = f;
            }
        }
    }
    return true;
}

bool node_matrix(Ctx &c, const JVal &n, Mat4 &out) {
    if(const JVal *m = n.get("matrix")) {
        if(m->size() != 16) return c.fail("node matrix must have 16 numbers");
        float d[16];
        for(int k = 0; k < 16; k++) {
            if(m->arr[k].kind != JVal::NUM) return c.fail("node matrix must have 16 numbers");
            d[k] = (float)m->arr[k].num;
        }
        memcpy(out.data(), d, 64);  // glTF matrices are column major, like Mat4
        return true;
    }
    if(n.has("rotation")) return c.fail("node rotations given as quaternions are not mapped (the reference decomposes assimp's matrix into Euler angles)");
    float t[3] = {0, 0, 0}, s[3] = {1, 1, 1}, r[3] = {0, 0, 0};
    for(int which = 0; which < 2; which++) {
        const JVal *v = n.get(which ? "scale" : "translation");
        if(!v) continue;
        if(v->size() != 3) return c.fail("node translation / scale must have 3 numbers");
        for(int k = 0; k < 3; k++) {
            if(v->arr[k].kind != JVal::NUM) return c.fail("node translation / scale must have 3 numbers");
            (which ? s : t)[k] = (float)v->arr[k].num;
        }
    }
    out = transform_matrix(t, r, s);
    return true;
}

bool walk(Ctx &c, size_t ni, const Mat4 &parent, bool is_root, const std::vector<int32_t> &mesh_material, GlbScene &out, int depth,
          std::vector<uint8_t> &visited) {
    const JVal *nodes = c.js->get("nodes");
    if(depth > 256) return c.fail("node hierarchy deeper than 256");
    // glTF hierarchies are strict trees: a node reached twice is a cycle or a shared child (which a small file
    // could use to make this walk exponential)
    if(visited[ni]) return c.fail("node " + std::to_string(ni) + " is reached twice (hierarchies must be trees)");
    visited[ni] = 1;
    const JVal &n = nodes->arr[ni];
    Mat4 local = Mat4::identity();
    if(!is_root && !node_matrix(c, n, local)) return false;
    const Mat4 world = mul(parent, local);
    if(const JVal *m = n.get("mesh")) {
        size_t mi;
        if(!index_of(m, mesh_material.size(), mi)) return c.fail("node names a mesh that does not exist");
        GlbInstance inst;
        inst.mesh = (int32_t)mi;
        inst.material = mesh_material[mi];
        inst.l2w = world;
        out.instances.push_back(inst);
    }
    if(const JVal *ch = n.get("children")) {
        for(const JVal &cv : ch->arr) {
            size_t ci;
            if(!index_of(&cv, nodes->size(), ci)) return c.fail("child node index out of range");
            if(!walk(c, ci, world, false, mesh_material, out, depth + 1, visited)) return false;
        }
    }
    return true;
}

}  // namespace

bool load_glb(const uint8_t *data, size_t n, GlbScene &out, std::string &err) {
    err.clear();
    out = GlbScene();
    auto u32 = [&](size_t off) {
        uint32_t v;
        memcpy(&v, data + off, 4);
        return v;
    };
    if(!data || n < 28) { err = "glb: file too short"; return false; }
    if(u32(0) != 0x46546C67u || u32(4) != 2u) { err = "glb: not a glTF 2.0 binary"; return false; }
    const size_t clen = u32(12);
    if(u32(16) != 0x4E4F534Au || 20 + clen > n) { err = "glb: first chunk is not JSON or is truncated"; return false; }
    const size_t boff = 20 + clen;
    size_t blen = 0;
    if(boff + 8 <= n) {
        blen = u32(boff);
        if(u32(boff + 4) != 0x004E4942u || boff + 8 + blen > n) { err = "glb: binary chunk is malformed or truncated"; return false; }
    }
    out.file.assign(data, data + n);
    out.bin_offset = boff + 8;
    out.bin_bytes = blen;

    JVal js;
    JParser jp{reinterpret_cast<const char *>(out.file.data()) + 20, reinterpret_cast<const char *>(out.file.data()) + 20 + clen, {}, 0};
    if(!jp.value(js) || js.kind != JVal::OBJ) { err = "glb: " + (jp.err.empty() ? std::string("JSON root is not an object") : jp.err); return false; }
    Ctx c{&js, out.file.data() + out.bin_offset, out.bin_bytes, &err};

    // ---- materials (loader.cpp:44-73)
    const JVal *jtex = js.get("textures"), *jimg = js.get("images"), *jbv = js.get("bufferViews");
    if(const JVal *mats = js.get("materials")) {
        for(const JVal &m : mats->arr) {
            GlbMaterial gm;
            gm.albedo[0] = gm.albedo[1] = gm.albedo[2] = 1.0f;
            float rough = 1.0f;
            gm.texture = -1;
            const JVal *pbr = m.get("pbrMetallicRoughness");
            if(pbr) {
                if(const JVal *col = pbr->get("baseColorFactor")) {
                    if(col->size() < 3) return c.fail("baseColorFactor needs at least 3 numbers");
                    for(int k = 0; k < 3; k++) gm.albedo[k] = (float)col->arr[k].num;
                }
                if(const JVal *r = pbr->get("roughnessFactor"))
                    if(r->kind == JVal::NUM) rough = (float)r->num;
                if(const JVal *bt = pbr->get("baseColorTexture")) {
                    size_t ti, ii, bvi;
                    if(!jtex || !index_of(bt->get("index"), jtex->size(), ti)) return c.fail("baseColorTexture names no texture");
                    if(!jimg || !index_of(jtex->arr[ti].get("source"), jimg->size(), ii)) return c.fail("texture names no image");
                    const JVal &img = jimg->arr[ii];
                    if(!jbv || !index_of(img.get("bufferView"), jbv->size(), bvi)) return c.fail("only images embedded in the binary chunk are supported");
                    const JVal &bv = jbv->arr[bvi];
                    const JVal *bo = bv.get("byteOffset"), *bl = bv.get("byteLength");
                    if(!bl || bl->kind != JVal::NUM) return c.fail("image bufferView without byteLength");
                    GlbTexture t;
                    const double o = bo && bo->kind == JVal::NUM ? bo->num : 0.0;
                    if(o < 0 || bl->num < 0 || o + bl->num > (double)out.bin_bytes) return c.fail("image bytes reach beyond the binary chunk");
                    t.offset = out.bin_offset + (size_t)o;
                    t.bytes = (size_t)bl->num;
                    t.image = (int32_t)ii;
                    if(const JVal *mt = img.get("mimeType"))
                        if(mt->kind == JVal::STR) t.mime = mt->str;
                    gm.texture = (int32_t)out.textures.size();
                    out.textures.push_back(t);  // one texture object per material that names one (loader.cpp:63-65)
                }
            }
            gm.roughness = rough > 0.001f ? rough : 0.001f;  // std::max(0.001f, res), loader.cpp:57
            gm.metallic = 1.0f;                              // loader.cpp:59-61
            out.materials.push_back(gm);
        }
    }

    // ---- meshes (loader.cpp:75-104)
    const JVal *meshes = js.get("meshes");
    if(!meshes || meshes->size() == 0) return c.fail("file has no meshes");
    std::vector<int32_t> mesh_material;
    for(const JVal &m : meshes->arr) {
        const JVal *prims = m.get("primitives");
        if(!prims || prims->size() != 1) return c.fail("a mesh with other than one primitive (the reference asserts one mesh per node, loader.cpp:116)");
        const JVal &p = prims->arr[0];
        if(const JVal *mode = p.get("mode"))
            if(mode->kind != JVal::NUM || mode->num != 4) return c.fail("only triangle lists (mode 4) are supported");
        const JVal *attrs = p.get("attributes");
        const JVal *accs = js.get("accessors");
        size_t a_pos, a_nor, a_uv = 0, a_idx;
        if(!attrs || !accs || !index_of(attrs->get("POSITION"), accs->size(), a_pos)) return c.fail("primitive without POSITION");
        if(!index_of(attrs->get("NORMAL"), accs->size(), a_nor)) return c.fail("primitive without NORMAL (normal generation is the importer's, not reproduced)");
        const bool has_uv = attrs->has("TEXCOORD_0");
        if(has_uv && !index_of(attrs->get("TEXCOORD_0"), accs->size(), a_uv)) return c.fail("TEXCOORD_0 names no accessor");
        if(!index_of(p.get("indices"), accs->size(), a_idx)) return c.fail("non-indexed primitives are not supported");
        std::vector<float> pos, nor, uv;
        std::vector<uint32_t> idx;
        size_t n_pos, n_nor, n_uv = 0, n_idx;
        if(!read_accessor(c, a_pos, 3, false, &pos, nullptr, n_pos)) return false;
        if(!read_accessor(c, a_nor, 3, false, &nor, nullptr, n_nor)) return false;
        if(has_uv && !read_accessor(c, a_uv, 2, false, &uv, nullptr, n_uv)) return false;
        if(!read_accessor(c, a_idx, 1, true, nullptr, &idx, n_idx)) return false;
        if(n_nor != n_pos || (has_uv && n_uv != n_pos)) return c.fail("vertex attribute counts differ");
        if(n_idx % 3 || n_idx == 0) return c.fail("index count is not a positive multiple of 3");
        if(n_idx / 3 > 0x3FFFFFFFu) return c.fail("too many triangles");
        const size_t base = out.triangles.size();
        out.triangles.resize(base + n_idx / 3);
        for(size_t t = 0; t < n_idx / 3; t++) {
            TriangleRec &r = out.triangles[base + t];
            for(int v = 0; v < 3; v++) {
                const uint32_t i = idx[t * 3 + v];
                if(i >= n_pos) return c.fail("vertex index out of range");
                for(int k = 0; k < 3; k++) {
                    r.ver[v * 3 + k] = pos[(size_t)i * 3 + k];
                    r.normal[v * 3 + k] = nor[(size_t)i * 3 + k];
                }
                r.uv[v * 2] = has_uv ? uv[(size_t)i * 2] : 0.0f;
                r.uv[v * 2 + 1] = has_uv ? uv[(size_t)i * 2 + 1] : 0.0f;
            }
        }
        out.tri_count.push_back((int32_t)(n_idx / 3));
        int32_t mat = -1;
        if(const JVal *pm = p.get("material")) {
            size_t mi;
            if(!index_of(pm, out.materials.size(), mi)) return c.fail("primitive names a material that does not exist");
            mat = (int32_t)mi;
        }
        mesh_material.push_back(mat);
    }

    // ---- nodes (loader.cpp:106-135, 152-156; ResourceManager.cpp:47-77)
    const JVal *scenes = js.get("scenes"), *nodes = js.get("nodes");
    if(!scenes || scenes->size() == 0 || !nodes) return c.fail("file has no scene");
    size_t si = 0;
    if(js.has("scene") && !index_of(js.get("scene"), scenes->size(), si)) return c.fail("default scene index out of range");
    const JVal *roots = scenes->arr[si].get("nodes");
    if(!roots || roots->size() != 1) return c.fail("scenes with other than one root node get a synthetic root from the importer; not mapped");
    size_t ri;
    if(!index_of(&roots->arr[0], nodes->size(), ri)) return c.fail("root node index out of range");
    std::vector<uint8_t> visited(nodes->size(), 0);
    if(!walk(c, ri, Mat4::identity(), true, mesh_material, out, 0, visited)) return false;
    if(out.instances.empty()) return c.fail("no node carries a mesh");
    return true;
}

}  // namespace rta
