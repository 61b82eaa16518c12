// TEST INFRASTRUCTURE, build container only: compiles the reference's OWN GLSL text of the BSDF
// library -- shader/basic/math.glsl, shader/basic/sobol.glsl (+ sobol_matrix.glsl),
// shader/materials/{materials,RoughConductor,RoughDielectric}.glsl -- as C++ against the glm
// headers vendored in the reference, and prints known-answer vectors for eval / pdf / sample /
// albedo of both materials and for the Sobol sampler.  tests/golden/glsl_bsdf_kat.npz is made from
// its output by oracle/glsl_kat.py; the oracle (pt_oracle.cpp) is checked against those vectors.
//
// The GLSL text is not part of this repository.  oracle/glsl_kat.py reads it where it lies under
// /root/reference, expands its `#include shader/...` lines the way the reference's read_shader does
// (src/tool/tool.cpp:42-68), rewrites the three parameter qualifiers C++ has no spelling for
//     `out float x` -> `float &x`,  `inout T x` -> `T &x`,  `in T x` -> `T x`
// drops the storage qualifiers of the direction-number table (`const uniform` / `shared`), and
// writes the result to oracle/_ref/glsl_materials.inc (git-ignored).  Every expression, constant
// and statement of the shader functions is the reference's, compiled here by g++ with
// -fsingle-precision-constant (GLSL's unsuffixed literals are floats) and -ffp-contract=off.
// What this harness supplies is only what the GL driver would: the buffer the shader reads
// material records from, a texture() that is never reached (the vectors use no albedo map),
// int -> float conversions GLSL does implicitly.
#include <cstdio>
#include <cstdlib>
#include <glm/glm.hpp>

namespace ref_glsl {
using namespace glm;

// ---- what the driver / GLSL's implicit conversions provide ------------------------------------
static float materialBuffer[256];
struct Texel { vec3 rgb; };
static int textureBuffer[4];
static inline Texel texture(int, vec2) {
    fprintf(stderr, "texture() reached: the vectors must not use an albedo map\n");
    abort();
}
static inline float step(float edge, int x) { return glm::step(edge, (float)x); }
static inline float step(float edge, float x) { return glm::step(edge, x); }
static inline float max(float a, float b) { return glm::max(a, b); }

#include "_ref/glsl_materials.inc"

}  // namespace ref_glsl

using namespace ref_glsl;

// One line per query:  which mptr wi wo seed dim  ->  value[3] pdf wo[3] eta dim_after
int main() {
    // records: [type, r, g, b, roughness, metallic, texIndex(-1) (, ior)], as Material::insert_buffer writes them
    int n_rec = 0;
    if(scanf("%d", &n_rec) != 1 || n_rec < 1 || n_rec > 255) return 2;
    for(int k = 0; k < n_rec; k++)
        if(scanf("%f", &materialBuffer[k]) != 1) return 2;
    int which, mptr;
    float wi[3], wo[3];
    unsigned seed, dim;
    while(scanf("%d %d %f %f %f %f %f %f %u %u", &which, &mptr, &wi[0], &wi[1], &wi[2], &wo[0], &wo[1], &wo[2], &seed, &dim) == 10) {
        if(which == 4) {  // the sampler alone: sobol(dim, seed)
            printf("%.9g\n", sobol(dim, seed));
            continue;
        }
        BSDFQueryRecord b;
        b.wi = vec3(wi[0], wi[1], wi[2]);
        b.wo = vec3(wo[0], wo[1], wo[2]);
        b.mptr = mptr;
        b.uv = vec2(0.25f, 0.75f);
        b.eta = 1.0f;
        sobolseed = seed;
        sobolcurdim = dim;
        vec3 v(0.0f);
        float pdf = 0.0f;
        switch(which) {
            case 0: v = eval_material(b); break;
            case 1: pdf = pdf_material(b); break;
            case 2: v = sample_material(b, pdf); break;
            case 3: v = albedo_material(b); break;
            default: return 2;
        }
        printf("%.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %u\n", v.x, v.y, v.z, pdf, b.wo.x, b.wo.y, b.wo.z, b.eta, sobolcurdim);
    }
    return 0;
}
