// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the
// product library.  Only tests/ and fixture generators may run what this file builds.
//
// Compiles three of the reference's own sources where they lie, as one translation unit:
//   src/Transform.cpp  src/instance/Instance.cpp  src/instance/Camera.cpp
// (the Makefile passes -I$(REF)/src -I$(REF)/include; nothing is copied).  They cannot be
// handed to g++ directly on glibc: <math.h> defines M_PI as a macro and the reference
// declares `const float M_PI = acos(-1)` (include/common.h:11) for a platform whose
// <math.h> does not.  Removing the macro after <cmath> restores the reference's own float
// constant, which is what Transform::matrix() (src/Transform.cpp:20-53) divides angles by.
// The ImGui calls of Transform::insert_gui stay unresolved at link time; they are never
// reached.
#include <cmath>
#undef M_PI
#include "Transform.cpp"
#include "instance/Instance.cpp"
#include "instance/Camera.cpp"

// prs9 = position, rotation in degrees, scale.  out16 = glm::mat4 memory (column major).
void ref_transform_matrix(const float *prs9, float *out16) {
    Transform t(vec3(prs9[0], prs9[1], prs9[2]), vec3(prs9[3], prs9[4], prs9[5]), vec3(prs9[6], prs9[7], prs9[8]));
    mat4 m = t.matrix();
    memcpy(out16, &m, sizeof m);
}

// A camera under one parent instance: parent9 / cam9 as above.  out64 = matrix_to_global of
// the camera, w2v_matrix, v2w_matrix, projection.  screen_wh_out = the reference's compiled
// SCREEN_W, SCREEN_H (src/Config.h:14-15), which projection() reads.
void ref_camera_matrices(float fov_x, float far_plane, const float *parent9, const float *cam9, float *out64,
                         int *screen_wh_out) {
    Instance *parent = new Instance("parent", nullptr);
    parent->transform = Transform(vec3(parent9[0], parent9[1], parent9[2]), vec3(parent9[3], parent9[4], parent9[5]),
                                  vec3(parent9[6], parent9[7], parent9[8]));
    Camera *cam = new Camera(fov_x, far_plane);
    cam->transform = Transform(vec3(cam9[0], cam9[1], cam9[2]), vec3(cam9[3], cam9[4], cam9[5]),
                               vec3(cam9[6], cam9[7], cam9[8]));
    parent->add_child(cam);
    mat4 m[4] = {cam->matrix_to_global(), cam->w2v_matrix(), cam->v2w_matrix(), cam->projection()};
    memcpy(out64, m, sizeof m);
    screen_wh_out[0] = SCREEN_W;
    screen_wh_out[1] = SCREEN_H;
    delete parent;   // ~Instance deletes the children it holds (src/instance/Instance.cpp:43-49)
}
