// Host-side fp32 vector / matrix arithmetic with a FIXED operation order.
//
// The scene arrays this library produces must be bit-identical to the ones the
// reference produces with glm, so every composite operation below spells out the
// association order glm uses (all cited relative to the reference tree):
//   mat4*vec4    (m0*v0 + m1*v1) + (m2*v2 + m3*v3)     include/glm/detail/type_mat4x4.inl:561-572
//   mat4*mat4    ((A0*b0 + A1*b1) + A2*b2) + A3*b3     include/glm/detail/type_mat4x4.inl:643-646
//   inverse      cofactor form, det = (d0+d1)+(d2+d3)  include/glm/detail/func_matrix.inl:298-350
// Build with -ffp-contract=off: a fused multiply-add anywhere here changes results.
#pragma once
#include <cmath>

namespace rta {

struct Vec3 {
    float x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

struct Vec4 {
    float x, y, z, w;
    Vec4() : x(0), y(0), z(0), w(0) {}
    Vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    Vec4(const Vec3 &v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
};

inline Vec4 operator*(const Vec4 &a, float s) { return Vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline Vec4 operator*(const Vec4 &a, const Vec4 &b) { return Vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline Vec4 operator+(const Vec4 &a, const Vec4 &b) { return Vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline Vec4 operator-(const Vec4 &a, const Vec4 &b) { return Vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// Column-major 4x4, c[j] is column j (same memory layout as glm::mat4 and as the
// std430 mat4 the device reads).
struct Mat4 {
    Vec4 c[4];
    Mat4() {}
    static Mat4 identity() {
        Mat4 m;
        m.c[0] = Vec4(1, 0, 0, 0); m.c[1] = Vec4(0, 1, 0, 0);
        m.c[2] = Vec4(0, 0, 1, 0); m.c[3] = Vec4(0, 0, 0, 1);
        return m;
    }
    // sixteen values given row by row (the reference's glm::matbyrow, src/tool/exglm.hpp:48-57)
    static Mat4 by_rows(const float d[16]) {
        Mat4 m;
        m.c[0] = Vec4(d[0], d[4], d[8], d[12]);
        m.c[1] = Vec4(d[1], d[5], d[9], d[13]);
        m.c[2] = Vec4(d[2], d[6], d[10], d[14]);
        m.c[3] = Vec4(d[3], d[7], d[11], d[15]);
        return m;
    }
    const float *data() const { return &c[0].x; }
    float *data() { return &c[0].x; }
    float at(int col, int row) const { return (&c[col].x)[row]; }
};

inline Vec4 mul(const Mat4 &m, const Vec4 &v) {
    Vec4 add0 = m.c[0] * v.x + m.c[1] * v.y;
    Vec4 add1 = m.c[2] * v.z + m.c[3] * v.w;
    return add0 + add1;
}

inline Mat4 mul(const Mat4 &a, const Mat4 &b) {
    Mat4 r;
    for(int j = 0; j < 4; j++) {
        const Vec4 &bj = b.c[j];
        r.c[j] = ((a.c[0] * bj.x + a.c[1] * bj.y) + a.c[2] * bj.z) + a.c[3] * bj.w;
    }
    return r;
}

inline Mat4 inverse(const Mat4 &m) {
#define M(col, row) m.at(col, row)
    float Coef00 = M(2,2) * M(3,3) - M(3,2) * M(2,3);
    float Coef02 = M(1,2) * M(3,3) - M(3,2) * M(1,3);
    float Coef03 = M(1,2) * M(2,3) - M(2,2) * M(1,3);
    float Coef04 = M(2,1) * M(3,3) - M(3,1) * M(2,3);
    float Coef06 = M(1,1) * M(3,3) - M(3,1) * M(1,3);
    float Coef07 = M(1,1) * M(2,3) - M(2,1) * M(1,3);
    float Coef08 = M(2,1) * M(3,2) - M(3,1) * M(2,2);
    float Coef10 = M(1,1) * M(3,2) - M(3,1) * M(1,2);
    float Coef11 = M(1,1) * M(2,2) - M(2,1) * M(1,2);
    float Coef12 = M(2,0) * M(3,3) - M(3,0) * M(2,3);
    float Coef14 = M(1,0) * M(3,3) - M(3,0) * M(1,3);
    float Coef15 = M(1,0) * M(2,3) - M(2,0) * M(1,3);
    float Coef16 = M(2,0) * M(3,2) - M(3,0) * M(2,2);
    float Coef18 = M(1,0) * M(3,2) - M(3,0) * M(1,2);
    float Coef19 = M(1,0) * M(2,2) - M(2,0) * M(1,2);
    float Coef20 = M(2,0) * M(3,1) - M(3,0) * M(2,1);
    float Coef22 = M(1,0) * M(3,1) - M(3,0) * M(1,1);
    float Coef23 = M(1,0) * M(2,1) - M(2,0) * M(1,1);
    Vec4 Fac0(Coef00, Coef00, Coef02, Coef03);
    Vec4 Fac1(Coef04, Coef04, Coef06, Coef07);
    Vec4 Fac2(Coef08, Coef08, Coef10, Coef11);
    Vec4 Fac3(Coef12, Coef12, Coef14, Coef15);
    Vec4 Fac4(Coef16, Coef16, Coef18, Coef19);
    Vec4 Fac5(Coef20, Coef20, Coef22, Coef23);
    Vec4 V0(M(1,0), M(0,0), M(0,0), M(0,0));
    Vec4 V1(M(1,1), M(0,1), M(0,1), M(0,1));
    Vec4 V2(M(1,2), M(0,2), M(0,2), M(0,2));
    Vec4 V3(M(1,3), M(0,3), M(0,3), M(0,3));
#undef M
    Vec4 Inv0 = (V1 * Fac0 - V2 * Fac1) + V3 * Fac2;
    Vec4 Inv1 = (V0 * Fac0 - V2 * Fac3) + V3 * Fac4;
    Vec4 Inv2 = (V0 * Fac1 - V1 * Fac3) + V3 * Fac5;
    Vec4 Inv3 = (V0 * Fac2 - V1 * Fac4) + V2 * Fac5;
    Vec4 SignA(+1, -1, +1, -1), SignB(-1, +1, -1, +1);
    Mat4 inv;
    inv.c[0] = Inv0 * SignA; inv.c[1] = Inv1 * SignB;
    inv.c[2] = Inv2 * SignA; inv.c[3] = Inv3 * SignB;
    Vec4 row0(inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x);
    Vec4 dot0 = m.c[0] * row0;
    float det = (dot0.x + dot0.y) + (dot0.z + dot0.w);
    float ood = 1.0f / det;
    for(int j = 0; j < 4; j++) inv.c[j] = inv.c[j] * ood;
    return inv;
}

}  // namespace rta
