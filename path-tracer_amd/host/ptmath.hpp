// Small vector / matrix library used by the host pipeline and by the HIP kernels.
//
// The reference does all of its vector arithmetic through GLM (dataTypes.hpp:10-12), which is not
// vendored in the reference and not installed here.  Every operation below therefore restates the
// *evaluation order* of GLM 0.9.9.x's scalar (non-SIMD) code path, because the bit pattern of the
// BVH, of the instance matrices and of the `-a` heat map depends on it.
// All float code in this repository is compiled with -ffp-contract=off so that `a*b+c` is two
// roundings on x86-64 and on gfx950 alike.
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ inline
#else
#define PT_HD inline
#endif

namespace pt {

typedef unsigned int uint;

// GLM scalar min/max: min(x,y) = (y < x) ? y : x ; max(x,y) = (x < y) ? y : x.  They differ from
// fminf/fmaxf only when an operand is NaN (NaN in x propagates, NaN in y is dropped).
PT_HD float gmin(float x, float y) { return (y < x) ? y : x; }
PT_HD float gmax(float x, float y) { return (x < y) ? y : x; }
PT_HD uint gmin(uint x, uint y) { return (y < x) ? y : x; }
PT_HD float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }

struct vec2 {
  float x, y;
  PT_HD vec2() : x(0), y(0) {}
  PT_HD explicit vec2(float s) : x(s), y(s) {}
  PT_HD vec2(float x_, float y_) : x(x_), y(y_) {}
};
PT_HD vec2 operator+(vec2 a, vec2 b) { return vec2(a.x + b.x, a.y + b.y); }
PT_HD vec2 operator-(vec2 a, vec2 b) { return vec2(a.x - b.x, a.y - b.y); }
PT_HD vec2 operator*(vec2 a, vec2 b) { return vec2(a.x * b.x, a.y * b.y); }
PT_HD vec2 operator*(float s, vec2 a) { return vec2(s * a.x, s * a.y); }
PT_HD vec2 operator*(vec2 a, float s) { return vec2(a.x * s, a.y * s); }
PT_HD vec2 operator-(vec2 a, float s) { return vec2(a.x - s, a.y - s); }

struct vec3 {
  float x, y, z;
  PT_HD vec3() : x(0), y(0), z(0) {}
  PT_HD explicit vec3(float s) : x(s), y(s), z(s) {}
  PT_HD vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
  PT_HD float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  PT_HD float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
PT_HD vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD vec3 operator/(vec3 a, vec3 b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
PT_HD vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
PT_HD vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
PT_HD vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
PT_HD vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
PT_HD vec3 operator/(float s, vec3 a) { return vec3(s / a.x, s / a.y, s / a.z); }
PT_HD vec3 operator+(vec3 a, float s) { return vec3(a.x + s, a.y + s, a.z + s); }
PT_HD vec3 operator-(vec3 a, float s) { return vec3(a.x - s, a.y - s, a.z - s); }
PT_HD vec3 operator+(float s, vec3 a) { return vec3(s + a.x, s + a.y, s + a.z); }
PT_HD vec3 operator-(float s, vec3 a) { return vec3(s - a.x, s - a.y, s - a.z); }
PT_HD vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
PT_HD vec3& operator*=(vec3& a, vec3 b) { a = a * b; return a; }

PT_HD vec3 gmin(vec3 a, vec3 b) { return vec3(gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)); }
PT_HD vec3 gmax(vec3 a, vec3 b) { return vec3(gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)); }
PT_HD vec3 gclamp(vec3 a, float lo, float hi) {
  return vec3(gclamp(a.x, lo, hi), gclamp(a.y, lo, hi), gclamp(a.z, lo, hi));
}

// glm::dot(vec3): tmp = a*b; (tmp.x + tmp.y) + tmp.z
PT_HD float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// glm::cross
PT_HD vec3 cross(vec3 a, vec3 b) {
  return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
// glm::normalize: v * inversesqrt(dot(v,v)), inversesqrt(x) = 1 / sqrt(x)
PT_HD vec3 normalize(vec3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
// glm::mix(x, y, a) with scalar a: x*(1-a) + y*a
PT_HD vec3 mix(vec3 x, vec3 y, float a) { return x * (1.0f - a) + y * a; }
// glm::reflect: I - N * dot(N, I) * 2
PT_HD vec3 reflect(vec3 I, vec3 N) { return I - N * dot(N, I) * 2.0f; }
// glm::fract: x - floor(x)
PT_HD float fract(float x) { return x - floorf(x); }
// glm::smoothstep(e0, e1, x): t = clamp((x-e0)/(e1-e0), 0, 1); t*t*(3 - 2*t)
PT_HD float smoothstep(float e0, float e1, float x) {
  float t = gclamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
  return t * t * (3.0f - 2.0f * t);
}
// glm::radians
PT_HD float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

struct vec4 {
  float x, y, z, w;
  PT_HD vec4() : x(0), y(0), z(0), w(0) {}
  PT_HD vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
  PT_HD vec4(vec3 v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
  PT_HD float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
};
PT_HD vec4 operator+(vec4 a, vec4 b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
PT_HD vec4 operator-(vec4 a, vec4 b) { return vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
PT_HD vec4 operator*(vec4 a, vec4 b) { return vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
PT_HD vec4 operator*(vec4 a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
PT_HD vec3 xyz(vec4 v) { return vec3(v.x, v.y, v.z); }

// Column-major 3x3 (glm::mat3): c[i] is column i.
struct mat3 {
  vec3 c[3];
};
// glm mat3 * vec3: row r = m[0][r]*v.x + m[1][r]*v.y + m[2][r]*v.z, left to right
PT_HD vec3 operator*(const mat3& m, vec3 v) {
  return vec3(m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z,
              m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z,
              m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z);
}

// Column-major 4x4 (glm::mat4): c[i] is column i, c[i][j] is row j of column i.
struct mat4 {
  vec4 c[4];
  PT_HD static mat4 identity() {
    mat4 m;
    m.c[0] = vec4(1, 0, 0, 0);
    m.c[1] = vec4(0, 1, 0, 0);
    m.c[2] = vec4(0, 0, 1, 0);
    m.c[3] = vec4(0, 0, 0, 1);
    return m;
  }
};
// glm mat4 * vec4: (m[0]*v.x + m[1]*v.y) + (m[2]*v.z + m[3]*v.w)
PT_HD vec4 operator*(const mat4& m, vec4 v) {
  vec4 add0 = m.c[0] * v.x + m.c[1] * v.y;
  vec4 add1 = m.c[2] * v.z + m.c[3] * v.w;
  return add0 + add1;
}
PT_HD mat4 operator*(const mat4& m, float s) {
  mat4 r;
  for (int i = 0; i < 4; i++) r.c[i] = m.c[i] * s;
  return r;
}
PT_HD mat4 transpose(const mat4& m) {
  mat4 r;
  r.c[0] = vec4(m.c[0].x, m.c[1].x, m.c[2].x, m.c[3].x);
  r.c[1] = vec4(m.c[0].y, m.c[1].y, m.c[2].y, m.c[3].y);
  r.c[2] = vec4(m.c[0].z, m.c[1].z, m.c[2].z, m.c[3].z);
  r.c[3] = vec4(m.c[0].w, m.c[1].w, m.c[2].w, m.c[3].w);
  return r;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// --- host-only: glm/gtc/matrix_transform.hpp restatements (mesh.cpp:18-41 uses them) ---
inline mat4 scale(const mat4& m, vec3 v) {
  mat4 r;
  r.c[0] = m.c[0] * v.x;
  r.c[1] = m.c[1] * v.y;
  r.c[2] = m.c[2] * v.z;
  r.c[3] = m.c[3];
  return r;
}
inline mat4 rotate(const mat4& m, float angle, vec3 v) {
  const float a = angle;
  const float c = cosf(a);
  const float s = sinf(a);
  vec3 axis = normalize(v);
  vec3 temp = (1.0f - c) * axis;
  float R00 = c + temp.x * axis.x;
  float R01 = temp.x * axis.y + s * axis.z;
  float R02 = temp.x * axis.z - s * axis.y;
  float R10 = temp.y * axis.x - s * axis.z;
  float R11 = c + temp.y * axis.y;
  float R12 = temp.y * axis.z + s * axis.x;
  float R20 = temp.z * axis.x + s * axis.y;
  float R21 = temp.z * axis.y - s * axis.x;
  float R22 = c + temp.z * axis.z;
  mat4 r;
  r.c[0] = m.c[0] * R00 + m.c[1] * R01 + m.c[2] * R02;
  r.c[1] = m.c[0] * R10 + m.c[1] * R11 + m.c[2] * R12;
  r.c[2] = m.c[0] * R20 + m.c[1] * R21 + m.c[2] * R22;
  r.c[3] = m.c[3];
  return r;
}
inline mat4 translate(const mat4& m, vec3 v) {
  mat4 r = m;
  r.c[3] = m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z + m.c[3];
  return r;
}
// glm::inverse(mat4): 18 sub-factor cofactor expansion, det = (d.x + d.y) + (d.z + d.w).
inline mat4 inverse(const mat4& m) {
#define M(i, j) m.c[i][j]
  float Coef00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
  float Coef02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
  float Coef03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
  float Coef04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
  float Coef06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
  float Coef07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
  float Coef08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
  float Coef10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
  float Coef11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
  float Coef12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
  float Coef14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
  float Coef15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
  float Coef16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
  float Coef18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
  float Coef19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
  float Coef20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
  float Coef22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
  float Coef23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
  vec4 Fac0(Coef00, Coef00, Coef02, Coef03);
  vec4 Fac1(Coef04, Coef04, Coef06, Coef07);
  vec4 Fac2(Coef08, Coef08, Coef10, Coef11);
  vec4 Fac3(Coef12, Coef12, Coef14, Coef15);
  vec4 Fac4(Coef16, Coef16, Coef18, Coef19);
  vec4 Fac5(Coef20, Coef20, Coef22, Coef23);
  vec4 Vec0(M(1, 0), M(0, 0), M(0, 0), M(0, 0));
  vec4 Vec1(M(1, 1), M(0, 1), M(0, 1), M(0, 1));
  vec4 Vec2(M(1, 2), M(0, 2), M(0, 2), M(0, 2));
  vec4 Vec3(M(1, 3), M(0, 3), M(0, 3), M(0, 3));
#undef M
  vec4 Inv0 = Vec1 * Fac0 - Vec2 * Fac1 + Vec3 * Fac2;
  vec4 Inv1 = Vec0 * Fac0 - Vec2 * Fac3 + Vec3 * Fac4;
  vec4 Inv2 = Vec0 * Fac1 - Vec1 * Fac3 + Vec3 * Fac5;
  vec4 Inv3 = Vec0 * Fac2 - Vec1 * Fac4 + Vec2 * Fac5;
  vec4 SignA(+1, -1, +1, -1);
  vec4 SignB(-1, +1, -1, +1);
  mat4 Inverse;
  Inverse.c[0] = Inv0 * SignA;
  Inverse.c[1] = Inv1 * SignB;
  Inverse.c[2] = Inv2 * SignA;
  Inverse.c[3] = Inv3 * SignB;
  vec4 Row0(Inverse.c[0].x, Inverse.c[1].x, Inverse.c[2].x, Inverse.c[3].x);
  vec4 Dot0 = m.c[0] * Row0;
  float Dot1 = (Dot0.x + Dot0.y) + (Dot0.z + Dot0.w);
  float OneOverDeterminant = 1.0f / Dot1;
  return Inverse * OneOverDeterminant;
}
#endif

}  // namespace pt
