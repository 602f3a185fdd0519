// Ray / box, ray / triangle and instance-transform primitives for one wave64 lane.
//
// Behaviour contract (reference intersection.hpp:9-75, gpuFunctions.hpp:32-41): the `-a` heat map is
// bit-exact against the CPU path only if every rounding is identical, so the arithmetic keeps the
// reference's operation order (no FMA contraction: -ffp-contract=off, correctly rounded division)
// and GLM's min/max NaN behaviour.
#pragma once

#include "../host/ptmath.hpp"
#include "device_scene.hpp"

namespace ptx {

using pt::vec3;

__device__ __forceinline__ vec3 f4xyz(const float4& q) { return vec3(q.x, q.y, q.z); }

// v_min_f32 / v_max_f32 / v_max3_f32 / v_min3_f32 written as asm so that the compiler does not put a
// canonicalising v_max_f32 x, x in front of every operand (it cannot prove the products are not
// signalling NaNs; the instructions themselves handle any input).
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// intersection.hpp:48-75 with glm::min/max semantics: min(x,y) = (y<x)?y:x, max(x,y) = (x<y)?y:x.
// EXACT_NAN = false uses v_min_f32 / v_max_f32 (IEEE minNum/maxNum), which return the same value as
// GLM's forms for every non-NaN input (a zero may differ in sign, which no comparison below can see).
// A NaN can only appear in t1/t2 as 0 * inf, i.e. when a component of the ray direction is exactly
// zero, so callers pick EXACT_NAN = true for a wave in which any lane holds such a ray.
// 1.0f / x in three instructions instead of the ten of the IEEE division sequence: v_rcp_f32 and one
// Newton step.  Checked against the correctly rounded quotient for every float (tools/rcp_check.hip,
// ptx_selftest_reciprocal): bit-identical whenever the biased exponent of x is in [1, 252], i.e.
// 2^-126 <= |x| < 2^126 (and NaN -> NaN); zeros, denormals, infinities and the two largest binades
// differ, so callers guard the range and fall back to the division.
__device__ __forceinline__ float rcpNewton(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  const float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
constexpr float kRcpLo = 0x1p-126f, kRcpHi = 0x1p126f;
#ifndef PTX_FAST_RCP
#define PTX_FAST_RCP 1
#endif
#ifndef PTX_FAST_RCP_TRI
#define PTX_FAST_RCP_TRI 1
#endif

__device__ __forceinline__ float absMin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float absMax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// inv = 1.0f / d per component (intersection.hpp: ray.invDirection) and the "a component is exactly
// zero" flag of the slab test.  Wave-uniform choice: all active lanes inside the checked range -> the
// short sequence (and no component can be zero); otherwise the IEEE division for everyone.
__device__ __forceinline__ void reciprocalDir(const vec3& d, vec3& inv, bool& zeroDir) {
  const float lo = absMin3(d.x, d.y, d.z), hi = absMax3(d.x, d.y, d.z);  // NaNs are skipped: NaN -> NaN either way
  const bool inRange = lo >= kRcpLo && hi < kRcpHi;
  if (PTX_FAST_RCP && __all(inRange)) {
    inv = vec3(rcpNewton(d.x), rcpNewton(d.y), rcpNewton(d.z));
    zeroDir = false;
  } else {
    inv = 1.0f / d;
    zeroDir = (d.x == 0.0f) || (d.y == 0.0f) || (d.z == 0.0f);
  }
}

template <bool EXACT_NAN>
__device__ __forceinline__ bool slabHit(const vec3& o, const vec3& inv, float rayT, const float4& bmin, const float4& bmax,
                                        float& tminOut) {
  const float t1x = (bmin.x - o.x) * inv.x, t1y = (bmin.y - o.y) * inv.y, t1z = (bmin.z - o.z) * inv.z;
  const float t2x = (bmax.x - o.x) * inv.x, t2y = (bmax.y - o.y) * inv.y, t2z = (bmax.z - o.z) * inv.z;
  float tmin, tmax;
  if (EXACT_NAN) {
    tmin = pt::gmax(pt::gmax(pt::gmin(t1x, t2x), pt::gmin(t1y, t2y)), pt::gmin(t1z, t2z));
    tmax = pt::gmin(pt::gmin(pt::gmax(t1x, t2x), pt::gmax(t1y, t2y)), pt::gmax(t1z, t2z));
  } else {
    tmin = vmax3(vmin(t1x, t2x), vmin(t1y, t2y), vmin(t1z, t2z));
    tmax = vmin3(vmax(t1x, t2x), vmax(t1y, t2y), vmax(t1z, t2z));
  }
  tminOut = tmin;
  return tmax >= tmin && tmin < rayT && tmax > 0.0f;
}

template <bool EXACT_NAN>
__device__ __forceinline__ float slabTest(const vec3& o, const vec3& inv, float rayT, const float4& bmin,
                                          const float4& bmax) {
  const float t1x = (bmin.x - o.x) * inv.x, t1y = (bmin.y - o.y) * inv.y, t1z = (bmin.z - o.z) * inv.z;
  const float t2x = (bmax.x - o.x) * inv.x, t2y = (bmax.y - o.y) * inv.y, t2z = (bmax.z - o.z) * inv.z;
  float tmin, tmax;
  if (EXACT_NAN) {
    tmin = pt::gmax(pt::gmax(pt::gmin(t1x, t2x), pt::gmin(t1y, t2y)), pt::gmin(t1z, t2z));
    tmax = pt::gmin(pt::gmin(pt::gmax(t1x, t2x), pt::gmax(t1y, t2y)), pt::gmax(t1z, t2z));
  } else {
    tmin = vmax3(vmin(t1x, t2x), vmin(t1y, t2y), vmin(t1z, t2z));
    tmax = vmin3(vmax(t1x, t2x), vmax(t1y, t2y), vmax(t1z, t2z));
  }
  return (tmax >= tmin && tmin < rayT && tmax > 0.0f) ? tmin : FLT_MAX;
}

// One step on a WideNode (wide_bvh.hpp): up to four children, boxes as bytes in the parent's frame.
// Returns how many children the ray hits (same acceptance as the slab test above: tmax >= tmin, tmin <
// ray.t, tmax > 0) and their references in s[0..n), nearest first.  SAFE = false folds origin and
// reciprocal direction into one fma per plane, t = byte * (scale * inv) + (frame - o) * inv; with an
// infinite inv that form turns into inf - inf, so a wave holding a ray with a zero direction component
// takes SAFE = true: plane = fma(byte, scale, frame), t = (plane - o) * inv, where the only NaN left is
// 0 * inf (origin exactly on a plane of a slab the ray is parallel to), which v_min/v_max skip.
template <int C>
__device__ __forceinline__ float ubf(uint32_t x) { return (float)((x >> (8 * C)) & 0xFFu); }  // v_cvt_f32_ubyteC

template <bool SAFE>
__device__ __forceinline__ uint32_t wideStep(const vec3& o, const vec3& inv, float rayT, const float4& q0, const float4& q1, const float4& q2,
                                             const float4& q3, uint32_t (&s)[4]) {
  const uint32_t eb = __float_as_uint(q0.w);
  const float sx = __uint_as_float((eb & 0xFFu) << 23), sy = __uint_as_float(((eb >> 8) & 0xFFu) << 23),
              sz = __uint_as_float(((eb >> 16) & 0xFFu) << 23);
  const uint32_t nKids = eb >> 24;
  const uint32_t lox = __float_as_uint(q1.x), loy = __float_as_uint(q1.y), loz = __float_as_uint(q1.z);
  const uint32_t hix = __float_as_uint(q1.w), hiy = __float_as_uint(q2.x), hiz = __float_as_uint(q2.y);
  float key[4];
  uint32_t nHit = 0;
  if (SAFE) {
    // general form: dequantise the plane, then the slab test of the binary path
#define PTX_WIDE_CHILD(c)                                                                                              \
  {                                                                                                                    \
    const float t1x = (__builtin_fmaf(ubf<c>(lox), sx, q0.x) - o.x) * inv.x, t2x = (__builtin_fmaf(ubf<c>(hix), sx, q0.x) - o.x) * inv.x; \
    const float t1y = (__builtin_fmaf(ubf<c>(loy), sy, q0.y) - o.y) * inv.y, t2y = (__builtin_fmaf(ubf<c>(hiy), sy, q0.y) - o.y) * inv.y; \
    const float t1z = (__builtin_fmaf(ubf<c>(loz), sz, q0.z) - o.z) * inv.z, t2z = (__builtin_fmaf(ubf<c>(hiz), sz, q0.z) - o.z) * inv.z; \
    const float tmin = vmax3(vmin(t1x, t2x), vmin(t1y, t2y), vmin(t1z, t2z));                                          \
    const float tmax = vmin3(vmax(t1x, t2x), vmax(t1y, t2y), vmax(t1z, t2z));                                          \
    const bool h = tmax >= tmin && tmin < rayT && tmax > 0.0f && (c < 2 || nKids > c);                                 \
    key[c] = h ? tmin : __builtin_inff();                                                                              \
    nHit += h ? 1u : 0u;                                                                                               \
  }
    PTX_WIDE_CHILD(0) PTX_WIDE_CHILD(1) PTX_WIDE_CHILD(2) PTX_WIDE_CHILD(3)
#undef PTX_WIDE_CHILD
  } else {
    // Every direction component is finite and non-zero here.  t = byte * (scale * inv) + (frame - o) * inv
    // is one fma per plane and monotone in the byte, so the sign of inv says which of lo / hi is the
    // entry plane: one select per axis picks the dwords that hold the four children's entry and exit
    // bytes, and no per-child min / max is needed.  The acceptance is widened to max(tmin, 0) <=
    // min(tmax, ray.t): a superset of the slab test's, which the wide layout allows (wide_bvh.hpp).
    const float ax = sx * inv.x, ay = sy * inv.y, az = sz * inv.z;
    const float bx = (q0.x - o.x) * inv.x, by = (q0.y - o.y) * inv.y, bz = (q0.z - o.z) * inv.z;
    const bool px = inv.x >= 0.0f, py = inv.y >= 0.0f, pz = inv.z >= 0.0f;
    const uint32_t nx = px ? lox : hix, fx = px ? hix : lox;
    const uint32_t ny = py ? loy : hiy, fy = py ? hiy : loy;
    const uint32_t nz = pz ? loz : hiz, fz = pz ? hiz : loz;
#define PTX_WIDE_CHILD(c)                                                                                              \
  {                                                                                                                    \
    const float tnx = __builtin_fmaf(ubf<c>(nx), ax, bx), tfx = __builtin_fmaf(ubf<c>(fx), ax, bx);                    \
    const float tny = __builtin_fmaf(ubf<c>(ny), ay, by), tfy = __builtin_fmaf(ubf<c>(fy), ay, by);                    \
    const float tnz = __builtin_fmaf(ubf<c>(nz), az, bz), tfz = __builtin_fmaf(ubf<c>(fz), az, bz);                    \
    const float tmin = vmax(vmax3(tnx, tny, tnz), 0.0f);                                                               \
    const float tmax = vmin(vmin3(tfx, tfy, tfz), rayT);                                                               \
    const bool h = tmin <= tmax; /* an empty slot (lo = 255, hi = 0) has entry > exit on every axis */                 \
    key[c] = h ? tmin : __builtin_inff();                                                                              \
    nHit += h ? 1u : 0u;                                                                                               \
  }
    PTX_WIDE_CHILD(0) PTX_WIDE_CHILD(1) PTX_WIDE_CHILD(2) PTX_WIDE_CHILD(3)
#undef PTX_WIDE_CHILD
  }
  s[0] = __float_as_uint(q3.x); s[1] = __float_as_uint(q3.y); s[2] = __float_as_uint(q3.z); s[3] = __float_as_uint(q3.w);
  // five-comparator sorting network on (key, reference); misses carry +inf and end up last
#define PTX_CSWAP(a, b)                                  \
  {                                                      \
    const bool sw = key[a] > key[b];                     \
    const float ka = key[a], kb = key[b];                \
    const uint32_t ra = s[a], rb = s[b];                 \
    key[a] = sw ? kb : ka; key[b] = sw ? ka : kb;        \
    s[a] = sw ? rb : ra;   s[b] = sw ? ra : rb;          \
  }
  PTX_CSWAP(0, 1) PTX_CSWAP(2, 3) PTX_CSWAP(0, 2) PTX_CSWAP(1, 3) PTX_CSWAP(1, 2)
#undef PTX_CSWAP
  return nHit;
}

// intersection.hpp:9-44 on a TriRec (edges precomputed on the host with the same subtractions).
// Written without early returns: every lane evaluates the whole test and the reference's five
// rejection conditions are combined at the end, each with the reference's own comparison (so a NaN
// from a degenerate determinant passes or fails exactly where it does there; a rejected lane's later
// values are never used).
__device__ __forceinline__ float triangleTest(const vec3& o, const vec3& d, const TriRec& tri, float& u, float& v) {
  const vec3 edge1 = f4xyz(tri.e1), edge2 = f4xyz(tri.e2);
  const vec3 h = pt::cross(d, edge2);
  const float a = pt::dot(edge1, h);
  const bool parallel = a > -1e-4f && a < 1e-4f;
  // f = 1.0f / a: lanes that go on have |a| >= 1e-4, so only the top of the range needs the guard
  // (a rejected lane's f is never used)
  float f;
  if (PTX_FAST_RCP_TRI && __all(parallel || fabsf(a) < kRcpHi)) f = rcpNewton(a);
  else f = 1.0f / a;
  const vec3 s = o - f4xyz(tri.v0);
  const float uu = f * pt::dot(s, h);
  const bool uOut = uu < 0.0f || uu > 1.0f;
  const vec3 q = pt::cross(s, edge1);
  const float vv = f * pt::dot(d, q);
  const bool vOut = vv < 0.0f || uu + vv > 1.0f;
  const float t = f * pt::dot(edge2, q);
  const bool ok = !parallel && !uOut && !vOut && t > 0.0f;
  u = uu;
  v = vv;
  return ok ? t : FLT_MAX;
}

// glm mat4 * vec4(v, w) restricted to xyz: (m0*x + m1*y) + (m2*z + m3*w) per row.
__device__ __forceinline__ vec3 xformRows(const float4& r0, const float4& r1, const float4& r2, const vec3& v, float w) {
  return vec3((r0.x * v.x + r0.y * v.y) + (r0.z * v.z + r0.w * w), (r1.x * v.x + r1.y * v.y) + (r1.z * v.z + r1.w * w),
              (r2.x * v.x + r2.y * v.y) + (r2.z * v.z + r2.w * w));
}

struct Hit {
  float t;        // FLT_MAX = miss
  float u, v;     // barycentrics (intersection.hpp:39)
  uint32_t tri;   // absolute triangle index
  uint32_t inst;  // instance index (0xFFFFFFFF = none)
};

}  // namespace ptx
