// Ray traversal for one wave64 lane: TLAS -> instance transform -> BLAS -> Moeller-Trumbore.
//
// Behaviour contract (reference intersection.hpp:9-233, gpuFunctions.hpp:32-41): binary BVH, both
// children slab-tested per internal step, nearer child first (ties keep the left child), the farther
// child is pushed if hit and is NOT re-tested when popped, `count` += 1 per child box hit; triangles
// of a leaf are tested in order and replace the hit on strict `<`; instances of a TLAS leaf are
// entered in order with the ray's t carried across them.  The `-a` heat map is bit-exact against the
// CPU path only if visit order, ray.t at every slab test and every rounding are identical, so the
// arithmetic below keeps the reference's operation order (no FMA contraction: -ffp-contract=off) and
// GLM's min/max NaN behaviour.
//
// Organisation (this is not the reference's two nested loops with two private stacks): one loop, one
// stack.  TLAS and BLAS nodes are the same 64-byte NodePair records in one array, so the common step
// -- fetch a pair, two slab tests, order, push -- is a single code path that lanes in the TLAS and
// lanes inside an instance execute together; entering an instance pushes a marker and swaps the
// lane's ray to object space, popping the marker swaps it back.
#pragma once

#include "../host/ptmath.hpp"
#include "device_scene.hpp"

namespace ptx {

using pt::vec3;

struct Hit {
  float t;        // FLT_MAX = miss
  float u, v;     // barycentrics (intersection.hpp:39)
  uint32_t tri;   // absolute triangle index
  uint32_t inst;  // instance index (UINT_MAX = none)
};

struct TraceCounters {
  uint32_t nodeSteps, triTests, instEntries;
};

__device__ __forceinline__ vec3 f4xyz(const float4& q) { return vec3(q.x, q.y, q.z); }

// intersection.hpp:48-75 with glm::min/max semantics (pt::gmin/gmax).
__device__ __forceinline__ float slabTest(const vec3& o, const vec3& inv, float rayT, const float4& bmin,
                                          const float4& bmax) {
  float t1x = (bmin.x - o.x) * inv.x, t1y = (bmin.y - o.y) * inv.y, t1z = (bmin.z - o.z) * inv.z;
  float t2x = (bmax.x - o.x) * inv.x, t2y = (bmax.y - o.y) * inv.y, t2z = (bmax.z - o.z) * inv.z;
  float tmin = pt::gmax(pt::gmax(pt::gmin(t1x, t2x), pt::gmin(t1y, t2y)), pt::gmin(t1z, t2z));
  float tmax = pt::gmin(pt::gmin(pt::gmax(t1x, t2x), pt::gmax(t1y, t2y)), pt::gmax(t1z, t2z));
  return (tmax >= tmin && tmin < rayT && tmax > 0.0f) ? tmin : FLT_MAX;
}

// intersection.hpp:9-44 on a TriRec (edges precomputed, same values).
__device__ __forceinline__ float triangleTest(const vec3& o, const vec3& d, const TriRec& tri, float& u, float& v) {
  const vec3 edge1 = f4xyz(tri.e1), edge2 = f4xyz(tri.e2);
  const vec3 h = pt::cross(d, edge2);
  const float a = pt::dot(edge1, h);
  if (a > -1e-4f && a < 1e-4f) return FLT_MAX;
  const float f = 1.0f / a;
  const vec3 s = o - f4xyz(tri.v0);
  const float uu = f * pt::dot(s, h);
  if (uu < 0.0f || uu > 1.0f) return FLT_MAX;
  const vec3 q = pt::cross(s, edge1);
  const float vv = f * pt::dot(d, q);
  if (vv < 0.0f || uu + vv > 1.0f) return FLT_MAX;
  const float t = f * pt::dot(edge2, q);
  if (t > 0.0f) {
    u = uu;
    v = vv;
    return t;
  }
  return FLT_MAX;
}

// glm mat4 * vec4(v, w) restricted to xyz: (m0*x + m1*y) + (m2*z + m3*w) per row.
__device__ __forceinline__ vec3 xformRows(const float4& r0, const float4& r1, const float4& r2, const vec3& v, float w) {
  return vec3((r0.x * v.x + r0.y * v.y) + (r0.z * v.z + r0.w * w), (r1.x * v.x + r1.y * v.y) + (r1.z * v.z + r1.w * w),
              (r2.x * v.x + r2.y * v.y) + (r2.z * v.z + r2.w * w));
}

// Scene::intersect (scene.hpp:37-39).  `count` accumulates the reference's box-hit counter.
template <bool STATS>
__device__ __forceinline__ Hit traceRay(const DeviceScene& sc, const vec3& worldO, const vec3& worldD, const vec3& worldInv,
                                        uint32_t& count, TraceCounters& tc) {
  uint32_t stack[kStackEntries];
  uint32_t sp = 0;
  stack[sp++] = kRefDone;

  Hit hit;
  hit.t = FLT_MAX;
  hit.u = 0.0f;
  hit.v = 0.0f;
  hit.tri = 0xFFFFFFFFu;
  hit.inst = 0xFFFFFFFFu;

  vec3 o = worldO, d = worldD, inv = worldInv;  // the lane's current ray: world space or object space
  uint32_t ref = sc.tlasRootRef;
  uint32_t tlasLeaf = 0;      // remaining instances of the TLAS leaf being processed (leaf ref) or 0
  uint32_t curInst = 0;       // instance being traversed
  bool inBlas = false;
  float tBeforeInst = FLT_MAX;

  for (;;) {
    if (ref == kRefDone) break;

    if (ref == kRefInstExit) {
      // GPUMesh::intersect returns: ray.t = transformedRay.t; intersectTLAS keeps the instance if it
      // lowered t (intersection.hpp:184-189).
      if (hit.t < tBeforeInst) hit.inst = curInst;
      o = worldO;
      d = worldD;
      inv = worldInv;
      inBlas = false;
      ref = tlasLeaf ? tlasLeaf : stack[--sp];
      tlasLeaf = 0;
      continue;
    }

    if (ref & kLeafFlag) {
      const uint32_t first = ref & 0x00FFFFFFu;
      const uint32_t n = ((ref >> 24) & 0x7Fu) + 1u;
      if (!inBlas) {
        // TLAS leaf: enter instance `first`, remember the rest (gpuFunctions.hpp:32-41).
        tlasLeaf = (n > 1u) ? makeLeafRef(first + 1u, n - 1u) : 0u;
        curInst = first;
        const InstXform x = sc.instXform[first];
        o = xformRows(x.row0, x.row1, x.row2, worldO, 1.0f);
        d = xformRows(x.row0, x.row1, x.row2, worldD, 0.0f);  // not normalised: t stays world-space
        inv = 1.0f / d;
        tBeforeInst = hit.t;
        inBlas = true;
        if (STATS) tc.instEntries++;
        stack[sp++] = kRefInstExit;
        ref = x.rootRef;
        continue;
      }
      // BLAS leaf (intersection.hpp:112-122)
      for (uint32_t i = 0; i < n; i++) {
        const TriRec tri = sc.tris[first + i];
        float u = 0.0f, v = 0.0f;
        const float dist = triangleTest(o, d, tri, u, v);
        if (dist < hit.t) {
          hit.t = dist;
          hit.u = u;
          hit.v = v;
          hit.tri = first + i;
        }
      }
      if (STATS) tc.triTests += n;
      ref = stack[--sp];
      continue;
    }

    // Internal node (intersection.hpp:133-161 / :201-230): one 64-byte record holds both children.
    const NodePair p = sc.pairs[ref];
    float d1 = slabTest(o, inv, hit.t, p.lmin, p.lmax);
    float d2 = slabTest(o, inv, hit.t, p.rmin, p.rmax);
    uint32_t r1 = __float_as_uint(p.lmin.w), r2 = __float_as_uint(p.rmin.w);
    if (STATS) tc.nodeSteps++;
    if (d1 > d2) {
      float td = d1; d1 = d2; d2 = td;
      uint32_t tr = r1; r1 = r2; r2 = tr;
    }
    if (d1 == FLT_MAX) {
      ref = stack[--sp];
      continue;
    }
    ref = r1;
    count++;
    if (d2 != FLT_MAX) {
      count++;
      stack[sp++] = r2;
    }
  }
  return hit;
}

}  // namespace ptx
