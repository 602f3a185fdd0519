// Tighter TLAS boxes for the radiance path (host side, ptx_create).
//
// The reference bounds an instance by the box of the eight transformed corners of its BLAS root box
// (mesh.cpp:18-41): for a rotated mesh that box has about twice the volume of the mesh's own world-space bounds
// (stand-in bust, scene 1: 0.47 on average), and on scene 1 -- 10,000 overlapping instances -- 70 % of all instance
// entries find nothing (profiles/r02/tlas_levels.txt).  The radiance path walks a re-layout of the reference's TLAS
// anyway (wide_bvh.hpp), so its boxes can be the tight ones: SAME tree, same leaves, same instances; every TLAS
// box is replaced by the intersection of the reference's box with the union of what lies below it, bottom-up from
// the exact bounds of every instance's transformed vertices.  A ray that hits a triangle of an instance passes
// through the tight box of that instance, so the hits -- and with them t, u, v -- are the reference's; what
// changes is how many boxes and instances a ray enters on the way.  The `-a` heat map counts the reference's
// tests and keeps the reference's boxes (NodePair layout, untouched).
// Cost: instances x triangles vertex transforms on the GPU (instanceBoundsKernel; scene 1: 1.5e10, 7.7 ms); scenes
// beyond 4e10 triangle-instance pairs use the corners of the BLAS boxes eight levels down instead of the vertices
// (ptx_backend.hip: deviceInstanceBounds).
#pragma once

#include <cfloat>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/ptx_backend.h"

namespace ptx {

struct Box3 {
  float mn[3], mx[3];
};

namespace tight_detail {

// inverse of the affine part of a column-major 4x4 (rows 0..2), in double: model = inverse(inv_model)
inline bool invertAffine(const float* m, double out[12]) {
  const double a = m[0], b = m[4], c = m[8], d = m[1], e = m[5], f = m[9], g = m[2], h = m[6], i = m[10];
  const double tx = m[12], ty = m[13], tz = m[14];
  const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
  const double r = 1.0 / det;
  const double A[9] = {(e * i - f * h) * r, (c * h - b * i) * r, (b * f - c * e) * r, (f * g - d * i) * r, (a * i - c * g) * r,
                       (c * d - a * f) * r, (d * h - e * g) * r, (b * g - a * h) * r, (a * e - b * d) * r};
  for (int k = 0; k < 9; k++) out[(k / 3) * 4 + (k % 3)] = A[k];
  for (int row = 0; row < 3; row++) out[row * 4 + 3] = -(A[row * 3] * tx + A[row * 3 + 1] * ty + A[row * 3 + 2] * tz);
  for (int k = 0; k < 12; k++)
    if (!std::isfinite(out[k])) return false;
  return true;
}

}  // namespace tight_detail

// Rounding allowance per instance and axis, added to the pad below: the bounds are computed with the fp32 model matrix M
// (the inverse of inv_model), and the device reaches the same points the other way round, through inv_model in fp32; the
// rounding error of either is proportional to |M| |v|, not to where the result lands.  When large object-space coordinates
// are cancelled by the translation (a mesh modelled far from its origin and moved back), the world-space result is small
// while that error is not, so a pad relative to the box alone could cut hits off.  allowance = 2^-20 x (|M0| |x|max +
// |M1| |y|max + |M2| |z|max + |M3|) over the geometry's object-space bounds (its BLAS root box).
inline void roundingAllowance(const ptx_scene_desc* d, std::vector<float>& out /* 3 per instance */) {
  out.assign((size_t)d->instance_count * 3u, 0.0f);
  for (uint32_t n = 0; n < d->instance_count; n++) {
    const ptx_instance& I = d->instances[n];
    double M[12];
    if (I.geometry_id >= d->geometry_count || !tight_detail::invertAffine(I.inv_model, M)) continue;
    const ptx_geometry& G = d->geometries[I.geometry_id];
    if (G.node_count == 0 || !d->blas_nodes) continue;
    const ptx_bvh_node& root = d->blas_nodes[G.node_offset];
    for (int a = 0; a < 3; a++) {
      double e = std::fabs(M[4 * a + 3]);
      for (int k = 0; k < 3; k++) e += std::fabs(M[4 * a + k]) * std::fmax(std::fabs((double)root.aabb_min[k]), std::fabs((double)root.aabb_max[k]));
      const double allow = std::ldexp(e, -20);
      out[3 * (size_t)n + a] = std::isfinite(allow) && allow < (double)FLT_MAX ? (float)allow : FLT_MAX;
    }
  }
}

// Raw bounds (min, max of the transformed vertices, from instanceBoundsKernel or from the host loop below) ->
// boxes padded outwards: by 2^-16 of the box's size and of its distance from the origin -- three orders of magnitude above
// the fp32 difference for a mesh modelled around its origin, far below the 1/255 steps of the wide layout -- plus the
// rounding allowance above.  An instance without usable bounds (matrix not invertible, no triangles, non-finite result)
// keeps an unbounded box: it is never tightened.
inline void padBounds(const float* raw /* 6 per instance */, const std::vector<uint8_t>& usable, const std::vector<float>& allowance,
                      std::vector<Box3>& out) {
  out.resize(usable.size());
  for (size_t n = 0; n < usable.size(); n++) {
    Box3& B = out[n];
    for (int a = 0; a < 3; a++) {
      B.mn[a] = -FLT_MAX;
      B.mx[a] = FLT_MAX;
    }
    if (!usable[n]) continue;
    const float* lo3 = raw + 6 * n;
    const float* hi3 = lo3 + 3;
    bool ok = true;
    for (int a = 0; a < 3; a++) ok = ok && std::isfinite(lo3[a]) && std::isfinite(hi3[a]) && lo3[a] <= hi3[a];
    if (!ok) continue;
    for (int a = 0; a < 3; a++) {
      const float pad = std::ldexp(std::fmax(hi3[a] - lo3[a], std::fmax(std::fabs(lo3[a]), std::fabs(hi3[a]))), -16) + FLT_MIN +
                        (allowance.size() > 3 * n + a ? allowance[3 * n + a] : 0.0f);
      const float lo = lo3[a] - pad, hi = hi3[a] + pad;
      if (std::isfinite(lo) && std::isfinite(hi)) {
        B.mn[a] = lo;
        B.mx[a] = hi;
      }
    }
  }
}

// The bounds of instance n's transformed vertices in double, through the double inverse of inv_model: the points the
// device's object-space test really stands for (ptx_check_tight_tlas holds the padded fp32 boxes against them).
inline bool exactInstanceBounds(const ptx_scene_desc* d, uint32_t n, double lo[3], double hi[3]) {
  const ptx_instance& I = d->instances[n];
  double M[12];
  if (!tight_detail::invertAffine(I.inv_model, M)) return false;
  const ptx_geometry& G = d->geometries[I.geometry_id];
  if (G.tri_count == 0) return false;
  for (int a = 0; a < 3; a++) { lo[a] = DBL_MAX; hi[a] = -DBL_MAX; }
  const float* v = d->triangles[G.tri_offset].v0;
  const size_t nv = (size_t)G.tri_count * 3u;
  for (size_t k = 0; k < nv; k++)
    for (int a = 0; a < 3; a++) {
      const double w = M[4 * a] * v[3 * k] + M[4 * a + 1] * v[3 * k + 1] + M[4 * a + 2] * v[3 * k + 2] + M[4 * a + 3];
      lo[a] = w < lo[a] ? w : lo[a];
      hi[a] = w > hi[a] ? w : hi[a];
    }
  return true;
}

// Host version of instanceBoundsKernel (ptx_check_tight_tlas: the device's bounds are checked against it).
inline void rawInstanceBoundsHost(const ptx_scene_desc* d, uint32_t lo, uint32_t hi, float* raw, std::vector<uint8_t>& usable) {
  for (uint32_t n = lo; n < hi; n++) {
    const ptx_instance& I = d->instances[n];
    double M[12];
    usable[n] = 0;
    if (!tight_detail::invertAffine(I.inv_model, M)) continue;
    const ptx_geometry& G = d->geometries[I.geometry_id];
    if (G.tri_count == 0) continue;
    float m[12];
    for (int k = 0; k < 12; k++) m[k] = (float)M[k];
    float l[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, h[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const float* v = d->triangles[G.tri_offset].v0;  // 9 floats per triangle, 3 vertices
    const size_t nv = (size_t)G.tri_count * 3u;
    for (size_t k = 0; k < nv; k++) {
      const float x = v[3 * k], y = v[3 * k + 1], z = v[3 * k + 2];
      for (int a = 0; a < 3; a++) {
        const float w = m[4 * a] * x + m[4 * a + 1] * y + m[4 * a + 2] * z + m[4 * a + 3];
        l[a] = w < l[a] ? w : l[a];
        h[a] = w > h[a] ? w : h[a];
      }
    }
    for (int a = 0; a < 3; a++) {
      raw[6 * n + a] = l[a];
      raw[6 * n + 3 + a] = h[a];
    }
    usable[n] = 1;
  }
}

// The reference's TLAS with every box replaced by (its own box) intersected with (the union of what lies below it).
// The links must have been validated (convertTree).  Returns the mean ratio new / old box volume over the leaves
// in *leafVolumeRatio (diagnostics).
inline std::vector<ptx_bvh_node> refitTlas(const ptx_scene_desc* d, const std::vector<Box3>& inst, double* leafVolumeRatio) {
  const uint32_t n = d->tlas_node_count;
  std::vector<ptx_bvh_node> out(d->tlas_nodes, d->tlas_nodes + n);
  // post-order without recursion: children before parents
  std::vector<uint32_t> order;
  order.reserve(n);
  std::vector<uint32_t> todo(1, 0u);
  while (!todo.empty()) {
    const uint32_t i = todo.back();
    todo.pop_back();
    order.push_back(i);
    if (out[i].count == 0) {
      todo.push_back(out[i].left_first);
      todo.push_back(out[i].left_first + 1u);
    }
  }
  double ratioSum = 0.0;
  uint64_t leaves = 0;
  for (size_t k = order.size(); k-- > 0;) {
    ptx_bvh_node& nd = out[order[k]];
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (nd.count > 0) {
      for (uint32_t j = 0; j < nd.count; j++) {
        const Box3& b = inst[nd.left_first + j];
        for (int a = 0; a < 3; a++) {
          mn[a] = b.mn[a] < mn[a] ? b.mn[a] : mn[a];
          mx[a] = b.mx[a] > mx[a] ? b.mx[a] : mx[a];
        }
      }
    } else {
      for (uint32_t c = 0; c < 2; c++) {
        const ptx_bvh_node& ch = out[nd.left_first + c];
        for (int a = 0; a < 3; a++) {
          mn[a] = ch.aabb_min[a] < mn[a] ? ch.aabb_min[a] : mn[a];
          mx[a] = ch.aabb_max[a] > mx[a] ? ch.aabb_max[a] : mx[a];
        }
      }
    }
    double vOld = 1.0, vNew = 1.0;
    for (int a = 0; a < 3; a++) {
      vOld *= (double)nd.aabb_max[a] - (double)nd.aabb_min[a];
      // never larger than the reference's box, never inverted
      const float lo = mn[a] > nd.aabb_min[a] ? mn[a] : nd.aabb_min[a];
      const float hi = mx[a] < nd.aabb_max[a] ? mx[a] : nd.aabb_max[a];
      if (lo <= hi) {
        nd.aabb_min[a] = lo;
        nd.aabb_max[a] = hi;
      }
      vNew *= (double)nd.aabb_max[a] - (double)nd.aabb_min[a];
    }
    if (nd.count > 0 && vOld > 0.0) {
      ratioSum += vNew / vOld;
      leaves++;
    }
  }
  if (leafVolumeRatio) *leafVolumeRatio = leaves ? ratioSum / (double)leaves : 1.0;
  return out;
}

}  // namespace ptx
