// Four-wide, quantised re-layout of the reference's binary BVHs for the radiance path (host side).
//
// Why: the traversal loop is bound by 16-byte lane-accesses to the vector memory path,
// and a NodePair costs four of them for two children.  A WideNode also costs four, for up to FOUR
// children: the same tree with every other level folded away (a child is replaced by its own two
// children, largest box first, while there is room), child boxes stored as 8-bit offsets in a frame
// spanned by the parent box.  Quantised boxes are rounded outwards, so a ray visits a superset of the
// leaves the binary tree would hand it; leaves, triangles and the Moeller-Trumbore arithmetic are
// untouched, hence the same hits with the same t/u/v.  (A ray that the exact fp32 slab test lets miss
// a box by one rounding while its triangle test would accept can now find that hit; the radiance
// tolerance covers it.)  The `-a` heat map counts tests on the BINARY tree and never uses this.
//
//   q0  origin.xyz (parent box min), w = exponent bytes ex | ey << 8 | ez << 16 (scale_a = 2^(e_a - 127))
//       | number of children << 24
//   q1  x: lo.x of children 0..3 (one byte each), y: lo.y, z: lo.z, w: hi.x
//   q2  x: hi.y, y: hi.z, z, w: unused
//   q3  the four child references (same encoding as everywhere); an empty slot has lo = 255, hi = 0
//   child plane = origin + byte * scale, evaluated on the device as one fma per plane
#pragma once

#include <cmath>
#include <string>
#include <vector>

#include "../../include/ptx_backend.h"
#include "device_scene.hpp"

namespace ptx {

struct WideNode {  // 64 B
  float4 q0;
  uint4 q1, q2, q3;
};

struct WideInfo {
  uint32_t rootRef = 0;
  uint32_t depth = 0;  // levels of wide nodes on the longest path
};

namespace wide_detail {

inline float halfArea(const ptx_bvh_node& n) {
  const float x = n.aabb_max[0] - n.aabb_min[0], y = n.aabb_max[1] - n.aabb_min[1], z = n.aabb_max[2] - n.aabb_min[2];
  const float a = x * y + y * z + z * x;
  return a > 0.0f ? a : 0.0f;
}

// Smallest exponent byte (scale = 2^(byte-127)) with 255 * scale >= extent, and outward-rounded bytes.
inline bool quantiseAxis(float origin, float extent, const float* lo, const float* hi, uint32_t n, uint32_t& expByte, uint8_t* qlo,
                         uint8_t* qhi) {
  if (!(extent >= 0.0f) || std::isinf(extent) || std::isnan(origin) || std::isinf(origin)) return false;
  int e = -126;
  if (extent > 0.0f) {
    e = (int)std::ceil(std::log2((double)extent / 255.0));
    if (e < -126) e = -126;
  }
  for (; e <= 127; e++) {
    const float scale = std::ldexp(1.0f, e);
    bool ok = true;
    for (uint32_t c = 0; c < n && ok; c++) {
      double ql = std::floor(((double)lo[c] - (double)origin) / (double)scale);
      double qh = std::ceil(((double)hi[c] - (double)origin) / (double)scale);
      if (ql < 0.0) ql = 0.0;
      if (qh < 0.0) qh = 0.0;
      if (ql > 255.0) ql = 255.0;
      int l = (int)ql, h = (int)(qh > 300.0 ? 300.0 : qh);
      // the device evaluates fmaf(byte, scale, origin): make sure rounding kept the box covered
      while (l > 0 && std::fmaf((float)l, scale, origin) > lo[c]) l--;
      while (h <= 255 && std::fmaf((float)h, scale, origin) < hi[c]) h++;
      if (h > 255 || std::fmaf((float)l, scale, origin) > lo[c]) {
        ok = false;
        break;
      }
      qlo[c] = (uint8_t)l;
      qhi[c] = (uint8_t)h;
    }
    if (ok) {
      expByte = (uint32_t)(e + 127);
      return true;
    }
  }
  return false;
}

}  // namespace wide_detail

// Builds the wide nodes of one reference-layout tree, appended to `out`.  leafRef(node) must return the
// reference of a leaf node (the caller owns the encoding and the limits, as for the binary layout).
// Returns an empty string on success, else an error text.
template <typename LeafRef>
std::string buildWideTree(const ptx_bvh_node* nodes, uint32_t n, LeafRef leafRef, std::vector<WideNode>& out, WideInfo& info) {
  using namespace wide_detail;
  info.depth = 0;
  if (nodes[0].count > 0) {
    info.rootRef = leafRef(nodes[0]);
    return "";
  }
  struct Todo {
    uint32_t node, wideIndex, depth;
  };
  std::vector<Todo> todo;
  auto reserve = [&](uint32_t node, uint32_t depth) {
    const uint32_t w = (uint32_t)out.size();
    out.emplace_back();
    todo.push_back({node, w, depth});
    return w;
  };
  if (out.size() >= kLeafFlag) return "too many wide BVH nodes";
  const size_t firstWide = out.size();
  info.rootRef = reserve(0, 1);
  while (!todo.empty()) {
    const Todo t = todo.back();
    todo.pop_back();
    if (out.size() - firstWide > n) return "malformed node links (cycle)";  // a tree has fewer wide nodes than nodes
    if (t.depth > info.depth) info.depth = t.depth;
    const ptx_bvh_node& parent = nodes[t.node];
    // fold: start from the two children, replace the internal child with the largest box by its own
    // children while a slot is free
    uint32_t kids[4] = {parent.left_first, parent.left_first + 1u, 0, 0};
    uint32_t nk = 2;
    while (nk < 4) {
      int best = -1;
      float bestArea = -1.0f;
      for (uint32_t c = 0; c < nk; c++)
        if (nodes[kids[c]].count == 0 && halfArea(nodes[kids[c]]) > bestArea) {
          bestArea = halfArea(nodes[kids[c]]);
          best = (int)c;
        }
      if (best < 0) break;
      const uint32_t l = nodes[kids[best]].left_first;
      if (l == 0 || l + 1u >= n) return "malformed node links";
      kids[best] = l;
      kids[nk++] = l + 1u;
    }
    float lo[3][4], hi[3][4];
    for (uint32_t c = 0; c < nk; c++)
      for (int a = 0; a < 3; a++) {
        lo[a][c] = nodes[kids[c]].aabb_min[a];
        hi[a][c] = nodes[kids[c]].aabb_max[a];
      }
    WideNode w;
    uint32_t expBytes = 0;
    uint8_t ql[3][4], qh[3][4];
    float origin[3];
    for (int a = 0; a < 3; a++) {
      // frame: the union of the children (equals the parent's box in a reference tree; recomputed so a
      // caller's loose parent box cannot break the coverage)
      float mn = lo[a][0], mx = hi[a][0];
      for (uint32_t c = 1; c < nk; c++) {
        mn = lo[a][c] < mn ? lo[a][c] : mn;
        mx = hi[a][c] > mx ? hi[a][c] : mx;
      }
      origin[a] = mn;
      for (uint32_t c = nk; c < 4; c++) {
        ql[a][c] = 255;
        qh[a][c] = 0;
      }
      uint32_t eb = 0;
      if (!quantiseAxis(mn, mx - mn, lo[a], hi[a], nk, eb, ql[a], qh[a])) return "a BVH box cannot be quantised (non-finite or inverted bounds)";
      expBytes |= eb << (8 * a);
    }
    expBytes |= nk << 24;
    auto pack = [](const uint8_t* b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); };
    w.q0 = make_float4(origin[0], origin[1], origin[2], __builtin_bit_cast(float, expBytes));
    w.q1 = make_uint4(pack(ql[0]), pack(ql[1]), pack(ql[2]), pack(qh[0]));
    w.q2 = make_uint4(pack(qh[1]), pack(qh[2]), 0u, 0u);
    uint32_t refs[4] = {kRefIdle, kRefIdle, kRefIdle, kRefIdle};  // never followed: an empty slot cannot be hit
    for (uint32_t c = 0; c < nk; c++) {
      if (nodes[kids[c]].count > 0) {
        refs[c] = leafRef(nodes[kids[c]]);
      } else {
        if (out.size() >= kLeafFlag) return "too many wide BVH nodes";
        refs[c] = reserve(kids[c], t.depth + 1u);
      }
    }
    w.q3 = make_uint4(refs[0], refs[1], refs[2], refs[3]);
    out[t.wideIndex] = w;
  }
  return "";
}

}  // namespace ptx
