// Binned-SAH BVH builder (host).  Input producer of the hot path: the BLAS/TLAS node arrays it
// emits are what both the HIP kernels and the CPU oracle traverse, and the `-a` heat map counts
// depend on the exact tree, so this restates the reference builder's arithmetic and tie-breaks
// (bvh.hpp:10-200, bvh.cpp:3-7) -- 64 bins per axis, asymmetric plane cost 2*Nl*Al + Nr*Ar,
// first strictly lower cost wins, depth-first node numbering with adjacent children -- while
// being organised differently: one non-recursive builder over pre-extracted bounds/centroid
// arrays (shared by triangles and instances) with reusable bin scratch, instead of a recursive
// template that allocates six std::vectors per axis per node.
#pragma once

#include <memory>
#include <utility>
#include <vector>

#include "scene_types.hpp"

namespace pt {

struct BuildInput {
  const vec3* lo;        // per primitive: getMin()
  const vec3* hi;        // per primitive: getMax()
  const vec3* centroid;  // per primitive: centroid
  uint count;
};

struct BuildResult {
  std::vector<BVHNode> nodes;  // trimmed to nodesUsed (geometry.cpp:52 / scene.cpp:34)
  std::vector<uint> order;     // order[i] = original index of the primitive now at slot i
  uint maxDepth = 0;           // deepest node (root = 0); bounds the traversal stack
  uint maxLeafSize = 0;
};

namespace detail {

inline float nodeCost(const BVHNode& n) {  // bvh.cpp:3-7
  vec3 dim = n.aabb.max - n.aabb.min;
  float area = 2.0f * (dim.x * dim.y + dim.y * dim.z + dim.z * dim.x);
  return n.count * area;
}

inline void fitBounds(BVHNode& n, const BuildInput& in, const std::vector<uint>& idx) {  // bvh.hpp:10-21
  n.aabb.min = vec3(FLT_MAX);
  n.aabb.max = vec3(-FLT_MAX);
  for (uint i = 0; i < n.count; i++) {
    uint p = idx[n.leftFirst + i];
    n.aabb.min = gmin(n.aabb.min, in.lo[p]);
    n.aabb.max = gmax(n.aabb.max, in.hi[p]);
  }
}

struct SplitScratch {
  static constexpr uint kBins = 64;  // bvh.hpp:59
  AABB bounds[kBins];
  uint counts[kBins];
  float leftArea[kBins - 1], rightArea[kBins - 1];
  uint leftCount[kBins - 1], rightCount[kBins - 1];
};

// bvh.hpp:53-123.  Returns the best plane cost (FLT_MAX if every axis is flat).
inline float bestSplit(const BVHNode& n, const BuildInput& in, const std::vector<uint>& idx,
                       SplitScratch& s, uint& bestAxis, float& splitPos) {
  constexpr uint B = SplitScratch::kBins;
  float bestCost = FLT_MAX;
  for (uint axis = 0; axis < 3; axis++) {
    float lo = FLT_MAX, hi = -FLT_MAX;
    for (uint i = 0; i < n.count; i++) {
      float c = in.centroid[idx[n.leftFirst + i]][axis];
      lo = gmin(lo, c);
      hi = gmax(hi, c);
    }
    if (lo == hi) continue;  // flat in this dimension

    for (uint b = 0; b < B; b++) {
      s.bounds[b] = AABB{};
      s.counts[b] = 0;
    }
    const float binScale = (float)B / (hi - lo);
    for (uint i = 0; i < n.count; i++) {
      uint p = idx[n.leftFirst + i];
      uint b = (uint)gmin((float)B - 1.0f, floorf((in.centroid[p][axis] - lo) * binScale));
      s.counts[b]++;
      s.bounds[b].grow(in.lo[p]);
      s.bounds[b].grow(in.hi[p]);
    }

    AABB lbox, rbox;
    uint lsum = 0, rsum = 0;
    for (uint i = 0; i < B - 1; i++) {
      lsum += s.counts[i];
      s.leftCount[i] = lsum;
      lbox.grow(s.bounds[i]);
      s.leftArea[i] = lbox.area();
      rsum += s.counts[B - 1 - i];
      s.rightCount[B - 2 - i] = rsum;
      rbox.grow(s.bounds[B - 1 - i]);
      s.rightArea[B - 2 - i] = rbox.area();
    }

    const float slab = (hi - lo) / (float)B;
    for (uint i = 0; i < B - 1; i++) {
      // Deliberately asymmetric, as the reference (bvh.hpp:114).
      float cost = 2.0f * (float)s.leftCount[i] * s.leftArea[i] + (float)s.rightCount[i] * s.rightArea[i];
      if (cost < bestCost) {
        splitPos = lo + slab * (float)(i + 1u);
        bestAxis = axis;
        bestCost = cost;
      }
    }
  }
  return bestCost;
}

}  // namespace detail

// bvh.hpp:126-200.  Node numbering equals the reference's recursion: a node's two children take the
// next two free slots when the node is split, and the whole left subtree is split before the right.
inline BuildResult buildBVH(const BuildInput& in) {
  BuildResult out;
  const uint n = in.count;
  out.order.resize(n);
  for (uint i = 0; i < n; i++) out.order[i] = i;
  if (n == 0) return out;
  out.nodes.assign(2u * n - 1u, BVHNode{});
  std::vector<uint>& idx = out.order;

  uint used = 1;
  out.nodes[0].leftFirst = 0;
  out.nodes[0].count = n;
  detail::fitBounds(out.nodes[0], in, idx);

  auto scratch = std::make_unique<detail::SplitScratch>();
  std::vector<std::pair<uint, uint>> todo;  // (node, depth)
  todo.emplace_back(0u, 0u);
  while (!todo.empty()) {
    auto [ni, depth] = todo.back();
    todo.pop_back();
    BVHNode& node = out.nodes[ni];
    if (depth > out.maxDepth) out.maxDepth = depth;

    uint axis = 0;
    float pos = 0.0f;
    float cost = detail::bestSplit(node, in, idx, *scratch, axis, pos);
    if (cost >= detail::nodeCost(node)) {
      if (node.count > out.maxLeafSize) out.maxLeafSize = node.count;
      continue;
    }

    // Hoare-style partition, swapping with the back (bvh.hpp:144-156).
    uint i = node.leftFirst;
    uint j = i + node.count - 1;
    while (j < UINT_MAX && i <= j) {
      if (in.centroid[idx[i]][axis] < pos) {
        i++;
      } else {
        std::swap(idx[i], idx[j--]);
      }
    }
    uint leftCount = i - node.leftFirst;
    if (leftCount == 0 || leftCount == node.count) {
      if (node.count > out.maxLeafSize) out.maxLeafSize = node.count;
      continue;
    }

    uint l = used++;
    uint r = used++;
    out.nodes[l].leftFirst = node.leftFirst;
    out.nodes[l].count = leftCount;
    out.nodes[r].leftFirst = i;
    out.nodes[r].count = node.count - leftCount;
    node.leftFirst = l;
    node.count = 0;
    detail::fitBounds(out.nodes[l], in, idx);
    detail::fitBounds(out.nodes[r], in, idx);
    todo.emplace_back(r, depth + 1);  // popped after the whole left subtree
    todo.emplace_back(l, depth + 1);
  }
  out.nodes.resize(used);
  return out;
}

}  // namespace pt
