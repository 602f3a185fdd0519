// Binned-SAH BVH builder (host).  Input producer of the hot path: the BLAS/TLAS node arrays it
// emits are what both the HIP kernels and the CPU oracle traverse, and the `-a` heat map counts
// depend on the exact tree, so this restates the reference builder's arithmetic and tie-breaks
// (bvh.hpp:10-200, bvh.cpp:3-7) -- 64 bins per axis, asymmetric plane cost 2*Nl*Al + Nr*Ar,
// first strictly lower cost wins, depth-first node numbering with adjacent children -- while
// being organised differently: one non-recursive builder over pre-extracted bounds/centroid
// arrays (shared by triangles and instances) with reusable bin scratch, instead of a recursive
// template that allocates six std::vectors per axis per node.
#pragma once

#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
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

namespace detail {

// The split search over primitives [first, first+count) of one node with the per-primitive passes
// (centroid range, binning) cut into chunks that run on several threads.  Every reduction in it is a
// float min / max or an integer sum, all exact and order-independent, so the chosen plane is the one
// bestSplit finds.
class WorkerPool {
 public:
  explicit WorkerPool(uint threads) : n_(threads < 1 ? 1 : threads) {
    for (uint t = 1; t < n_; t++) workers_.emplace_back([this, t] { loop(t); });
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
      generation_++;
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  uint size() const { return n_; }
  // runs job(t) for t = 0..size()-1 (t = 0 on the calling thread) and returns when all are done
  void run(const std::function<void(uint)>& job) {
    {
      std::lock_guard<std::mutex> g(m_);
      job_ = &job;
      pending_ = n_ - 1;
      generation_++;
    }
    cv_.notify_all();
    job(0);
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [this] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  void loop(uint t) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(uint)>* job;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return generation_ != seen; });
        seen = generation_;
        if (stop_) return;
        job = job_;
      }
      (*job)(t);
      {
        std::lock_guard<std::mutex> g(m_);
        if (--pending_ == 0) done_.notify_one();
      }
    }
  }
  uint n_;
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(uint)>* job_ = nullptr;
  uint pending_ = 0;
  uint64_t generation_ = 0;
  bool stop_ = false;
};

inline float bestSplitParallel(const BVHNode& n, const BuildInput& in, const std::vector<uint>& idx, WorkerPool& pool,
                               std::vector<SplitScratch>& perThread, uint& bestAxis, float& splitPos) {
  constexpr uint B = SplitScratch::kBins;
  const uint T = pool.size();
  float bestCost = FLT_MAX;
  std::vector<float> los(T), his(T);
  for (uint axis = 0; axis < 3; axis++) {
    pool.run([&](uint t) {
      const uint a = (uint)((uint64_t)n.count * t / T), b = (uint)((uint64_t)n.count * (t + 1) / T);
      float lo = FLT_MAX, hi = -FLT_MAX;
      for (uint i = a; i < b; i++) {
        float c = in.centroid[idx[n.leftFirst + i]][axis];
        lo = gmin(lo, c);
        hi = gmax(hi, c);
      }
      los[t] = lo;
      his[t] = hi;
    });
    float lo = FLT_MAX, hi = -FLT_MAX;
    for (uint t = 0; t < T; t++) {
      lo = gmin(lo, los[t]);
      hi = gmax(hi, his[t]);
    }
    if (lo == hi) continue;
    const float binScale = (float)B / (hi - lo);
    pool.run([&](uint t) {
      SplitScratch& s = perThread[t];
      for (uint b = 0; b < B; b++) {
        s.bounds[b] = AABB{};
        s.counts[b] = 0;
      }
      const uint a = (uint)((uint64_t)n.count * t / T), e = (uint)((uint64_t)n.count * (t + 1) / T);
      for (uint i = a; i < e; i++) {
        uint p = idx[n.leftFirst + i];
        uint b = (uint)gmin((float)B - 1.0f, floorf((in.centroid[p][axis] - lo) * binScale));
        s.counts[b]++;
        s.bounds[b].grow(in.lo[p]);
        s.bounds[b].grow(in.hi[p]);
      }
    });
    SplitScratch& s = perThread[0];
    for (uint t = 1; t < T; t++)
      for (uint b = 0; b < B; b++) {
        s.counts[b] += perThread[t].counts[b];
        s.bounds[b].grow(perThread[t].bounds[b]);
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

namespace detail {
// One split step of the builder on `node` (bounds already fitted).  Returns false if the node stays a
// leaf; otherwise partitions idx and fills the two children (not yet numbered).
template <typename Search>
inline bool splitNode(BVHNode& node, const BuildInput& in, std::vector<uint>& idx, Search&& search, BVHNode& left, BVHNode& right) {
  uint axis = 0;
  float pos = 0.0f;
  float cost = search(node, axis, pos);
  if (cost >= nodeCost(node)) return false;
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
  if (leftCount == 0 || leftCount == node.count) return false;
  left = BVHNode{};
  right = BVHNode{};
  left.leftFirst = node.leftFirst;
  left.count = leftCount;
  right.leftFirst = i;
  right.count = node.count - leftCount;
  fitBounds(left, in, idx);
  fitBounds(right, in, idx);
  return true;
}

// The serial builder on the sub-problem rooted at `root` (a copy of the subtree's root node): local
// numbering, nodes[0] = root, children of a split node at the next two free slots, left subtree first.
struct Subtree {
  std::vector<BVHNode> nodes;
  uint maxDepth = 0, maxLeafSize = 0;
};
inline void buildSubtree(const BVHNode& root, const BuildInput& in, std::vector<uint>& idx, SplitScratch& scratch, Subtree& out) {
  out.nodes.assign(2u * root.count - 1u, BVHNode{});
  out.nodes[0] = root;
  uint used = 1;
  std::vector<std::pair<uint, uint>> todo;
  todo.emplace_back(0u, 0u);
  while (!todo.empty()) {
    auto [ni, depth] = todo.back();
    todo.pop_back();
    if (depth > out.maxDepth) out.maxDepth = depth;
    BVHNode l, r;
    BVHNode& node = out.nodes[ni];
    if (!splitNode(node, in, idx, [&](const BVHNode& nd, uint& axis, float& pos) { return bestSplit(nd, in, idx, scratch, axis, pos); }, l, r)) {
      if (node.count > out.maxLeafSize) out.maxLeafSize = node.count;
      continue;
    }
    const uint li = used++, ri = used++;
    out.nodes[li] = l;
    out.nodes[ri] = r;
    node.leftFirst = li;
    node.count = 0;
    todo.emplace_back(ri, depth + 1);
    todo.emplace_back(li, depth + 1);
  }
  out.nodes.resize(used);
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

// The same tree, the same node numbering and the same primitive order as buildBVH, built on several
// threads (the `-a` contract depends on the exact tree, so only work whose result does
// not depend on the order of execution is distributed):
//   1. the top of the tree is split by one driver in the reference's order; for its large nodes the
//      per-primitive passes of the split search run on all threads (bestSplitParallel);
//   2. every subtree of at most `grain` primitives is an independent task on its own slice of the
//      primitive order: the serial builder with local numbering (buildSubtree);
//   3. numbering: in the reference's depth-first allocation all descendants of a node take consecutive
//      slots starting where the allocation stood when the node was reached, in the subtree's own local
//      order -- so a finished subtree is copied to its place with one offset.
inline BuildResult buildBVHParallel(const BuildInput& in, uint threads) {
  const uint n = in.count;
  if (threads < 3 || n < 32768u) return buildBVH(in);  // two threads do not repay the top-level hand-offs
  BuildResult out;
  out.order.resize(n);
  for (uint i = 0; i < n; i++) out.order[i] = i;
  std::vector<uint>& idx = out.order;
  detail::WorkerPool pool(threads);
  std::vector<detail::SplitScratch> scratch(threads);
  const uint grain = std::max(4096u, n / (8u * threads));
  const uint parallelSearchFrom = 4u * grain;

  // top structure: nodes in creation order, children by index into `top`
  struct TopNode {
    BVHNode node;
    int left = -1, right = -1;  // children in `top` (internal) ...
    int task = -1;              // ... or the task that builds this node's subtree
    uint depth = 0;
  };
  std::vector<TopNode> top;
  std::vector<int> taskNode;  // task -> top node
  top.emplace_back();
  top[0].node.leftFirst = 0;
  top[0].node.count = n;
  detail::fitBounds(top[0].node, in, idx);
  std::vector<int> todo{0};
  while (!todo.empty()) {
    const int ti = todo.back();
    todo.pop_back();
    if (top[ti].node.count <= grain) {
      top[ti].task = (int)taskNode.size();
      taskNode.push_back(ti);
      continue;
    }
    BVHNode l, r;
    BVHNode nodeCopy = top[ti].node;
    bool split;
    if (nodeCopy.count >= parallelSearchFrom)
      split = detail::splitNode(nodeCopy, in, idx, [&](const BVHNode& nd, uint& axis, float& pos) { return detail::bestSplitParallel(nd, in, idx, pool, scratch, axis, pos); }, l, r);
    else
      split = detail::splitNode(nodeCopy, in, idx, [&](const BVHNode& nd, uint& axis, float& pos) { return detail::bestSplit(nd, in, idx, scratch[0], axis, pos); }, l, r);
    if (!split) continue;  // a large leaf
    const int li = (int)top.size(), ri = li + 1;
    top.emplace_back();
    top.emplace_back();
    top[li].node = l;
    top[ri].node = r;
    top[li].depth = top[ri].depth = top[ti].depth + 1;
    top[ti].left = li;
    top[ti].right = ri;
    todo.push_back(ri);
    todo.push_back(li);
  }

  // tasks
  std::vector<detail::Subtree> sub(taskNode.size());
  std::atomic<uint> nextTask{0};
  pool.run([&](uint t) {
    for (;;) {
      const uint k = nextTask.fetch_add(1);
      if (k >= taskNode.size()) break;
      detail::buildSubtree(top[taskNode[k]].node, in, idx, scratch[t], sub[k]);
    }
  });

  // numbering in the reference's depth-first allocation order
  out.nodes.assign(2u * n - 1u, BVHNode{});
  uint used = 1;
  std::vector<std::pair<int, uint>> walk{{0, 0u}};  // (top node, its slot)
  while (!walk.empty()) {
    auto [ti, slot] = walk.back();
    walk.pop_back();
    const TopNode& tn = top[ti];
    if (tn.task >= 0) {
      const detail::Subtree& st = sub[tn.task];
      const uint base = used;  // local index k >= 1 -> slot base + k - 1
      auto moved = [&](BVHNode nd) {
        if (nd.count == 0) nd.leftFirst = base + nd.leftFirst - 1u;
        return nd;
      };
      out.nodes[slot] = moved(st.nodes[0]);
      for (size_t k = 1; k < st.nodes.size(); k++) out.nodes[base + k - 1u] = moved(st.nodes[k]);
      used += (uint)st.nodes.size() - 1u;
      if (tn.depth + st.maxDepth > out.maxDepth) out.maxDepth = tn.depth + st.maxDepth;
      if (st.maxLeafSize > out.maxLeafSize) out.maxLeafSize = st.maxLeafSize;
      continue;
    }
    out.nodes[slot] = tn.node;
    if (tn.depth > out.maxDepth) out.maxDepth = tn.depth;
    if (tn.left < 0) {
      if (tn.node.count > out.maxLeafSize) out.maxLeafSize = tn.node.count;
      continue;
    }
    const uint l = used++, r = used++;
    out.nodes[slot].leftFirst = l;
    out.nodes[slot].count = 0;
    walk.emplace_back(tn.right, r);  // after the whole left subtree
    walk.emplace_back(tn.left, l);
  }
  out.nodes.resize(used);
  return out;
}

}  // namespace pt
