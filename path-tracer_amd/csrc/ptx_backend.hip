// C ABI of the gfx950 path-tracing backend (include/ptx_backend.h): scene validation and re-layout,
// one-arena upload, kernel launches, timing.  Replaces renderGPU's body (gpuRender.cu:172-311).
#include "../../include/ptx_backend.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "device_scene.hpp"
#include "kernels.hpp"
#ifdef PTX_EXPERIMENT_PHASES  // DESIGN.md section 3, "phase kernel": measured slower, kept for A/B builds only
#include "../../experiments/radiance_kernel.hpp"
#endif
#ifdef PTX_EXPERIMENT_EXCHANGE  // DESIGN.md section 3, "block exchange": measured slower, kept for A/B builds only
#include "../../experiments/exchange_kernel.hpp"
#endif
#include "pool_kernel.hpp"
#ifdef PTX_EXPERIMENT_QUEUES  // global shared ray queues: measured 6-50x slower (profiles/r03/global_queues.txt), A/B builds only
#include "../../experiments/queue_kernel.hpp"
#endif
#include "tight_bounds.hpp"
#include "wide_bvh.hpp"

using namespace ptx;

// Scratch of ptx_accumulate_device_noalloc (include/ptx_backend.h): one array per GPU, held until the
// process ends.  In host code this name is only a handle for hipGetSymbolAddress; the device address is
// resolved once per device in ptx_create (scratchOnDevice) and kernels receive it as an argument.
static __device__ __align__(256) unsigned char gNoAllocScratch[PTX_NOALLOC_SCRATCH_BYTES];

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg) {
  g_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      char buf_[512];                                                                               \
      snprintf(buf_, sizeof(buf_), "HIP error at %s:%d: %s (%s)", __FILE__, __LINE__,                \
               hipGetErrorString(e_), #expr);                                                       \
      return fail(PTX_EHIP, buf_);                                                                  \
    }                                                                                               \
  } while (0)

// Device address of gNoAllocScratch on `device` (which must be the current device): every GPU holds
// its own copy of the array at its own address.
constexpr int kMaxDevices = 64;
int scratchOnDevice(int device, unsigned char** out) {
  static std::mutex mu;
  static unsigned char* ptrs[kMaxDevices] = {};
  if (device < 0 || device >= kMaxDevices) return fail(PTX_EINVAL, "device index out of range");
  std::lock_guard<std::mutex> lock(mu);
  if (!ptrs[device]) {
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(gNoAllocScratch)));
    if (!p || ((uintptr_t)p & 255u)) return fail(PTX_EHIP, "hipGetSymbolAddress returned an unusable address for the static scratch array");
    ptrs[device] = static_cast<unsigned char*>(p);
  }
  *out = ptrs[device];
  return PTX_OK;
}

struct TreeInfo {
  uint32_t rootRef = 0;
  uint32_t depth = 0;
};

// Converts one reference-layout tree (32-byte nodes, children adjacent at left_first, left_first+1)
// into NodePair records appended to `pairs`.  primBase is added to leaf primitive indices.
int convertTree(const ptx_bvh_node* nodes, uint32_t n, uint32_t primBase, uint32_t primCount, bool isTlas,
                std::vector<NodePair>& pairs, TreeInfo& info) {
  const char* what = isTlas ? "TLAS" : "BLAS";
  if (n == 0 || (n & 1u) == 0u) return fail(PTX_EINVAL, std::string(what) + ": node count must be odd and > 0");
  const uint32_t pairBase = (uint32_t)pairs.size();
  const uint32_t nPairs = (n - 1u) / 2u;
  if ((uint64_t)pairBase + nPairs >= kLeafFlag) return fail(PTX_ELIMIT, "too many BVH nodes");
  pairs.resize((size_t)pairBase + nPairs);

  auto makeRef = [&](const ptx_bvh_node& nd, uint32_t& ref) -> int {
    if (nd.count > 0) {
      if (nd.count > (isTlas ? kMaxTlasLeafPrims : kMaxBlasLeafPrims))
        return fail(PTX_ELIMIT, std::string(what) + ": a leaf holds more primitives than a node reference can encode (128 triangles / 64 instances)");
      if ((uint64_t)nd.left_first + nd.count > primCount) return fail(PTX_EINVAL, std::string(what) + ": leaf range outside the primitive array");
      if ((uint64_t)primBase + nd.left_first + nd.count > (isTlas ? kMaxInstIndex : kMaxTriIndex))
        return fail(PTX_ELIMIT, "primitive index exceeds what a node reference can encode (2^23 triangles / 2^24 instances)");
      ref = isTlas ? makeTlasLeafRef(primBase + nd.left_first, nd.count) : makeLeafRef(primBase + nd.left_first, nd.count);
    } else {
      if ((nd.left_first & 1u) == 0u || (uint64_t)nd.left_first + 1u >= n)
        return fail(PTX_EINVAL, std::string(what) + ": child index not of the form 2k+1 / out of range");
      ref = pairBase + (nd.left_first - 1u) / 2u;
    }
    return PTX_OK;
  };

  int rc = makeRef(nodes[0], info.rootRef);
  if (rc) return rc;

  // depth-first walk: fills pairs, measures depth, rejects cycles (each node may be visited once)
  std::vector<uint8_t> seen(n, 0);
  std::vector<std::pair<uint32_t, uint32_t>> todo;
  todo.emplace_back(0u, 0u);
  seen[0] = 1;
  info.depth = 0;
  while (!todo.empty()) {
    auto [i, depth] = todo.back();
    todo.pop_back();
    if (depth > info.depth) info.depth = depth;
    const ptx_bvh_node& nd = nodes[i];
    if (nd.count > 0) continue;
    const uint32_t l = nd.left_first, r = l + 1u;
    if ((l & 1u) == 0u || r >= n || seen[l] || seen[r]) return fail(PTX_EINVAL, std::string(what) + ": malformed node links");
    seen[l] = seen[r] = 1;
    NodePair& p = pairs[(size_t)pairBase + (l - 1u) / 2u];
    uint32_t lref, rref;
    if ((rc = makeRef(nodes[l], lref))) return rc;
    if ((rc = makeRef(nodes[r], rref))) return rc;
    p.lmin = make_float4(nodes[l].aabb_min[0], nodes[l].aabb_min[1], nodes[l].aabb_min[2], __builtin_bit_cast(float, lref));
    p.lmax = make_float4(nodes[l].aabb_max[0], nodes[l].aabb_max[1], nodes[l].aabb_max[2], 0.0f);
    p.rmin = make_float4(nodes[r].aabb_min[0], nodes[r].aabb_min[1], nodes[r].aabb_min[2], __builtin_bit_cast(float, rref));
    p.rmax = make_float4(nodes[r].aabb_max[0], nodes[r].aabb_max[1], nodes[r].aabb_max[2], 0.0f);
    todo.emplace_back(r, depth + 1u);
    todo.emplace_back(l, depth + 1u);
  }
  return PTX_OK;
}

// Slots the binary-layout walk can touch: slot 0 holds the bottom marker, every internal level of the
// TLAS and of the BLAS defers at most one child -- so sp reaches tlasDepth + blasDepth + 1 -- and every
// stepping iteration writes slot sp (the write that is not followed by sp++ leaves garbage above the top).
// The bound keeps the slot of the instance-exit marker that round 1 pushed (no longer pushed: the instance
// is left by the pop that reaches a TLAS-level entry), as one entry of slack: depth sum + 3 entries.
inline uint32_t binaryStackBound(uint32_t tlasDepth, uint32_t blasDepth) { return tlasDepth + blasDepth + 3u; }

template <typename T>
size_t arenaReserve(size_t& cursor, size_t count) {
  cursor = (cursor + 255u) & ~size_t(255);
  size_t off = cursor;
  cursor += count * sizeof(T);
  return off;
}

double nowSeconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Raw world-space bounds of every instance's transformed vertices, computed on `device` (instanceBoundsKernel):
// 6 floats per instance, and which instances have usable bounds.  Temporary device buffers only.
// The kernel's cost is instances x triangles.  Up to kExactPairBudget triangle-instance pairs (scene 1: 5.1e9, 7.7 ms)
// the bounds are those of the vertices themselves; beyond, every geometry is stood in for by the corners of its BLAS
// boxes eight levels down (at most 256 boxes: 2,048 points, still conservative, about 1.5x the exact volume), so that
// a scene with a million instances cannot turn ptx_create into minutes.
constexpr uint64_t kExactPairBudget = 40000000000ull;
uint64_t exactPairBudget() {  // PTX_TIGHT_EXACT_BUDGET=<pairs> overrides (tests force the stand-in path with 0)
  static const char* env = getenv("PTX_TIGHT_EXACT_BUDGET");
  return env ? strtoull(env, nullptr, 10) : kExactPairBudget;
}

void proxyPoints(const ptx_scene_desc* d, const ptx_geometry& G, std::vector<float>& pts) {
  const ptx_bvh_node* nodes = d->blas_nodes + G.node_offset;
  std::vector<std::pair<uint32_t, uint32_t>> todo(1, {0u, 0u});
  while (!todo.empty()) {
    auto [i, depth] = todo.back();
    todo.pop_back();
    const ptx_bvh_node& nd = nodes[i];
    if (nd.count == 0 && depth < 8u) {
      todo.push_back({nd.left_first, depth + 1u});
      todo.push_back({nd.left_first + 1u, depth + 1u});
      continue;
    }
    for (int c = 0; c < 9; c++)  // 8 corners + the first again: three "triangles" of three points
      for (int a = 0; a < 3; a++) pts.push_back((((c & 7) >> a) & 1) ? nd.aabb_max[a] : nd.aabb_min[a]);
  }
}

int deviceInstanceBounds(const ptx_scene_desc* d, int device, std::vector<float>& raw, std::vector<uint8_t>& usable,
                         bool* exactOut = nullptr) {
  const uint32_t n = d->instance_count;
  raw.assign((size_t)n * 6u, 0.0f);
  usable.assign(n, 0);
  std::vector<InstanceModel> models(n);
  uint64_t pairs = 0;
  for (uint32_t i = 0; i < n; i++)
    if (d->instances[i].geometry_id < d->geometry_count) pairs += d->geometries[d->instances[i].geometry_id].tri_count;
  const bool exact = pairs <= exactPairBudget();
  if (exactOut) *exactOut = exact;
  std::vector<float> proxy;                       // 9 floats per "triangle" when !exact
  std::vector<uint32_t> proxyOffset, proxyCount;  // per geometry, in "triangles"
  if (!exact) {
    proxyOffset.resize(d->geometry_count);
    proxyCount.resize(d->geometry_count);
    for (uint32_t g = 0; g < d->geometry_count; g++) {
      proxyOffset[g] = (uint32_t)(proxy.size() / 9u);
      if (d->geometries[g].tri_count != 0 && d->geometries[g].node_count != 0) proxyPoints(d, d->geometries[g], proxy);
      proxyCount[g] = (uint32_t)(proxy.size() / 9u) - proxyOffset[g];
    }
  }
  for (uint32_t i = 0; i < n; i++) {
    const ptx_instance& I = d->instances[i];
    if (I.geometry_id >= d->geometry_count) return fail(PTX_EINVAL, "instance refers to a missing geometry/material");
    const ptx_geometry& G = d->geometries[I.geometry_id];
    double M[12];
    InstanceModel& m = models[i];
    memset(&m, 0, sizeof(m));
    if (G.tri_count != 0 && tight_detail::invertAffine(I.inv_model, M)) {
      for (int k = 0; k < 12; k++) m.m[k] = (float)M[k];
      m.triOffset = exact ? G.tri_offset : proxyOffset[I.geometry_id];
      m.triCount = exact ? G.tri_count : proxyCount[I.geometry_id];
      usable[i] = m.triCount != 0u;
    }  // else triCount 0: the kernel writes an inverted box, which padBounds never sees
  }
  HIP_TRY(hipSetDevice(device));
  float *dVerts = nullptr, *dOut = nullptr;
  InstanceModel* dModels = nullptr;
  const size_t vertBytes = exact ? (size_t)d->triangle_total * sizeof(ptx_triangle) : proxy.size() * sizeof(float);
  const void* vertSrc = exact ? (const void*)d->triangles : (const void*)proxy.data();
  hipError_t e = hipMalloc((void**)&dVerts, vertBytes ? vertBytes : 4);
  if (e == hipSuccess) e = hipMalloc((void**)&dModels, (size_t)n * sizeof(InstanceModel));
  if (e == hipSuccess) e = hipMalloc((void**)&dOut, raw.size() * sizeof(float));
  if (e == hipSuccess && vertBytes) e = hipMemcpy(dVerts, vertSrc, vertBytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dModels, models.data(), (size_t)n * sizeof(InstanceModel), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    instanceBoundsKernel<<<dim3((n + kBoundsGroup - 1u) / kBoundsGroup), dim3(256)>>>(dVerts, dModels, n, dOut);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(raw.data(), dOut, raw.size() * sizeof(float), hipMemcpyDeviceToHost);
  if (dVerts) (void)hipFree(dVerts);
  if (dModels) (void)hipFree(dModels);
  if (dOut) (void)hipFree(dOut);
  HIP_TRY(e);
  return PTX_OK;
}

}  // namespace

struct ptx_scene {
  int device = 0;
  void* arena = nullptr;
  size_t arenaBytes = 0;
  DeviceScene dev{};
  uint32_t tlasDepth = 0, blasDepth = 0;
  uint32_t wideStackBound = 0;  // 0: the wide layout is not used for this scene
  double uploadSeconds = 0;
  // per-handle scratch, grown on demand
  void* outBuf = nullptr;
  size_t outBytes = 0;
  DeviceStats* dStats = nullptr;
  uint32_t* dWorkCounter = nullptr;
  uint32_t* dOverflow = nullptr;  // traversal-stack overflow area (grown on demand)
  size_t ovfBytes = 0;
  int numCus = 0;
  hipEvent_t evStart = nullptr, evStop = nullptr;
  hipStream_t lastStream = nullptr;
  bool timed = false;
  uint8_t* dRows = nullptr;   // device-side output stage: the BMP pixel array
  size_t rowsBytes = 0;
  float4* dPlanes = nullptr;  // per-stream accumulation planes of a sub-stream launch
  size_t planeBytes = 0;
  unsigned char* staticScratch = nullptr;  // this device's copy of gNoAllocScratch (resolved in ptx_create)
  unsigned char* dQueue = nullptr;         // queues and chain records of queueKernel (queue_kernel.hpp), grown on demand
  size_t queueBytes = 0;
  bool queueLaunched = false;              // the last radiance launch was a queueKernel: its error word is read at the next synchronisation
  // ptx_render_multi / ptx_render_counts_multi: this handle's own stream and "my part is on the root" event,
  // and (root handle only) the buffer the parts are gathered into
  hipStream_t ownStream = nullptr;
  hipEvent_t evPart = nullptr;
  void* dGather = nullptr;
  size_t gatherBytes = 0;
};

namespace {

int ensureOut(ptx_scene* s, size_t bytes) {
  if (s->outBytes >= bytes) return PTX_OK;
  if (s->outBuf) HIP_TRY(hipFree(s->outBuf));
  s->outBuf = nullptr;
  s->outBytes = 0;
  HIP_TRY(hipMalloc(&s->outBuf, bytes));
  s->outBytes = bytes;
  return PTX_OK;
}

int makeFrame(const ptx_scene* s, const ptx_camera* cam, const ptx_params* prm, bool heatmap, FrameConstants& fc) {
  if (!s || !cam || !prm) return fail(PTX_EINVAL, "null argument");
  if (prm->width == 0 || prm->height == 0) return fail(PTX_EINVAL, "empty image");
  if ((uint64_t)prm->width * prm->height > (1ull << 24))
    return fail(PTX_ELIMIT, "more than 2^24 pixels: pixel coordinates would no longer be exact in float (main.cpp:199)");
  if (!heatmap && prm->spp == 0) return fail(PTX_EINVAL, "spp must be > 0 (a share of zero samples needs frame_spp > 0 and a device-buffer entry point)");
  if (!heatmap && prm->frame_spp != 0 && prm->frame_spp < prm->spp) return fail(PTX_EINVAL, "frame_spp < spp");
  if (prm->part_count > 1u && prm->part_index >= prm->part_count) return fail(PTX_EINVAL, "part_index >= part_count");
  using pt::vec3;
  memset(&fc, 0, sizeof(fc));
  const vec3 pos(cam->position[0], cam->position[1], cam->position[2]);
  const vec3 at(cam->target[0], cam->target[1], cam->target[2]);
  const vec3 up(cam->up[0], cam->up[1], cam->up[2]);
  // camera.hpp:14-20
  const vec3 zaxis = pt::normalize(at - pos);
  const vec3 xaxis = pt::normalize(pt::cross(zaxis, up));
  const vec3 yaxis = pt::cross(xaxis, zaxis);
  const vec3 nz = -zaxis;
  fc.camPos[0] = pos.x; fc.camPos[1] = pos.y; fc.camPos[2] = pos.z;
  fc.viewC0[0] = xaxis.x; fc.viewC0[1] = xaxis.y; fc.viewC0[2] = xaxis.z;
  fc.viewC1[0] = yaxis.x; fc.viewC1[1] = yaxis.y; fc.viewC1[2] = yaxis.z;
  fc.viewC2[0] = nz.x; fc.viewC2[1] = nz.y; fc.viewC2[2] = nz.z;
  // camera.hpp:8-12
  const float resx = (float)prm->width, resy = (float)prm->height;
  fc.halfResX = 0.5f * resx;
  fc.halfResY = 0.5f * resy;
  const float z = (0.5f * resy) / tanf(0.5f * pt::radians(cam->fov_deg));
  fc.negZ = -z;
  // gpuFunctions.hpp:51-52 with angle = (float)(-M_PI) (gpuRender.cu:16)
  const float angle = (float)(-M_PI);
  fc.envQy = (float)sin(angle / 2.0);
  fc.envQw = (float)cos(angle / 2.0);
  fc.width = prm->width;
  fc.height = prm->height;
  fc.spp = heatmap ? 1u : prm->spp;
  fc.maxBounces = heatmap ? 1u : prm->max_bounces;
  fc.jitter = (!heatmap && (prm->frame_spp ? prm->frame_spp : prm->spp) > 1u) ? 1u : 0u;
  uint32_t streams = (heatmap || prm->sub_streams == 0u) ? 1u : prm->sub_streams;
  if (streams > 1u && prm->seed_mode != PTX_SEED_STREAM)
    return fail(PTX_EINVAL, "sub_streams > 1 needs PTX_SEED_STREAM: the reference's seeding has one RNG stream per pixel");
  if (streams > 256u) return fail(PTX_EINVAL, "sub_streams > 256");
  if (streams > fc.spp) streams = fc.spp;  // no empty streams
  fc.subStreams = streams;
  fc.sppBase = fc.spp / streams;
  fc.sppExtra = fc.spp % streams;
  fc.invPixels = 1.0 / ((double)prm->width * (double)prm->height);
  {
    uint32_t sh = 0;
    while ((1u << sh) < prm->width) sh++;  // width <= 2^24
    fc.widthShift = sh;
    fc.widthMul = (uint32_t)((((uint64_t)1 << (24 + sh)) + prm->width - 1u) / prm->width);  // <= 2^25
    const uint64_t band = 8ull * prm->width;  // <= 2^27
    sh = 0;
    while ((1ull << sh) < band) sh++;
    fc.bandShift = sh;
    fc.bandMul = (uint32_t)((((uint64_t)1 << (24 + sh)) + band - 1u) / band);
  }
  {
    // Long work items (16 samples or more of one RNG stream: the reference's seeding) are handed to the pool kernel's waves
    // scattered over the image, item w -> pixel slot (w * a) mod pixels with a coprime to the pixel count and near
    // pixels / golden ratio, so that every wave holds expensive and cheap pixels (pool_kernel.hpp); short items keep the
    // 8x8 blocks of adjacent pixels.  PTX_SCATTER=0 / 1 overrides (A/B measurements).
    const char* env = getenv("PTX_SCATTER");
    const bool scatter = env ? env[0] == '1' : fc.sppBase >= 16u;
    if (scatter && !heatmap) {
      const uint64_t n = (uint64_t)prm->width * prm->height;
      uint64_t a = (uint64_t)((double)n * 0.6180339887498949);
      auto gcd = [](uint64_t x, uint64_t y) { while (y) { const uint64_t t = x % y; x = y; y = t; } return x; };
      while (a > 1 && gcd(a, n) != 1) a--;
      fc.scatterMul = (uint32_t)(a > 1 ? a : 0);
    }
  }
  fc.seedBase = 1023u;
  // stream id of this call's k-th stream = part_index * sub_streams + k with the REQUESTED sub_streams: a part whose
  // share of the samples is smaller than that runs fewer streams, and its ids must not slide into its neighbour's
  if (prm->seed_mode == PTX_SEED_STREAM) fc.seedBase += prm->part_index * (prm->sub_streams > 1u && !heatmap ? prm->sub_streams : 1u) * 0x9E3779B9u;
  else if (prm->seed_mode != PTX_SEED_REFERENCE_GPU) return fail(PTX_EINVAL, "unknown seed_mode");
  fc.partIndex = prm->part_index;
  fc.partCount = prm->part_count;
  fc.tilesX = (prm->width + 7u) / 8u;
  fc.tilesY = (prm->height + 7u) / 8u;
  if ((uint64_t)prm->width * prm->height * streams >= (1ull << 32)) return fail(PTX_ELIMIT, "pixels x sub_streams exceeds 2^32 work items");
  return PTX_OK;
}

// Persistent launch: the grid is sized to the chip (CUs x resident blocks per CU), not to the image.
// The occupancy query is made once per kernel and device (the no-allocation entry point may not make
// runtime queries that could allocate).
template <typename K>
int persistentGrid(ptx_scene* s, K kernel, uint32_t& blocks) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> cache;
  int perCu = 0;
  {
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(reinterpret_cast<const void*>(kernel), s->device);
    auto it = cache.find(key);
    if (it == cache.end()) {
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kernel, (int)kBlockThreads, 0));
      if (perCu < 1) perCu = 1;
      cache[key] = perCu;
    } else {
      perCu = it->second;
    }
  }
  static const char* cap = getenv("PTX_MAX_BLOCKS_PER_CU");  // experiments only
  if (cap && atoi(cap) >= 1 && atoi(cap) < perCu) perCu = atoi(cap);
  blocks = (uint32_t)(s->numCus * perCu);
  return PTX_OK;
}

// Device memory a launch needs beyond the scene, the inputs and the outputs.  Either the handle's own
// buffers, grown on demand (ScratchOwner::handle), or slices of the library's static array (the
// no-allocation entry point).
struct Scratch {
  uint32_t* workCounter = nullptr;  // one dword, cleared before every launch
  uint32_t* overflow = nullptr;     // traversal-stack entries beyond the LDS part, [entry][thread]
  float4* planes = nullptr;         // per-stream accumulation planes of a sub-stream launch
};

// What a radiance / counts launch will need: decided before any memory is touched, so that the two
// scratch providers see the same numbers.
struct LaunchPlan {
  uint32_t blocks = 0;
  bool wide = false;
  size_t ovfBytes = 0;    // >= 4
  size_t planeBytes = 0;  // 0: the kernel writes the caller's buffer itself
};

template <typename K>
int planLaunch(ptx_scene* s, K kernel, const FrameConstants& fc, bool wide, uint32_t ldsStack, uint32_t raysPerLane, LaunchPlan& plan,
               uint32_t parkSlots = kParkSlots) {
  uint32_t blocks = 0;
  int rc = persistentGrid(s, kernel, blocks);
  if (rc) return rc;
  // one lane per work item: pixels x RNG streams (checked in makeFrame to fit 32 bits)
  const uint32_t work = fc.width * fc.height * fc.subStreams;
  const uint32_t perBlock = kBlockThreads * raysPerLane;
  const uint32_t needed = (work + perBlock - 1u) / perBlock;
  if (blocks > needed) blocks = needed;
  const uint32_t bound = wide ? s->wideStackBound : binaryStackBound(s->tlasDepth, s->blasDepth);
  const size_t deepEntries = bound > ldsStack ? bound - ldsStack : 0u;  // entries beyond the kernel's LDS part
  plan.blocks = blocks;
  plan.wide = wide;
  plan.ovfBytes = ((deepEntries ? deepEntries : 1u) + parkSlots) * (size_t)raysPerLane * blocks * kBlockThreads * sizeof(uint32_t);
  return PTX_OK;
}

template <typename K>
int launchPath(ptx_scene* s, K kernel, const FrameConstants& fc, const LaunchPlan& plan, const Scratch& scr, void* dOut,
               hipStream_t stream) {
  if (getenv("PTX_DEBUG_LAUNCH")) fprintf(stderr, "[ptx] pathKernel: %u blocks (%d CUs), %u streams\n", plan.blocks, s->numCus, fc.subStreams);
  HIP_TRY(hipMemsetAsync(scr.workCounter, 0, sizeof(uint32_t), stream));
  HIP_TRY(hipEventRecord(s->evStart, stream));
  kernel<<<dim3(plan.blocks), dim3(kBlockThreads), 0, stream>>>(s->dev, fc, scr.workCounter, dOut, s->dStats, scr.overflow);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(s->evStop, stream));
  s->lastStream = stream;
  s->timed = true;
  return PTX_OK;
}

// The handle's own scratch buffers, grown (never shrunk) to what the plan asks for.
int handleScratch(ptx_scene* s, const LaunchPlan& plan, Scratch& scr) {
  if (s->ovfBytes < plan.ovfBytes) {
    if (s->dOverflow) HIP_TRY(hipFree(s->dOverflow));
    s->dOverflow = nullptr;
    s->ovfBytes = 0;
    HIP_TRY(hipMalloc((void**)&s->dOverflow, plan.ovfBytes));
    s->ovfBytes = plan.ovfBytes;
  }
  if (s->planeBytes < plan.planeBytes) {
    if (s->dPlanes) HIP_TRY(hipFree(s->dPlanes));
    s->dPlanes = nullptr;
    s->planeBytes = 0;
    HIP_TRY(hipMalloc((void**)&s->dPlanes, plan.planeBytes));
    s->planeBytes = plan.planeBytes;
  }
  scr.workCounter = s->dWorkCounter;
  scr.overflow = s->dOverflow;
  scr.planes = s->dPlanes;
  return PTX_OK;
}

// The library's static scratch array (ptx_accumulate_device_noalloc): header | overflow | planes, each
// 256-byte aligned.  Nothing is allocated; a launch that needs more than the array holds is refused.
int staticScratch(ptx_scene* s, const LaunchPlan& plan, Scratch& scr) {
  if (!s->staticScratch) return fail(PTX_EHIP, "the static scratch array was not resolved on this device");
  auto up256 = [](size_t v) { return (v + 255u) & ~size_t(255); };
  const size_t oOvf = 256u, oPlanes = oOvf + up256(plan.ovfBytes), need = oPlanes + up256(plan.planeBytes);
  if (need > (size_t)PTX_NOALLOC_SCRATCH_BYTES) {
    char buf[256];
    snprintf(buf, sizeof(buf),
             "ptx_accumulate_device_noalloc: this launch needs %zu bytes of scratch (stack overflow area %zu + stream planes %zu), "
             "the static array holds %zu: use fewer sub_streams / a smaller frame, or ptx_accumulate_device",
             need, plan.ovfBytes, plan.planeBytes, (size_t)PTX_NOALLOC_SCRATCH_BYTES);
    return fail(PTX_ELIMIT, buf);
  }
  unsigned char* base = s->staticScratch;
  scr.workCounter = reinterpret_cast<uint32_t*>(base);
  scr.overflow = reinterpret_cast<uint32_t*>(base + oOvf);
  scr.planes = reinterpret_cast<float4*>(base + oPlanes);
  return PTX_OK;
}

enum class ScratchOwner { handle, library };

int launchCounts(ptx_scene* s, const FrameConstants& fc, uint32_t* dOut, hipStream_t stream, bool stats) {
  auto go = [&](auto kernel) {
    LaunchPlan plan;
    Scratch scr;
    int rc = planLaunch(s, kernel, fc, false, kLdsStack, 1u, plan);
    if (!rc) rc = handleScratch(s, plan, scr);
    if (!rc) rc = launchPath(s, kernel, fc, plan, scr, dOut, stream);
    return rc;
  };
  return stats ? go(pathKernel<kModeCounts, true, false>) : go(pathKernel<kModeCounts, false, false>);
}

// PTX_WIDE=0 in the environment keeps the radiance path on the binary layout (A/B measurements).
bool useWide(const ptx_scene* s) {
  static const char* env = getenv("PTX_WIDE");
  return s->wideStackBound != 0u && !(env && env[0] == '0');
}

// Which kernel traces radiance on the wide layout: poolKernel (pool_kernel.hpp) unless PTX_QUEUE=0 (pathKernel: A/B, tests);
// PTX_QUEUE=2 the global-queue experiment where it is compiled in.  Read at every launch: tests switch inside one process.
int radianceKernelChoice() {
  const char* env = getenv("PTX_QUEUE");
  return env ? atoi(env) : 1;
}
// Chains per lane of the grid: 2 for short work items (sub-streams), 3 for long ones, where all of the frame's chains should
// be under way from the start (profiles/r03/pool_chains_ab.txt); PTX_POOL_CHAINS overrides.
uint32_t poolChainsPerLane(const FrameConstants& fc) {
  const char* env = getenv("PTX_POOL_CHAINS");
  const int v = env ? atoi(env) : (fc.sppBase >= 16u ? 3 : 2);
  return (uint32_t)(v < 1 ? 1 : (v > (int)kPMaxChainsPerLane ? (int)kPMaxChainsPerLane : v));
}
#ifdef PTX_EXPERIMENT_QUEUES
uint32_t queueChainsPerLane() {
  const char* env = getenv("PTX_QUEUE_CHAINS");
  const int v = env ? atoi(env) : 3;
  return (uint32_t)(v < 1 ? 1 : (v > 32 ? 32 : v));
}
uint32_t queueShards() {
  const char* env = getenv("PTX_QUEUE_SHARDS");
  const int v = env ? atoi(env) : 4;
  return (uint32_t)(v < 1 ? 1 : (v > (int)kQMaxShards ? (int)kQMaxShards : v));
}
#endif

int launchRadiance(ptx_scene* s, const FrameConstants& fc, float* dOut, bool resolve, hipStream_t stream, bool stats,
                   ScratchOwner owner = ScratchOwner::handle) {
  const bool wide = useWide(s);
  // with one stream and no resolve the kernel writes (sum, samples) itself; otherwise every stream of a
  // pixel is a work item of its own and writes its own plane, and the planes are added in stream order
  // afterwards (deterministic, unlike float atomics), and resolved if the caller wants pixels
  const bool direct = fc.subStreams == 1u && !resolve;
  const uint32_t nPixels = fc.width * fc.height;
  auto go = [&](auto kernel, uint32_t ldsStack, uint32_t raysPerLane = 1u) {
    LaunchPlan plan;
    Scratch scr;
    int rc = planLaunch(s, kernel, fc, wide, ldsStack, raysPerLane, plan);
    if (rc) return rc;
    plan.planeBytes = direct ? 0u : (size_t)fc.subStreams * nPixels * sizeof(float4);
    rc = owner == ScratchOwner::library ? staticScratch(s, plan, scr) : handleScratch(s, plan, scr);
    if (rc) return rc;
    if ((rc = launchPath(s, kernel, fc, plan, scr, direct ? (void*)dOut : (void*)scr.planes, stream))) return rc;
    if (!direct) {
      combineKernel<<<dim3((nPixels + 255u) / 256u), dim3(256), 0, stream>>>(scr.planes, fc.subStreams, nPixels, dOut, resolve);
      HIP_TRY(hipGetLastError());
    }
    return (int)PTX_OK;
  };
  // The pool kernel (pool_kernel.hpp): a private pool of chains per wave, shading in batches of 64.
  auto goPool = [&](auto kernel, auto pacedKernel) {
    LaunchPlan plan;
    Scratch scr;
    int rc = planLaunch(s, kernel, fc, true, kPStack, 1u, plan, kPParkSlots);
    if (rc) return rc;
    plan.planeBytes = direct ? 0u : (size_t)fc.subStreams * nPixels * sizeof(float4);
    if ((rc = handleScratch(s, plan, scr))) return rc;
    const uint32_t threads = plan.blocks * kBlockThreads;
    const uint32_t work = nPixels * fc.subStreams;
    PoolLayout q{};
    q.chainsPerLane = std::min(poolChainsPerLane(fc), (work + threads - 1u) / threads);
    const size_t bytes = (size_t)(threads / 64u) * 64u * q.chainsPerLane * kPSlotBytes + 256u;
    if (s->queueBytes < bytes) {
      if (s->dQueue) HIP_TRY(hipFree(s->dQueue));
      s->dQueue = nullptr;
      s->queueBytes = 0;
      HIP_TRY(hipMalloc((void**)&s->dQueue, bytes));
      s->queueBytes = bytes;
    }
    q.base = s->dQueue;
    if (getenv("PTX_DEBUG_LAUNCH"))
      fprintf(stderr, "[ptx] poolKernel: %u blocks (%d CUs), %u chains per lane, scatter %u, %u streams, %zu bytes of chain records\n", plan.blocks,
              s->numCus, q.chainsPerLane, fc.scatterMul, fc.subStreams, bytes);
    // static partition: long items, and (nearly) all of them are taken in the first rounds: pacedPoolKernel (pool_kernel.hpp, PACED);
    // PTX_POOL_PACED=0 launches poolKernel all the same (A/B; same frames)
    q.staticPartition = (fc.sppBase >= 16u && (uint64_t)work <= (uint64_t)threads * q.chainsPerLane + threads / 4u) ? 1u : 0u;
    if (const char* e = getenv("PTX_POOL_PACED")) if (e[0] == '0') q.staticPartition = 0u;
    HIP_TRY(hipMemsetAsync(scr.workCounter, 0, sizeof(uint32_t), stream));
    HIP_TRY(hipMemsetAsync(scr.workCounter + 32, 0, sizeof(uint32_t), stream));
    HIP_TRY(hipEventRecord(s->evStart, stream));
    if (q.staticPartition)
      pacedKernel<<<dim3(plan.blocks), dim3(kBlockThreads), 0, stream>>>(s->dev, fc, q, scr.workCounter, direct ? (void*)dOut : (void*)scr.planes,
                                                                          s->dStats, scr.overflow);
    else
      kernel<<<dim3(plan.blocks), dim3(kBlockThreads), 0, stream>>>(s->dev, fc, q, scr.workCounter, direct ? (void*)dOut : (void*)scr.planes, s->dStats,
                                                                     scr.overflow);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->evStop, stream));
    s->lastStream = stream;
    s->timed = true;
    if (!direct) {
      combineKernel<<<dim3((nPixels + 255u) / 256u), dim3(256), 0, stream>>>(scr.planes, fc.subStreams, nPixels, dOut, resolve);
      HIP_TRY(hipGetLastError());
    }
    return (int)PTX_OK;
  };
#ifdef PTX_EXPERIMENT_QUEUES
  // The queue kernel (queue_kernel.hpp): chains in memory, extend / shade roles around two ray queues.
  auto goQueue = [&](auto kernel) {
    LaunchPlan plan;
    Scratch scr;
    int rc = planLaunch(s, kernel, fc, true, kQStack, 1u, plan, kQParkSlots);
    if (rc) return rc;
    plan.planeBytes = direct ? 0u : (size_t)fc.subStreams * nPixels * sizeof(float4);
    if ((rc = handleScratch(s, plan, scr))) return rc;
    const uint32_t threads = plan.blocks * kBlockThreads;
    const uint32_t work = nPixels * fc.subStreams;
    QueueLayout q{};
    q.chainsPerLane = std::min(queueChainsPerLane(), (work + threads - 1u) / threads);
    const uint64_t chains = (uint64_t)q.chainsPerLane * threads;
    if (chains >= (1ull << 24)) return fail(PTX_ELIMIT, "queue kernel: more than 2^24 chains in flight");
    q.shards = queueShards();
    while ((1ull << q.ringLog2) < chains) q.ringLog2++;
    const uint64_t ringWord = (uint64_t)q.shards << q.ringLog2 << 4;  // bytes of one word array over all shards
    uint64_t cur = (queueCtlBytes(q.shards) + 64u /* debug counters of profiling builds */ + 255u) & ~255ull;
    uint32_t* ringOffsets[6] = {&q.ray0, &q.ray1, &q.shade0, &q.shade1, &q.shade2, &q.shade3};
    for (uint32_t* o : ringOffsets) { *o = (uint32_t)cur; cur += ringWord; }
    const uint64_t zeroBytes = cur;  // control words and rings: a tag is never 0
    uint32_t* chainOffsets[3] = {&q.p0, &q.p1, &q.p2};
    for (uint32_t* o : chainOffsets) { *o = (uint32_t)cur; cur += chains * 16u; }
    if (cur >= (1ull << 32) - 4096u) return fail(PTX_ELIMIT, "queue kernel: scratch exceeds the 4 GiB a buffer descriptor addresses (fewer shards or chains)");
    if (s->queueBytes < cur) {
      if (s->dQueue) HIP_TRY(hipFree(s->dQueue));
      s->dQueue = nullptr;
      s->queueBytes = 0;
      HIP_TRY(hipMalloc((void**)&s->dQueue, cur));
      s->queueBytes = cur;
    }
    q.base = s->dQueue;
    q.totalBytes = (uint32_t)cur;
    HIP_TRY(hipMemsetAsync(s->dQueue, 0, zeroBytes, stream));
    HIP_TRY(hipEventRecord(s->evStart, stream));
    kernel<<<dim3(plan.blocks), dim3(kBlockThreads), 0, stream>>>(s->dev, fc, q, direct ? (void*)dOut : (void*)scr.planes, s->dStats, scr.overflow);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->evStop, stream));
    s->lastStream = stream;
    s->timed = true;
    s->queueLaunched = true;
    if (!direct) {
      combineKernel<<<dim3((nPixels + 255u) / 256u), dim3(256), 0, stream>>>(scr.planes, fc.subStreams, nPixels, dOut, resolve);
      HIP_TRY(hipGetLastError());
    }
    return (int)PTX_OK;
  };
#endif
#ifdef PTX_EXPERIMENT_PHASES
  // experimental builds: the phase kernel (radiance_kernel.hpp) unless PTX_RAD_LEGACY=1
  static const char* legacy = getenv("PTX_RAD_LEGACY");
  if (wide && !(legacy && legacy[0] == '1')) return stats ? go(radianceKernel<true>, kRadStack, kRadRays) : go(radianceKernel<false>, kRadStack, kRadRays);
#endif
#ifdef PTX_EXPERIMENT_EXCHANGE
  static const char* exchange = getenv("PTX_EXCHANGE");  // experimental builds: the block-exchange kernel (exchange_kernel.hpp)
  if (wide && exchange && exchange[0] == '1') return stats ? go(exchangeKernel<true>, kXStack) : go(exchangeKernel<false>, kXStack);
#endif
  s->queueLaunched = false;
#ifdef PTX_EXPERIMENT_QUEUES
  if (wide && owner == ScratchOwner::handle && radianceKernelChoice() == 2) return stats ? goQueue(queueKernel<true>) : goQueue(queueKernel<false>);
#endif
  if (wide && owner == ScratchOwner::handle && radianceKernelChoice() == 1) return stats ? goPool(poolKernel<true>, pacedPoolKernel<true>) : goPool(poolKernel<false>, pacedPoolKernel<false>);
  if (wide) return stats ? go(pathKernel<kModeRadiance, true, true>, kLdsStack) : go(pathKernel<kModeRadiance, false, true>, kLdsStack);
  return stats ? go(pathKernel<kModeRadiance, true, false>, kLdsStack) : go(pathKernel<kModeRadiance, false, false>, kLdsStack);
}

// After a queueKernel launch has finished: its error word (a bounded spin ran out: queue_kernel.hpp).
int queueVerdict(ptx_scene* s) {
  if (!s->queueLaunched) return PTX_OK;
#ifdef PTX_EXPERIMENT_QUEUES
  uint32_t ctl[4] = {};
  HIP_TRY(hipMemcpy(ctl, s->dQueue, sizeof(ctl), hipMemcpyDeviceToHost));
  s->queueLaunched = false;
  if (ctl[kCtlError] != 0u) {
    char buf[200];
    snprintf(buf, sizeof(buf), "queue kernel gave up (code %u: 1 ray entry never arrived, 2 shade entry never arrived, 3 no progress for 5 s); %u items pulled, %u finished",
             ctl[kCtlError], ctl[kCtlWork], ctl[kCtlDone]);
    return fail(PTX_EHIP, buf);
  }
#endif
  return PTX_OK;
}

int kernelSeconds(ptx_scene* s, double* out) {
  if (!s->timed) return fail(PTX_EINVAL, "no launch has been made through this handle");
  HIP_TRY(hipEventSynchronize(s->evStop));
  if (int rc = queueVerdict(s)) return rc;
  float ms = 0.0f;
  HIP_TRY(hipEventElapsedTime(&ms, s->evStart, s->evStop));
  *out = (double)ms * 1e-3;
  return PTX_OK;
}

// A part of a sample-slice split that received no samples (more parts than samples): its accumulation
// buffer is all zeros -- (sum, samples) = 0 -- so that the collective still has a contribution from every
// rank.  Returns 0 if handled, a negative error, or 1 if the call is an ordinary one.
int emptyShare(ptx_scene* s, const ptx_params* prm, float* dAccum, hipStream_t stream) {
  if (!s || !prm || prm->spp != 0u || prm->frame_spp == 0u) return 1;
  if (!dAccum || prm->width == 0 || prm->height == 0) return fail(PTX_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemsetAsync(dAccum, 0, (size_t)prm->width * prm->height * sizeof(float4), stream));
  return PTX_OK;
}

int readStats(ptx_scene* s, ptx_stats* st) {
  DeviceStats h;
  HIP_TRY(hipMemcpy(&h, s->dStats, sizeof(h), hipMemcpyDeviceToHost));
  st->rays = h.rays; st->node_steps = h.nodeSteps; st->tri_tests = h.triTests; st->inst_entries = h.instEntries;
  st->hits = h.hits; st->misses = h.misses; st->box_hits = h.boxHits;
  return PTX_OK;
}

}  // namespace

extern "C" {

const char* ptx_last_error(void) { return g_error.c_str(); }

const char* ptx_version(void) {
  static std::string v;
  if (v.empty()) {
    int rt = 0;
    (void)hipRuntimeGetVersion(&rt);
    v = "ptx_backend 0.1 gfx950 HIP " + std::to_string(rt);
  }
  return v.c_str();
}

#ifndef PTX_BUILD_ID
#define PTX_BUILD_ID "unstamped"
#endif
const char* ptx_build_id(void) { return PTX_BUILD_ID; }

uint64_t ptx_noalloc_scratch_bytes(void) { return (uint64_t)PTX_NOALLOC_SCRATCH_BYTES; }

int ptx_device_count(int* count) {
  if (!count) return fail(PTX_EINVAL, "null argument");
  *count = 0;
  HIP_TRY(hipGetDeviceCount(count));
  return PTX_OK;
}

int ptx_create(const ptx_scene_desc* d, int device, ptx_scene** out) {
  if (!d || !out) return fail(PTX_EINVAL, "null argument");
  *out = nullptr;
  if (!d->tlas_nodes || !d->instances || !d->geometries || !d->blas_nodes || !d->triangles || !d->face_normals ||
      !d->materials || !d->environment)
    return fail(PTX_EINVAL, "scene description has a null required array");
  if (d->instance_count == 0 || d->geometry_count == 0 || d->material_count == 0 || d->env_width == 0 || d->env_height == 0)
    return fail(PTX_EINVAL, "scene description has an empty required array");
  if (d->texture_count != 0 && (!d->textures || !d->texels)) return fail(PTX_EINVAL, "texture_count > 0 with a null texture table or texel array");
  if ((uint64_t)d->env_width * d->env_height >= (1ull << 32)) return fail(PTX_ELIMIT, "environment map too large");
  const double t0 = nowSeconds();

  // ---- host re-layout ----
  std::vector<NodePair> pairs;
  TreeInfo tlas;
  int rc = convertTree(d->tlas_nodes, d->tlas_node_count, 0u, d->instance_count, true, pairs, tlas);
  if (rc) return rc;
  const uint32_t tlasPairEnd = (uint32_t)pairs.size();

  std::vector<TreeInfo> blas(d->geometry_count);
  std::vector<GeomShade> geomShade(d->geometry_count);
  uint32_t blasDepth = 0;
  for (uint32_t g = 0; g < d->geometry_count; g++) {
    const ptx_geometry& G = d->geometries[g];
    if ((uint64_t)G.node_offset + G.node_count > d->blas_node_total || (uint64_t)G.tri_offset + G.tri_count > d->triangle_total)
      return fail(PTX_EINVAL, "geometry ranges outside the pooled arrays");
    if ((uint64_t)G.face_normal_offset + G.tri_count > d->face_normal_total) return fail(PTX_EINVAL, "face normal range outside the array");
    if (G.vtx_normal_offset != PTX_NONE && (!d->vtx_normals || (uint64_t)G.vtx_normal_offset + 3ull * G.tri_count > d->vtx_normal_total))
      return fail(PTX_EINVAL, "vertex normal range outside the array");
    if (G.vtx_uv_offset != PTX_NONE && (!d->vtx_uvs || (uint64_t)G.vtx_uv_offset + 3ull * G.tri_count > d->vtx_uv_total))
      return fail(PTX_EINVAL, "uv range outside the array");
    rc = convertTree(d->blas_nodes + G.node_offset, G.node_count, G.tri_offset, G.tri_count, false, pairs, blas[g]);
    if (rc) return rc;
    if (blas[g].depth > blasDepth) blasDepth = blas[g].depth;
    geomShade[g].triBase = G.tri_offset;
    geomShade[g].faceNormalBase = G.face_normal_offset;
    geomShade[g].vtxNormalBase = G.vtx_normal_offset;
    geomShade[g].vtxUvBase = G.vtx_uv_offset;
  }
  // the same trees, four wide and quantised, for the radiance path (wide_bvh.hpp)
  std::vector<WideNode> wideNodes;
  WideInfo tlasWide;
  std::vector<WideInfo> blasWide(d->geometry_count);
  uint32_t blasWideDepth = 0, tlasWideEnd = 0;
  {
    // the radiance path's TLAS: the reference's tree with tight boxes (tight_bounds.hpp); PTX_TIGHT_TLAS=0 keeps
    // the reference's boxes (A/B measurements)
    static const char* tightEnv = getenv("PTX_TIGHT_TLAS");
    std::vector<ptx_bvh_node> tlasTight;
    if (!(tightEnv && tightEnv[0] == '0')) {
      std::vector<float> raw;
      std::vector<uint8_t> usable;
      if ((rc = deviceInstanceBounds(d, device, raw, usable))) return rc;
      std::vector<Box3> instBounds;
      std::vector<float> allowance;
      roundingAllowance(d, allowance);
      padBounds(raw.data(), usable, allowance, instBounds);
      tlasTight = refitTlas(d, instBounds, nullptr);
    }
    const ptx_bvh_node* tlasForWide = tlasTight.empty() ? d->tlas_nodes : tlasTight.data();
    std::string err = buildWideTree(tlasForWide, d->tlas_node_count,
                                    [&](const ptx_bvh_node& nd) { return makeTlasLeafRef(nd.left_first, nd.count); }, wideNodes, tlasWide);
    tlasWideEnd = (uint32_t)wideNodes.size();
    for (uint32_t g = 0; g < d->geometry_count && err.empty(); g++) {
      const ptx_geometry& G = d->geometries[g];
      err = buildWideTree(d->blas_nodes + G.node_offset, G.node_count,
                          [&](const ptx_bvh_node& nd) { return makeLeafRef(G.tri_offset + nd.left_first, nd.count); }, wideNodes, blasWide[g]);
      if (blasWide[g].depth > blasWideDepth) blasWideDepth = blasWide[g].depth;
    }
    if (!err.empty()) return fail(PTX_EINVAL, "wide BVH: " + err);
  }
  // a wide step defers up to three children; + the two markers + the spare slots above the top
  uint32_t wideStackBound = 3u * (tlasWide.depth + blasWideDepth) + 2u + 4u;
  if (wideStackBound > kStackEntries) wideStackBound = 0u;  // radiance stays on the binary layout

  if (binaryStackBound(tlas.depth, blasDepth) > kStackEntries) {  // sanity bound on the overflow area
    char buf[160];
    snprintf(buf, sizeof(buf), "BVH too deep for the traversal stack: TLAS depth %u + BLAS depth %u + 3 > %u", tlas.depth,
             blasDepth, kStackEntries);
    return fail(PTX_ELIMIT, buf);
  }

  for (uint32_t i = 0; i < d->instance_count; i++)
    if (d->instances[i].geometry_id >= d->geometry_count || d->instances[i].material_id >= d->material_count)
      return fail(PTX_EINVAL, "instance refers to a missing geometry/material");

  std::vector<TriRec> tris(d->triangle_total);
  for (uint32_t i = 0; i < d->triangle_total; i++) {
    const ptx_triangle& t = d->triangles[i];
    // edge1 = v1 - v0, edge2 = v2 - v0: the subtractions of intersection.hpp:11-12, done once
    tris[i].v0 = make_float4(t.v0[0], t.v0[1], t.v0[2], 0.0f);
    tris[i].e1 = make_float4(t.v1[0] - t.v0[0], t.v1[1] - t.v0[1], t.v1[2] - t.v0[2], 0.0f);
    tris[i].e2 = make_float4(t.v2[0] - t.v0[0], t.v2[1] - t.v0[1], t.v2[2] - t.v0[2], 0.0f);
  }

  std::vector<InstXform> ix(d->instance_count);
  std::vector<InstShade> is(d->instance_count);
  for (uint32_t i = 0; i < d->instance_count; i++) {
    const ptx_instance& I = d->instances[i];
    if (I.geometry_id >= d->geometry_count || I.material_id >= d->material_count) return fail(PTX_EINVAL, "instance refers to a missing geometry/material");
    const float* m = I.inv_model;
    ix[i].row0 = make_float4(m[0], m[4], m[8], m[12]);
    ix[i].row1 = make_float4(m[1], m[5], m[9], m[13]);
    ix[i].row2 = make_float4(m[2], m[6], m[10], m[14]);
    ix[i].rootRef = blas[I.geometry_id].rootRef;
    ix[i].wideRootRef = blasWide[I.geometry_id].rootRef;
    ix[i].pad[0] = ix[i].pad[1] = 0;
    const float* nm = I.normal_matrix;
    is[i].row0 = make_float4(nm[0], nm[4], nm[8], nm[12]);
    is[i].row1 = make_float4(nm[1], nm[5], nm[9], nm[13]);
    is[i].row2 = make_float4(nm[2], nm[6], nm[10], nm[14]);
    is[i].geometryId = I.geometry_id;
    is[i].materialId = I.material_id;
    is[i].pad[0] = is[i].pad[1] = 0;
  }

  std::vector<MaterialRec> mats(d->material_count);
  for (uint32_t i = 0; i < d->material_count; i++) {
    const ptx_material& M = d->materials[i];
    if ((M.albedo_texture != PTX_NONE && M.albedo_texture >= d->texture_count) ||
        (M.emissive_texture != PTX_NONE && M.emissive_texture >= d->texture_count))
      return fail(PTX_EINVAL, "material refers to a missing texture");
    mats[i].albedoMetal = make_float4(M.albedo[0], M.albedo[1], M.albedo[2], M.metalness);
    mats[i].emissiveRough = make_float4(M.emissive[0], M.emissive[1], M.emissive[2], M.roughness);
    mats[i].f0 = make_float4(M.f0[0], M.f0[1], M.f0[2], 0.0f);
    mats[i].albedoTex = M.albedo_texture;
    mats[i].emissiveTex = M.emissive_texture;
    mats[i].pad[0] = mats[i].pad[1] = 0;
  }
  std::vector<TextureRec> texs(d->texture_count);
  for (uint32_t i = 0; i < d->texture_count; i++) {
    const ptx_texture& T = d->textures[i];
    if (T.texel_offset + (uint64_t)T.width * T.height > d->texel_total || d->texel_total >= (1ull << 32))
      return fail(PTX_EINVAL, "texture range outside the texel array");
    texs[i].width = T.width; texs[i].height = T.height; texs[i].texelBase = (uint32_t)T.texel_offset; texs[i].pad = 0;
  }
  auto pad4 = [](const float* src, size_t n) {
    std::vector<float4> v(n);
    for (size_t i = 0; i < n; i++) v[i] = make_float4(src[3 * i], src[3 * i + 1], src[3 * i + 2], 0.0f);
    return v;
  };
  std::vector<float4> faceN = pad4(d->face_normals, d->face_normal_total);
  std::vector<float4> vtxN = d->vtx_normals ? pad4(d->vtx_normals, d->vtx_normal_total) : std::vector<float4>();
  std::vector<float4> texels = d->texels ? pad4(d->texels, d->texel_total) : std::vector<float4>();
  std::vector<float4> env = pad4(d->environment, (size_t)d->env_width * d->env_height);

  // ---- one arena, one allocation ----
  size_t cur = 0;
  const size_t oPairs = arenaReserve<NodePair>(cur, pairs.size() ? pairs.size() : 1);
  const size_t oWide = arenaReserve<WideNode>(cur, wideNodes.size() ? wideNodes.size() : 1);
  const size_t oTris = arenaReserve<TriRec>(cur, tris.size());
  const size_t oIx = arenaReserve<InstXform>(cur, ix.size());
  const size_t oIs = arenaReserve<InstShade>(cur, is.size());
  const size_t oGs = arenaReserve<GeomShade>(cur, geomShade.size());
  const size_t oFn = arenaReserve<float4>(cur, faceN.size());
  const size_t oVn = arenaReserve<float4>(cur, vtxN.size() ? vtxN.size() : 1);
  const size_t oUv = arenaReserve<float2>(cur, d->vtx_uv_total ? d->vtx_uv_total : 1);
  const size_t oMat = arenaReserve<MaterialRec>(cur, mats.size());
  const size_t oTex = arenaReserve<TextureRec>(cur, texs.size() ? texs.size() : 1);
  const size_t oTexels = arenaReserve<float4>(cur, texels.size() ? texels.size() : 1);
  const size_t oEnv = arenaReserve<float4>(cur, env.size());
  cur = ((cur + 255u) & ~size_t(255)) + 256u;  // tail pad: the unified fetch reads 64 B from a 48-B TriRec
  if (oIx / 16 + (uint64_t)ix.size() * 4u >= (1ull << 32) || oTris / 16 + (uint64_t)tris.size() * 3u >= (1ull << 32))
    return fail(PTX_ELIMIT, "traversal arrays exceed the 64 GiB addressable by 16-byte offsets");

  HIP_TRY(hipSetDevice(device));
  ptx_scene* s = new ptx_scene();
  s->device = device;
  s->arenaBytes = cur;
  hipError_t e = hipMalloc(&s->arena, cur);
  if (e != hipSuccess) {
    delete s;
    return fail(PTX_EHIP, std::string("hipMalloc of the scene arena failed: ") + hipGetErrorString(e));
  }
  char* base = (char*)s->arena;
  auto up = [&](size_t off, const void* src, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpy(base + off, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
  };
  hipError_t errs[] = {
      up(oWide, wideNodes.data(), wideNodes.size() * sizeof(WideNode)),
      up(oPairs, pairs.data(), pairs.size() * sizeof(NodePair)), up(oTris, tris.data(), tris.size() * sizeof(TriRec)),
      up(oIx, ix.data(), ix.size() * sizeof(InstXform)),         up(oIs, is.data(), is.size() * sizeof(InstShade)),
      up(oGs, geomShade.data(), geomShade.size() * sizeof(GeomShade)), up(oFn, faceN.data(), faceN.size() * sizeof(float4)),
      up(oVn, vtxN.data(), vtxN.size() * sizeof(float4)),        up(oUv, d->vtx_uvs, (size_t)d->vtx_uv_total * sizeof(float2)),
      up(oMat, mats.data(), mats.size() * sizeof(MaterialRec)),  up(oTex, texs.data(), texs.size() * sizeof(TextureRec)),
      up(oTexels, texels.data(), texels.size() * sizeof(float4)), up(oEnv, env.data(), env.size() * sizeof(float4)),
      hipMalloc((void**)&s->dStats, sizeof(DeviceStats)),        hipEventCreate(&s->evStart), hipEventCreate(&s->evStop),
      hipMalloc((void**)&s->dWorkCounter, 256),                  hipDeviceGetAttribute(&s->numCus, hipDeviceAttributeMultiprocessorCount, device)};
  for (hipError_t er : errs)
    if (er != hipSuccess) {
      std::string msg = std::string("scene upload failed: ") + hipGetErrorString(er);
      ptx_destroy(s);
      return fail(PTX_EHIP, msg);
    }
  DeviceScene& D = s->dev;
  D.arena = (const float4*)base;
  D.pairs16 = (uint32_t)(oPairs / 16);
  D.wide16 = (uint32_t)(oWide / 16);
  D.tris16 = (uint32_t)(oTris / 16);
  D.inst16 = (uint32_t)(oIx / 16);
  D.pairs = (const NodePair*)(base + oPairs);
  D.tris = (const TriRec*)(base + oTris);
  D.instXform = (const InstXform*)(base + oIx);
  D.instShade = (const InstShade*)(base + oIs);
  D.geomShade = (const GeomShade*)(base + oGs);
  D.faceNormals = (const float4*)(base + oFn);
  D.vtxNormals = (const float4*)(base + oVn);
  D.vtxUvs = (const float2*)(base + oUv);
  D.materials = (const MaterialRec*)(base + oMat);
  D.textures = (const TextureRec*)(base + oTex);
  D.texels = (const float4*)(base + oTexels);
  D.environment = (const float4*)(base + oEnv);
  D.tlasRootRef = tlas.rootRef;
  D.tlasRootRefWide = tlasWide.rootRef;
  D.tlasNodeEnd[0] = tlasPairEnd;
  D.tlasNodeEnd[1] = tlasWideEnd;
  s->wideStackBound = wideStackBound;
  D.envWidth = d->env_width;
  D.envHeight = d->env_height;
  D.envHeightF = (float)d->env_height;
  s->tlasDepth = tlas.depth;
  s->blasDepth = blasDepth;
  // set-up for the no-allocation entry point: this device's address of the static scratch array, and
  // the occupancy of the kernels it launches (so that the call itself makes no runtime query)
  rc = scratchOnDevice(device, &s->staticScratch);
  uint32_t blocks = 0;
#ifdef PTX_EXPERIMENT_PHASES
  if (!rc) rc = persistentGrid(s, radianceKernel<false>, blocks);
#endif
  if (!rc) rc = persistentGrid(s, poolKernel<false>, blocks);
  if (!rc) rc = persistentGrid(s, pacedPoolKernel<false>, blocks);
  if (!rc) rc = persistentGrid(s, pathKernel<kModeRadiance, false, true>, blocks);
  if (!rc) rc = persistentGrid(s, pathKernel<kModeRadiance, false, false>, blocks);
  if (rc) {
    std::string msg = g_error;
    ptx_destroy(s);
    return fail(rc, msg);
  }
  s->uploadSeconds = nowSeconds() - t0;
  *out = s;
  return PTX_OK;
}

int ptx_selftest_reciprocal(int device, uint64_t* mismatches) {
  if (!mismatches) return fail(PTX_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(device));
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, sizeof(*d)));
  hipError_t e = hipMemset(d, 0, sizeof(*d));
  if (e == hipSuccess) {
    reciprocalCheckKernel<<<dim3(4096), dim3(256)>>>(d);
    e = hipGetLastError();
  }
  unsigned long long h = 0;
  if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIP_TRY(e);
  *mismatches = h;
  return PTX_OK;
}

int ptx_check_tight_tlas(const ptx_scene_desc* d, int device, uint64_t report[6]) {
  if (!d || !report || !d->tlas_nodes || !d->instances || !d->geometries || !d->triangles) return fail(PTX_EINVAL, "null argument");
  for (int i = 0; i < 6; i++) report[i] = 0;
  std::vector<NodePair> pairs;
  TreeInfo info;
  int rc = convertTree(d->tlas_nodes, d->tlas_node_count, 0u, d->instance_count, true, pairs, info);  // validates the links
  if (rc) return rc;
  for (uint32_t g = 0; g < d->geometry_count; g++)
    if ((uint64_t)d->geometries[g].tri_offset + d->geometries[g].tri_count > d->triangle_total) return fail(PTX_EINVAL, "geometry ranges outside the pooled arrays");
  for (uint32_t i = 0; i < d->instance_count; i++)
    if (d->instances[i].geometry_id >= d->geometry_count) return fail(PTX_EINVAL, "instance refers to a missing geometry/material");
  std::vector<float> raw, rawHost((size_t)d->instance_count * 6u, 0.0f);
  std::vector<uint8_t> usable, usableHost(d->instance_count, 0);
  bool exact = true;
  if (device < 0) {  // no GPU: the host loop for every instance (slow for large scenes; CPU tests use small ones)
    raw.assign((size_t)d->instance_count * 6u, 0.0f);
    usable.assign(d->instance_count, 0);
    rawInstanceBoundsHost(d, 0u, d->instance_count, raw.data(), usable);
  } else {
    // the stand-in path walks the BLAS boxes: validate those links too (ptx_create has done so by then)
    for (uint32_t g = 0; g < d->geometry_count && !d->blas_nodes; g++) return fail(PTX_EINVAL, "null argument");
    for (uint32_t g = 0; g < d->geometry_count; g++) {
      const ptx_geometry& G = d->geometries[g];
      if ((uint64_t)G.node_offset + G.node_count > d->blas_node_total) return fail(PTX_EINVAL, "geometry ranges outside the pooled arrays");
      std::vector<NodePair> scratchPairs;
      TreeInfo ti;
      if ((rc = convertTree(d->blas_nodes + G.node_offset, G.node_count, G.tri_offset, G.tri_count, false, scratchPairs, ti))) return rc;
    }
    if ((rc = deviceInstanceBounds(d, device, raw, usable, &exact))) return rc;
  }
  // the same bounds from the host loop, for every instance up to 64 and an even sample of 64 beyond
  const uint32_t n = d->instance_count, step = n > 64u ? n / 64u : 1u;
  for (uint32_t i = 0; i < n; i += step) {
    rawInstanceBoundsHost(d, i, i + 1u, rawHost.data(), usableHost);
    report[1]++;
    bool same = usable[i] == usableHost[i];
    if (same && usable[i])
      for (int k = 0; k < 6; k++) {
        const float a = raw[6 * (size_t)i + k], b = rawHost[6 * (size_t)i + k];
        // exact mode: the same bits; stand-in mode (BLAS box corners): the box must hold the vertices' bounds
        same = same && (exact ? a == b : (k < 3 ? a <= b : a >= b));
      }
    if (!same) report[2]++;
  }
  std::vector<Box3> inst;
  std::vector<float> allowance;
  roundingAllowance(d, allowance);
  padBounds(raw.data(), usable, allowance, inst);
  // the padded boxes hold the exact (double) bounds of what the device's object-space test stands for
  for (uint32_t i = 0; i < n; i += step) {
    double lo[3], hi[3];
    if (!usable[i] || !exactInstanceBounds(d, i, lo, hi)) continue;
    for (int k = 0; k < 3; k++)
      if ((double)inst[i].mn[k] > lo[k] || (double)inst[i].mx[k] < hi[k]) {
        report[4]++;
        break;
      }
  }
  double ratio = 1.0;
  const std::vector<ptx_bvh_node> tight = refitTlas(d, inst, &ratio);
  for (uint32_t i = 0; i < d->tlas_node_count; i++) {
    const ptx_bvh_node &a = tight[i], &b = d->tlas_nodes[i];
    bool inside = a.left_first == b.left_first && a.count == b.count;
    for (int k = 0; k < 3; k++) inside = inside && a.aabb_min[k] >= b.aabb_min[k] && a.aabb_max[k] <= b.aabb_max[k] && a.aabb_min[k] <= a.aabb_max[k];
    if (!inside) report[3]++;
    if (a.count > 0)  // a leaf's box holds the padded bounds of its instances, unless the reference's own box cut them
      for (uint32_t j = 0; j < a.count; j++)
        for (int k = 0; k < 3; k++)
          if (usable[a.left_first + j] && (a.aabb_min[k] > fmaxf(inst[a.left_first + j].mn[k], b.aabb_min[k]) ||
                                           a.aabb_max[k] < fminf(inst[a.left_first + j].mx[k], b.aabb_max[k])))
            report[4]++;
  }
  report[0] = n;
  report[5] = (uint64_t)(ratio * 1e6);
  return PTX_OK;
}

int ptx_check_wide_layout(const ptx_scene_desc* d, uint64_t report[6]) {
  if (!d || !report || !d->tlas_nodes || !d->geometries || !d->blas_nodes) return fail(PTX_EINVAL, "null argument");
  for (int i = 0; i < 6; i++) report[i] = 0;
  // one pass per tree: build, then walk the wide nodes from the root
  auto check = [&](const ptx_bvh_node* nodes, uint32_t n, bool isTlas, uint32_t primBase) -> int {
    std::vector<WideNode> wide;
    WideInfo info;
    // this entry point takes descriptions that never went through ptx_create: validate the links first
    // (children at 2k+1 / 2k+2, in range, every node reached once), as buildWideTree assumes
    if (n == 0 || (n & 1u) == 0u) return fail(PTX_EINVAL, "node count must be odd and > 0");
    {
      std::vector<uint8_t> seen(n, 0);
      std::vector<uint32_t> walk(1, 0u);
      seen[0] = 1;
      while (!walk.empty()) {
        const uint32_t i = walk.back();
        walk.pop_back();
        if (nodes[i].count > 0) continue;
        const uint32_t l = nodes[i].left_first;
        if ((l & 1u) == 0u || (uint64_t)l + 1u >= n || seen[l] || seen[l + 1u]) return fail(PTX_EINVAL, "malformed node links");
        seen[l] = seen[l + 1u] = 1;
        walk.push_back(l);
        walk.push_back(l + 1u);
      }
    }
    // leaves are referenced by their node index here, so that coverage can be checked against nodes[]
    std::string err = buildWideTree(nodes, n, [&](const ptx_bvh_node& nd) { return kLeafFlag | (uint32_t)(&nd - nodes); }, wide, info);
    (void)isTlas; (void)primBase;
    if (!err.empty()) return fail(PTX_EINVAL, "wide BVH: " + err);
    std::vector<uint32_t> leafSeen(n, 0);
    uint64_t leaves = 0;
    for (uint32_t i = 0; i < n; i++) leaves += nodes[i].count > 0;
    if (info.depth > report[4]) report[4] = info.depth;
    report[0] += wide.size();
    report[5] += (n - leaves) - wide.size();
    if ((int32_t)info.rootRef < 0) {  // single-leaf tree
      leafSeen[info.rootRef & ~kLeafFlag]++;
    }
    // every wide node knows which binary nodes its children are only through the boxes: recompute the
    // fold to pair child slots with binary nodes (same rule as the builder: its determinism is part of
    // what is checked), then compare boxes
    struct Item { uint32_t node, wideIndex; };
    std::vector<Item> todo;
    if ((int32_t)info.rootRef >= 0) todo.push_back({0u, info.rootRef});
    while (!todo.empty()) {
      const Item it = todo.back();
      todo.pop_back();
      const WideNode& w = wide[it.wideIndex];
      const uint32_t eb = __builtin_bit_cast(uint32_t, w.q0.w);
      const uint32_t nk = eb >> 24;
      const float origin[3] = {w.q0.x, w.q0.y, w.q0.z};
      const uint32_t lo[3] = {w.q1.x, w.q1.y, w.q1.z}, hi[3] = {w.q1.w, w.q2.x, w.q2.y};
      const uint32_t refs[4] = {w.q3.x, w.q3.y, w.q3.z, w.q3.w};
      // the builder's fold
      uint32_t kids[4] = {nodes[it.node].left_first, nodes[it.node].left_first + 1u, 0, 0};
      uint32_t cnt = 2;
      while (cnt < 4) {
        int best = -1;
        float bestArea = -1.0f;
        for (uint32_t c = 0; c < cnt; c++)
          if (nodes[kids[c]].count == 0 && wide_detail::halfArea(nodes[kids[c]]) > bestArea) {
            bestArea = wide_detail::halfArea(nodes[kids[c]]);
            best = (int)c;
          }
        if (best < 0) break;
        const uint32_t l = nodes[kids[best]].left_first;
        kids[best] = l;
        kids[cnt++] = l + 1u;
      }
      if (cnt != nk) { report[3]++; continue; }
      for (uint32_t c = 0; c < nk; c++) {
        report[1]++;
        bool covers = true;
        for (int a = 0; a < 3; a++) {
          const float scale = __builtin_bit_cast(float, ((eb >> (8 * a)) & 0xFFu) << 23);
          const float qlo = std::fmaf((float)((lo[a] >> (8 * c)) & 0xFFu), scale, origin[a]);
          const float qhi = std::fmaf((float)((hi[a] >> (8 * c)) & 0xFFu), scale, origin[a]);
          if (!(qlo <= nodes[kids[c]].aabb_min[a]) || !(qhi >= nodes[kids[c]].aabb_max[a])) covers = false;
        }
        if (!covers) report[2]++;
        if (nodes[kids[c]].count > 0) {
          if (refs[c] != (kLeafFlag | kids[c])) report[3]++;
          else leafSeen[kids[c]]++;
        } else if ((int32_t)refs[c] < 0 || refs[c] >= wide.size()) {
          report[3]++;
        } else {
          todo.push_back({kids[c], refs[c]});
        }
      }
    }
    for (uint32_t i = 0; i < n; i++)
      if (nodes[i].count > 0 && leafSeen[i] != 1u) report[3]++;
    return PTX_OK;
  };
  if (d->tlas_node_count == 0 || (d->tlas_node_count & 1u) == 0u) return fail(PTX_EINVAL, "TLAS: node count must be odd and > 0");
  int rc = check(d->tlas_nodes, d->tlas_node_count, true, 0u);
  if (rc) return rc;
  for (uint32_t g = 0; g < d->geometry_count; g++) {
    const ptx_geometry& G = d->geometries[g];
    if (G.node_count == 0 || (G.node_count & 1u) == 0u || (uint64_t)G.node_offset + G.node_count > d->blas_node_total)
      return fail(PTX_EINVAL, "BLAS: bad node range");
    if ((rc = check(d->blas_nodes + G.node_offset, G.node_count, false, G.tri_offset))) return rc;
  }
  return PTX_OK;
}

void ptx_destroy(ptx_scene* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->arena) (void)hipFree(s->arena);
  if (s->outBuf) (void)hipFree(s->outBuf);
  if (s->dStats) (void)hipFree(s->dStats);
  if (s->dWorkCounter) (void)hipFree(s->dWorkCounter);
  if (s->dOverflow) (void)hipFree(s->dOverflow);
  if (s->dRows) (void)hipFree(s->dRows);
  if (s->dPlanes) (void)hipFree(s->dPlanes);
  if (s->dQueue) (void)hipFree(s->dQueue);
  if (s->dGather) (void)hipFree(s->dGather);
  if (s->ownStream) (void)hipStreamDestroy(s->ownStream);
  if (s->evPart) (void)hipEventDestroy(s->evPart);
  if (s->evStart) (void)hipEventDestroy(s->evStart);
  if (s->evStop) (void)hipEventDestroy(s->evStop);
  delete s;
}

int ptx_counts_device(ptx_scene* s, const ptx_camera* cam, const ptx_params* prm, uint32_t* dCounts, void* stream) {
  FrameConstants fc;
  int rc = makeFrame(s, cam, prm, true, fc);
  if (rc) return rc;
  if (!dCounts) return fail(PTX_EINVAL, "null output");
  HIP_TRY(hipSetDevice(s->device));
  return launchCounts(s, fc, dCounts, (hipStream_t)stream, false);
}

int ptx_render_counts(ptx_scene* s, const ptx_camera* cam, const ptx_params* prm, uint32_t* out, ptx_stats* stats) {
  FrameConstants fc;
  int rc = makeFrame(s, cam, prm, true, fc);
  if (rc) return rc;
  if (!out) return fail(PTX_EINVAL, "null output");
  HIP_TRY(hipSetDevice(s->device));
  const size_t bytes = (size_t)prm->width * prm->height * sizeof(uint32_t);
  if ((rc = ensureOut(s, bytes))) return rc;
  if ((rc = launchCounts(s, fc, (uint32_t*)s->outBuf, nullptr, false))) return rc;
  HIP_TRY(hipMemcpy(out, s->outBuf, bytes, hipMemcpyDeviceToHost));
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->upload_s = s->uploadSeconds;
    if ((rc = kernelSeconds(s, &stats->render_s))) return rc;
  }
  return PTX_OK;
}

int ptx_accumulate_device(ptx_scene* s, const ptx_camera* cam, const ptx_params* prm, float* dAccum, void* stream) {
  FrameConstants fc;
  int rc = emptyShare(s, prm, dAccum, (hipStream_t)stream);
  if (rc <= 0) return rc;
  if ((rc = makeFrame(s, cam, prm, false, fc))) return rc;
  if (!dAccum) return fail(PTX_EINVAL, "null output");
  if ((uintptr_t)dAccum & 15u) return fail(PTX_EINVAL, "d_accum must be 16-byte aligned");
  HIP_TRY(hipSetDevice(s->device));
  return launchRadiance(s, fc, dAccum, false, (hipStream_t)stream, false);
}

int ptx_accumulate_device_noalloc(ptx_scene* s, const ptx_camera* cam, const ptx_params* prm, float* dAccum, void* stream) {
  FrameConstants fc;
  int rc = emptyShare(s, prm, dAccum, (hipStream_t)stream);
  if (rc <= 0) return rc;
  if ((rc = makeFrame(s, cam, prm, false, fc))) return rc;
  if (!dAccum) return fail(PTX_EINVAL, "null output");
  if ((uintptr_t)dAccum & 15u) return fail(PTX_EINVAL, "d_accum must be 16-byte aligned");
  HIP_TRY(hipSetDevice(s->device));
  return launchRadiance(s, fc, dAccum, false, (hipStream_t)stream, false, ScratchOwner::library);
}

int ptx_resolve_device(ptx_scene* s, const float* dAccum, uint32_t width, uint32_t height, float* dOut, void* stream) {
  if (!s || !dAccum || !dOut || width == 0 || height == 0) return fail(PTX_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(s->device));
  const uint32_t n = width * height;
  resolveKernel<<<dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream>>>((const float4*)dAccum, n, dOut);
  HIP_TRY(hipGetLastError());
  return PTX_OK;
}

int ptx_render(ptx_scene* s, const ptx_camera* cam, const ptx_params* prm, float* out, ptx_stats* stats) {
  FrameConstants fc;
  int rc = makeFrame(s, cam, prm, false, fc);
  if (rc) return rc;
  if (!out) return fail(PTX_EINVAL, "null output");
  HIP_TRY(hipSetDevice(s->device));
  const size_t bytes = (size_t)prm->width * prm->height * 3u * sizeof(float);
  if ((rc = ensureOut(s, bytes))) return rc;
  if ((rc = launchRadiance(s, fc, (float*)s->outBuf, true, nullptr, false))) return rc;
  HIP_TRY(hipMemcpy(out, s->outBuf, bytes, hipMemcpyDeviceToHost));
  if ((rc = queueVerdict(s))) return rc;
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->upload_s = s->uploadSeconds;
    if ((rc = kernelSeconds(s, &stats->render_s))) return rc;
  }
  return PTX_OK;
}

namespace {
int ensureRows(ptx_scene* s, const ptx_params* prm, uint32_t rowBytes, size_t& bytes) {
  if ((uint64_t)rowBytes < 3ull * prm->width) return fail(PTX_EINVAL, "row_bytes < 3 * width");
  bytes = (size_t)rowBytes * prm->height;
  if (s->rowsBytes < bytes) {
    if (s->dRows) HIP_TRY(hipFree(s->dRows));
    s->dRows = nullptr;
    s->rowsBytes = 0;
    HIP_TRY(hipMalloc((void**)&s->dRows, bytes));
    s->rowsBytes = bytes;
  }
  HIP_TRY(hipMemsetAsync(s->dRows, 0, bytes, nullptr));  // row padding
  return PTX_OK;
}
}  // namespace

int ptx_render_counts_bgr8(ptx_scene* s, const ptx_camera* cam, const ptx_params* prm, uint8_t* out, uint32_t rowBytes,
                           uint32_t* maxCount, ptx_stats* stats) {
  FrameConstants fc;
  int rc = makeFrame(s, cam, prm, true, fc);
  if (rc) return rc;
  if (!out || !maxCount) return fail(PTX_EINVAL, "null output");
  HIP_TRY(hipSetDevice(s->device));
  const uint32_t n = prm->width * prm->height;
  size_t bytes = 0;
  if ((rc = ensureOut(s, (size_t)n * sizeof(uint32_t)))) return rc;
  if ((rc = ensureRows(s, prm, rowBytes, bytes))) return rc;
  if ((rc = launchCounts(s, fc, (uint32_t*)s->outBuf, nullptr, false))) return rc;
  uint32_t* dMax = s->dWorkCounter + 8;  // the work counter's allocation has room; launches clear only word 0
  HIP_TRY(hipMemsetAsync(dMax, 0, sizeof(uint32_t), nullptr));
  const uint32_t blocks = (n + 255u) / 256u;
  maxCountKernel<<<dim3(blocks < 1024u ? blocks : 1024u), dim3(256)>>>((const uint32_t*)s->outBuf, n, dMax);
  heatmapBytesKernel<<<dim3(blocks), dim3(256)>>>((const uint32_t*)s->outBuf, dMax, prm->width, n, rowBytes, s->dRows);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, s->dRows, bytes, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(maxCount, dMax, sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->upload_s = s->uploadSeconds;
    if ((rc = kernelSeconds(s, &stats->render_s))) return rc;
  }
  return PTX_OK;
}

int ptx_render_bgr8(ptx_scene* s, const ptx_camera* cam, const ptx_params* prm, uint8_t* out, uint32_t rowBytes, ptx_stats* stats) {
  FrameConstants fc;
  int rc = makeFrame(s, cam, prm, false, fc);
  if (rc) return rc;
  if (!out) return fail(PTX_EINVAL, "null output");
  HIP_TRY(hipSetDevice(s->device));
  const uint32_t n = prm->width * prm->height;
  size_t bytes = 0;
  if ((rc = ensureOut(s, (size_t)n * 3u * sizeof(float)))) return rc;
  if ((rc = ensureRows(s, prm, rowBytes, bytes))) return rc;
  if ((rc = launchRadiance(s, fc, (float*)s->outBuf, true, nullptr, false))) return rc;
  quantiseKernel<<<dim3((n + 255u) / 256u), dim3(256)>>>((const float*)s->outBuf, prm->width, n, rowBytes, s->dRows);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, s->dRows, bytes, hipMemcpyDeviceToHost));
  if ((rc = queueVerdict(s))) return rc;
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->upload_s = s->uploadSeconds;
    if ((rc = kernelSeconds(s, &stats->render_s))) return rc;
  }
  return PTX_OK;
}

namespace {

// Split of a frame's samples over `parts`: the first (spp % parts) parts take one more (the same rule as
// path-tracer_amd/py/multigpu.py: split_samples).
uint32_t shareOf(uint32_t spp, uint32_t parts, uint32_t part) { return spp / parts + (part < spp % parts ? 1u : 0u); }

int multiSetup(ptx_scene* const* scenes, uint32_t count, size_t gatherBytes) {
  if (!scenes || count == 0 || count > 64) return fail(PTX_EINVAL, "ptx_*_multi: need 1..64 scene handles");
  for (uint32_t i = 0; i < count; i++) {
    ptx_scene* s = scenes[i];
    if (!s) return fail(PTX_EINVAL, "ptx_*_multi: null scene handle");
    for (uint32_t j = 0; j < i; j++)
      if (scenes[j] == s) return fail(PTX_EINVAL, "ptx_*_multi: a scene handle is listed twice (one launch per handle may be in flight)");
    HIP_TRY(hipSetDevice(s->device));
    if (!s->ownStream) HIP_TRY(hipStreamCreateWithFlags(&s->ownStream, hipStreamNonBlocking));
    if (!s->evPart) HIP_TRY(hipEventCreateWithFlags(&s->evPart, hipEventDisableTiming));
  }
  ptx_scene* root = scenes[0];
  HIP_TRY(hipSetDevice(root->device));
  if (root->gatherBytes < gatherBytes) {
    if (root->dGather) HIP_TRY(hipFree(root->dGather));
    root->dGather = nullptr;
    root->gatherBytes = 0;
    HIP_TRY(hipMalloc(&root->dGather, gatherBytes));
    root->gatherBytes = gatherBytes;
  }
  return PTX_OK;
}

// part i's buffer (on its own device) -> plane i of the root's gather buffer; the root's stream then waits for it
int sendToRoot(ptx_scene* root, ptx_scene* part, uint32_t index, const void* src, size_t bytes) {
  HIP_TRY(hipSetDevice(part->device));
  HIP_TRY(hipMemcpyPeerAsync((char*)root->dGather + (size_t)index * bytes, root->device, src, part->device, bytes, part->ownStream));
  HIP_TRY(hipEventRecord(part->evPart, part->ownStream));
  HIP_TRY(hipSetDevice(root->device));
  HIP_TRY(hipStreamWaitEvent(root->ownStream, part->evPart, 0));
  return PTX_OK;
}

}  // namespace

int ptx_render_multi(ptx_scene* const* scenes, uint32_t count, const ptx_camera* cam, const ptx_params* frame, float* out,
                     ptx_stats* stats) {
  if (!cam || !frame || !out) return fail(PTX_EINVAL, "null argument");
  if (count > 1 && frame->seed_mode != PTX_SEED_STREAM)
    return fail(PTX_EINVAL, "ptx_render_multi: more than one GPU needs PTX_SEED_STREAM (one RNG stream set per GPU)");
  if (frame->spp == 0) return fail(PTX_EINVAL, "spp must be > 0");
  const size_t nPixels = (size_t)frame->width * frame->height;
  const size_t plane = nPixels * sizeof(float4);
  int rc = multiSetup(scenes, count, plane * count);
  if (rc) return rc;
  ptx_scene* root = scenes[0];
  const double t0 = nowSeconds();
  for (uint32_t i = 0; i < count; i++) {
    ptx_scene* s = scenes[i];
    ptx_params p = *frame;
    p.spp = shareOf(frame->spp, count, i);
    p.frame_spp = frame->spp;
    p.part_index = frame->part_index * count + i;  // a frame that is itself a part of a larger job keeps distinct streams
    p.part_count = (frame->part_count > 1u ? frame->part_count : 1u) * count;
    HIP_TRY(hipSetDevice(s->device));
    // part 0 accumulates straight into plane 0 of the gather buffer; the others into their own buffer, then copy
    float* dst = (float*)root->dGather;
    if (i != 0) {
      if ((rc = ensureOut(s, plane))) return rc;
      dst = (float*)s->outBuf;
    }
    rc = emptyShare(s, &p, dst, s->ownStream);
    if (rc < 0) return rc;
    if (rc > 0) {
      FrameConstants fc;
      if ((rc = makeFrame(s, cam, &p, false, fc))) return rc;
      if ((rc = launchRadiance(s, fc, dst, false, s->ownStream, false))) return rc;
    }
    if (i != 0 && (rc = sendToRoot(root, s, i, dst, plane))) return rc;
  }
  // root: add the parts in part order, divide by the summed sample count, grade, tonemap (gpuRender.cu:158-169)
  HIP_TRY(hipSetDevice(root->device));
  if ((rc = ensureOut(root, nPixels * 3u * sizeof(float)))) return rc;
  combineKernel<<<dim3((uint32_t)((nPixels + 255u) / 256u)), dim3(256), 0, root->ownStream>>>((const float4*)root->dGather, count,
                                                                                                (uint32_t)nPixels, root->outBuf, true);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, root->outBuf, nPixels * 3u * sizeof(float), hipMemcpyDeviceToHost, root->ownStream));
  HIP_TRY(hipStreamSynchronize(root->ownStream));
  for (uint32_t i = 0; i < count; i++) {
    HIP_TRY(hipSetDevice(scenes[i]->device));
    if ((rc = queueVerdict(scenes[i]))) return rc;
  }
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->upload_s = root->uploadSeconds;
    stats->render_s = nowSeconds() - t0;  // whole job: slowest part + gather + resolve + copy back
  }
  return PTX_OK;
}

int ptx_render_counts_multi(ptx_scene* const* scenes, uint32_t count, const ptx_camera* cam, const ptx_params* frame, uint32_t* out,
                            ptx_stats* stats) {
  if (!cam || !frame || !out) return fail(PTX_EINVAL, "null argument");
  const size_t nPixels = (size_t)frame->width * frame->height;
  const size_t plane = nPixels * sizeof(uint32_t);
  int rc = multiSetup(scenes, count, plane * count);
  if (rc) return rc;
  ptx_scene* root = scenes[0];
  const double t0 = nowSeconds();
  for (uint32_t i = 0; i < count; i++) {
    ptx_scene* s = scenes[i];
    ptx_params p = *frame;
    p.part_index = i;  // tile partition: 8x8 tiles with tile % count == i, zeros elsewhere
    p.part_count = count;
    FrameConstants fc;
    if ((rc = makeFrame(s, cam, &p, true, fc))) return rc;
    HIP_TRY(hipSetDevice(s->device));
    uint32_t* dst = (uint32_t*)root->dGather;
    if (i != 0) {
      if ((rc = ensureOut(s, plane))) return rc;
      dst = (uint32_t*)s->outBuf;
    }
    if ((rc = launchCounts(s, fc, dst, s->ownStream, false))) return rc;
    if (i != 0 && (rc = sendToRoot(root, s, i, dst, plane))) return rc;
  }
  HIP_TRY(hipSetDevice(root->device));
  if ((rc = ensureOut(root, plane))) return rc;
  sumCountsKernel<<<dim3((uint32_t)((nPixels + 255u) / 256u)), dim3(256), 0, root->ownStream>>>((const uint32_t*)root->dGather, count,
                                                                                                  (uint32_t)nPixels, (uint32_t*)root->outBuf);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, root->outBuf, plane, hipMemcpyDeviceToHost, root->ownStream));
  HIP_TRY(hipStreamSynchronize(root->ownStream));
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->upload_s = root->uploadSeconds;
    stats->render_s = nowSeconds() - t0;
  }
  return PTX_OK;
}

int ptx_measure(ptx_scene* s, const ptx_camera* cam, const ptx_params* prm, int heatmap, ptx_stats* stats) {
  FrameConstants fc;
  int rc = makeFrame(s, cam, prm, heatmap != 0, fc);
  if (rc) return rc;
  if (!stats) return fail(PTX_EINVAL, "null stats");
  HIP_TRY(hipSetDevice(s->device));
  const size_t bytes = (size_t)prm->width * prm->height * 4u * sizeof(float);
  if ((rc = ensureOut(s, bytes))) return rc;
  HIP_TRY(hipMemset(s->dStats, 0, sizeof(DeviceStats)));
  if (heatmap) rc = launchCounts(s, fc, (uint32_t*)s->outBuf, nullptr, true);
  else rc = launchRadiance(s, fc, (float*)s->outBuf, false, nullptr, true);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  memset(stats, 0, sizeof(*stats));
  stats->upload_s = s->uploadSeconds;
  if ((rc = kernelSeconds(s, &stats->render_s))) return rc;
  return readStats(s, stats);
}

int ptx_last_kernel_seconds(ptx_scene* s, double* seconds) {
  if (!s || !seconds) return fail(PTX_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  return kernelSeconds(s, seconds);
}

}  // extern "C"

#if defined(PTX_QUEUE_DEBUG) && defined(PTX_EXPERIMENT_QUEUES)
// Profiling builds only: the event counters of the last queueKernel launch through this handle (queue_kernel.hpp).
extern "C" int ptx_debug_queue_counters(ptx_scene* s, unsigned int* out16) {
  if (!s || !s->dQueue) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpy(out16, s->dQueue + queueCtlBytes(queueShards()), 64, hipMemcpyDeviceToHost) != hipSuccess) return -2;
  return 0;
}
#endif

#ifdef PTX_POOL_PROFILE
// Profiling builds only: reads and clears the pool kernel's region profile (pool_kernel.hpp).
extern "C" int ptx_debug_pool_profile(unsigned long long* out32) {
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(ptx::gPoolProfile), sizeof(unsigned long long) * 32) != hipSuccess) return -2;
  unsigned long long zero[32] = {};
  if (hipMemcpyToSymbol(HIP_SYMBOL(ptx::gPoolProfile), zero, sizeof(zero)) != hipSuccess) return -2;
  return 0;
}
#endif

#ifdef PTX_POOL_PROFILE
extern "C" int ptx_debug_pool_waves(unsigned long long* out, unsigned waves) {
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (waves > 8192u) waves = 8192u;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ptx::gPoolWaves), sizeof(unsigned long long) * 8u * waves) != hipSuccess) return -2;
  return 0;
}
#endif

#ifdef PTX_WAVE_PROFILE
// Profiling builds only (not declared in include/ptx_backend.h): reads and clears the wave profile.
extern "C" int ptx_debug_wave_profile(unsigned long long* out32) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(ptx::gWaveProfile), sizeof(unsigned long long) * 32) != hipSuccess) return -2;
  unsigned long long zero[32] = {};
  if (hipMemcpyToSymbol(HIP_SYMBOL(ptx::gWaveProfile), zero, sizeof(zero)) != hipSuccess) return -2;
  return 0;
}
#endif

#if defined(PTX_WAVE_PROFILE) && defined(PTX_EXPERIMENT_PHASES)
extern "C" int ptx_debug_sp_hist(unsigned long long* out128) {
  if (hipMemcpyFromSymbol(out128, HIP_SYMBOL(ptx::gSpHist), sizeof(unsigned long long) * 128) != hipSuccess) return -2;
  return 0;
}
#endif

#ifdef PTX_VISIT_HIST
// Profiling builds only: node steps per pair index of the instrumented launches so far; also reports
// where the TLAS pairs start and how many pairs there are (pairs are stored TLAS first).
extern "C" int ptx_debug_visit_hist(unsigned int* out, unsigned int n) {
  if (n > (1u << 20)) n = 1u << 20;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ptx::gVisitHist), sizeof(unsigned int) * n) != hipSuccess) return -2;
  return 0;
}
#endif
