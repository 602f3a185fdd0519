// queueKernel: the radiance path as extend / shade roles around two ray queues in HBM (north_star's
// "wavefront-style ray queue"; replaces the loop of gpuRender.cu:24-116 and the per-pixel chain of :130-156).
//
// pathKernel (kernels.hpp) ties a pixel's chain of samples to one lane and shades inside the traversal wave
// with whatever lanes happen to wait (16 of 64 on the bench frame).  Here a CHAIN -- one work item at a time: a
// pixel, or one RNG stream of a pixel, with its throughput, RNG state, sample and bounce counters and running sum
// -- lives in memory (arrays P0..P2 indexed by chain id) and moves between lanes one path segment at a time:
//
//   rayQ    ring of rays ready to be traced            {origin, direction, chain id}        2 x 16 B
//   shadeQ  ring of finished rays waiting for shading  {ray, t, u, v, triangle, instance, chain id}  4 x 16 B
//
//   extend role   every lane of the persistent grid traces one ray, exactly as in pathKernel's wide radiance
//                 path (same state machine, same arithmetic).  When PTX_QUEUE_SERVE_LANES lanes of a wave have
//                 finished, the wave delivers their hit records to shadeQ and takes new rays from rayQ.
//   shade role    a wave that sees 64 entries waiting in shadeQ (or that starves) claims a batch, one entry per
//                 lane, parks its own traversal state, shades all of them at once -- hit shading, environment
//                 lookup, accumulation, next sample's camera ray, next work item from the work counter -- and
//                 posts the next rays to rayQ.  Shading runs with (nearly) all 64 lanes instead of 16.
//
// Chains in flight = chainsPerLane x lanes of the grid (about all 1.2 M pixels of the bench frame at once), so
// with the reference's seeding -- one serial RNG stream per pixel -- every pixel's chain advances together instead
// of 3 pixels per lane one after the other (the frame's tail of pathKernel: DESIGN.md).
//
// Visibility between workgroups inside the launch (gfx950: L2 per XCD, L1 per CU; cdna_hip_programming.md
// Guideline 16): every byte that changes hands is written by a write-through (sc1) store and read by an L1-
// bypassing (sc1) load; counters are agent-scope atomics.  Ring entries validate themselves: each 16-byte word
// carries the lap tag of its ring position, a consumer re-reads its entry until all tags match.  Chain records
// have no tag: their stores are drained (s_waitcnt vmcnt(0)) before the ray that hands the chain on is posted.
// Every spin is bounded; a spin that runs out raises the error word, all waves leave, the host reports PTX_EHIP.
// Per chain the order of RNG draws, of the floating-point additions into the sum and every rounding are those of
// pathKernel, so the image is bit-identical to it for equal seeds and streams.
#pragma once

#include "../path-tracer_amd/csrc/kernels.hpp"
#include "../path-tracer_amd/csrc/pool_kernel.hpp"  // laneRank, fbits, bitsf

namespace ptx {

#ifndef PTX_QUEUE_MIN_WAVES
#define PTX_QUEUE_MIN_WAVES 6
#endif
#ifndef PTX_QUEUE_LDS_STACK
#define PTX_QUEUE_LDS_STACK 26
#endif
#ifndef PTX_QUEUE_SERVE_LANES
#define PTX_QUEUE_SERVE_LANES 16   // finished lanes that make a wave deliver and refill
#endif
#ifndef PTX_QUEUE_COOLDOWN
#define PTX_QUEUE_COOLDOWN 8      // iterations a starved wave waits before it asks the queues again
#endif
constexpr uint32_t kQStack = PTX_QUEUE_LDS_STACK;
constexpr uint32_t kQParkSlots = 26u;       // dwords of traversal state a wave parks around a shading batch
constexpr uint32_t kQMaxShards = 16u;
constexpr uint32_t kQSpinLimit = 1u << 18;  // re-reads of a ring entry before the launch is declared broken
constexpr unsigned long long kQIdleTicks = 500000000ull;  // 5 s of the 100 MHz clock without progress anywhere

// Scratch of one launch (one allocation, 256-byte aligned parts; byte offsets from `base`, < 4 GiB in all):
//   control words   line 0: work counter, finished items, error; then per shard four lines: rayHead, rayTail,
//                   shadeHead, shadeTail (every counter on a 128-byte line of its own)
//   ring words      [shard][1 << ringLog2] x 16 B each: ray0 {o.xyz, tag} ray1 {d.xyz, tag8 | id24}
//                   shade0 {o.xyz, tag} shade1 {d.xyz, tag} shade2 {t, u, v, tag} shade3 {tri, inst, id, tag}
//   chain records   [chains] x 16 B each: p0 {col.xyz, rng} p1 {pixel | stream, samples left, bounce, -} p2 {sum.xyz, -}
// The control words and the ring words are zeroed before every launch (a tag is never 0).
struct QueueLayout {
  unsigned char* base;
  uint32_t totalBytes;
  uint32_t ringLog2, shards, chainsPerLane;
  uint32_t ray0, ray1, shade0, shade1, shade2, shade3, p0, p1, p2;
};
constexpr uint32_t kCtlWork = 0u, kCtlDone = 1u, kCtlError = 2u;
__host__ __device__ inline uint32_t ctlWord(uint32_t shard, uint32_t which) { return 32u * (1u + 4u * shard + which); }  // 0 rayHead 1 rayTail 2 shadeHead 3 shadeTail
__host__ __device__ inline uint32_t queueCtlBytes(uint32_t shards) { return 128u * (1u + 4u * shards); }

struct QueueKernelArgs {
  DeviceScene sc;
  FrameConstants fc;
  QueueLayout q;
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t __attribute__((address_space(1))) * CtlWords;
#define PTX_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// 16-byte sc1 accesses through a buffer descriptor (aux bit 4 = sc1; bit 31 = volatile: a poll re-reads)
#define PTX_QLOAD(rsrc, off) __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off), 0, (int)(16u | 0x80000000u))
__device__ __forceinline__ void queueStore(__amdgpu_buffer_rsrc_t rsrc, uint32_t off, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  const u32x4 v = {a, b, c, d};
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)off, 0, 16);
}

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// Profiling builds (-DPTX_QUEUE_DEBUG): per-wave event counts, added to 16 words behind the control words when a wave leaves
// (read with ptx_debug_queue_counters): 0 services, 1 pop claims tried, 2 pop claims lost, 3 rays taken, 4 batch claims tried,
// 5 batch claims lost, 6 batches, 7 lanes in batches, 8 re-reads of ray entries, 9 re-reads of shade entries, 10 idle rounds,
// 11 loop iterations, 12 starved services, 13 rays delivered
#ifdef PTX_QUEUE_DEBUG
#define PTX_QDBG(k, n) do { dbg[k] += (n); } while (0)
#else
#define PTX_QDBG(k, n) do {} while (0)
#endif

template <bool STATS>
__global__ void __launch_bounds__(kBlockThreads, PTX_QUEUE_MIN_WAVES) queueKernel(DeviceScene sc, FrameConstants fcArg, QueueLayout qArg,
                                                                                   void* __restrict__ outRaw, DeviceStats* stats,
                                                                                   uint32_t* stackOverflow) {
  (void)fcArg; (void)qArg;
  __shared__ uint32_t sStack[kQStack * kBlockThreads];
  StackMem stk;
  stk.lds = sStack + threadIdx.x;
  stk.stride = gridDim.x * kBlockThreads;
  uint32_t* const parkGlobal = stackOverflow + (blockIdx.x * kBlockThreads + threadIdx.x);
  stk.overflow = parkGlobal + (size_t)kQParkSlots * stk.stride;
  const uint32_t lane = threadIdx.x & 63u;

  // ---- lane state: the ray this lane traces ----
  uint32_t ref = kRefIdle;  // kRefIdle here: the lane holds no ray; kRefDone: its ray is finished and waits for delivery
  uint32_t sp = 1u;
  uint32_t chain = 0u;      // chain id of the ray
  bool zeroDir = false, wZeroDir = false, inInst = false;
  uint32_t tlasLeaf = 0u, curInst = 0u;
  vec3 wo(0.0f), wd(0.0f, 0.0f, -1.0f), winv(0.0f);
  vec3 o(0.0f), d(0.0f, 0.0f, -1.0f), inv(0.0f);
  Hit hit;
  hit.t = FLT_MAX; hit.u = 0.0f; hit.v = 0.0f; hit.tri = 0xFFFFFFFFu; hit.inst = 0xFFFFFFFFu;

  uint32_t nRays = 0, nNodeSteps = 0, nTriTests = 0, nInstEntries = 0, nHits = 0, nMisses = 0;
  unsigned long long nBoxHits = 0;
  uint32_t count = 0;

  // ---- wave state (uniform) ----
  uint32_t birthsDone = 0u;   // chains this wave's lanes have started (the prologue: chainsPerLane batches of births)
  uint32_t cooldown = 0u;     // iterations until a starved wave asks again
  uint32_t popShard = uniform(blockIdx.x);  // reduced modulo the shard count where it is used
  unsigned long long idleSince = 0ull;
  uint32_t idleMark = 0u;
  bool leave = false;
#ifdef PTX_QUEUE_DEBUG
  uint32_t dbg[16] = {};
#endif

  for (;;) {
    PTX_QDBG(11, 1u);
    const bool fin = ref == kRefDone, vacant = ref == kRefIdle;
    const unsigned long long finMask = __ballot(fin), vacantMask = __ballot(vacant);
    const uint32_t nFin = (uint32_t)__popcll(finMask), nVacant = (uint32_t)__popcll(vacantMask);
    bool serve = nFin >= PTX_QUEUE_SERVE_LANES || nFin + nVacant == 64u;
    if (nFin + nVacant >= PTX_QUEUE_SERVE_LANES && cooldown == 0u) serve = true;
    if (cooldown) cooldown--;
    if (serve) {
      const FrameConstants fc = loadKernelArgument<FrameConstants>((uint32_t)offsetof(QueueKernelArgs, fc));
      const QueueLayout q = loadKernelArgument<QueueLayout>((uint32_t)offsetof(QueueKernelArgs, q));
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(q.base, 0, (int)q.totalBytes, 0x00020000);
      CtlWords ctl = (CtlWords)q.base;
      const uint32_t ringMask = (1u << q.ringLog2) - 1u;
      const uint32_t home = uniform(blockIdx.x % q.shards);
      const uint32_t ps = popShard % q.shards;
      const uint32_t nPixels = fc.width * fc.height;
      const uint32_t totalWork = nPixels * fc.subStreams;
      const bool births = birthsDone < q.chainsPerLane;
      PTX_QDBG(0, 1u);

      // ---- the counters: one round trip for the delivery's reservation and a look at the queues, one more for the claims ----
      const unsigned long long wantMask = finMask | vacantMask;
      const uint32_t nWant = nFin + nVacant;
      uint32_t pushBase = 0u, nPop = 0u, popBase = 0u, nShade = 0u, shadeBase = 0u;
      {
        uint32_t rh = 0u, rt = 0u, sh = 0u, st = 0u, err = 0u;
        if (lane == 0u) {
          if (nFin) pushBase = __hip_atomic_fetch_add(ctl + ctlWord(home, 3u), nFin, PTX_RLX_AGENT);
          if (!births) {
            rh = __hip_atomic_load(ctl + ctlWord(ps, 0u), PTX_RLX_AGENT);
            rt = __hip_atomic_load(ctl + ctlWord(ps, 1u), PTX_RLX_AGENT);
            sh = __hip_atomic_load(ctl + ctlWord(ps, 2u), PTX_RLX_AGENT);
            st = __hip_atomic_load(ctl + ctlWord(ps, 3u), PTX_RLX_AGENT);
            err = __hip_atomic_load(ctl + kCtlError, PTX_RLX_AGENT);
            const int32_t avail = (int32_t)(rt - rh), backlog = (int32_t)(st - sh);
            uint32_t np = avail > 0 ? min((uint32_t)avail, nWant) : 0u;
            // a full batch is waiting, or this wave will have lanes with nothing to do
            uint32_t ns = (backlog >= 64 || (np < nWant && backlog > 0)) ? min((uint32_t)backlog, 64u) : 0u;
            uint32_t er = rh, es = sh;
            bool okr = false, oks = false;
            if (np) okr = __hip_atomic_compare_exchange_strong(ctl + ctlWord(ps, 0u), &er, rh + np, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ns) oks = __hip_atomic_compare_exchange_strong(ctl + ctlWord(ps, 2u), &es, sh + ns, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nPop = okr ? np : 0u;
            nShade = oks ? ns : 0u;
            popBase = rh;
            shadeBase = sh;
          }
          if (err != 0u) nPop = 0xFFFFFFFFu;  // somebody declared the launch broken
        }
        pushBase = uniform(pushBase);
        nPop = uniform(nPop);
        popBase = uniform(popBase);
        nShade = uniform(nShade);
        shadeBase = uniform(shadeBase);
        if (nPop == 0xFFFFFFFFu) break;
#ifdef PTX_QUEUE_DEBUG
        {
          const uint32_t triedPop = uniform(lane == 0u && !births && (int32_t)(rt - rh) > 0 && nWant ? 1u : 0u);
          const uint32_t triedShade = uniform(lane == 0u && !births && ((int32_t)(st - sh) >= 64 || ((int32_t)(rt - rh) < (int32_t)nWant && (int32_t)(st - sh) > 0)) ? 1u : 0u);
          PTX_QDBG(1, triedPop); PTX_QDBG(2, triedPop && !nPop ? 1u : 0u);
          PTX_QDBG(4, triedShade); PTX_QDBG(5, triedShade && !nShade ? 1u : 0u);
          PTX_QDBG(3, nPop); PTX_QDBG(13, nFin);
        }
#endif
      }

      // ---- (A) deliver the finished rays: one shadeQ entry each ----
      if (fin) {
        const uint32_t pos = pushBase + laneRank(finMask);
        const uint32_t off = (home << q.ringLog2 | (pos & ringMask)) * 16u;
        const uint32_t tag = 0x80000000u | (pos >> q.ringLog2);
        queueStore(rs, q.shade0 + off, fbits(wo.x), fbits(wo.y), fbits(wo.z), tag);
        queueStore(rs, q.shade1 + off, fbits(wd.x), fbits(wd.y), fbits(wd.z), tag);
        queueStore(rs, q.shade2 + off, fbits(hit.t), fbits(hit.u), fbits(hit.v), tag);
        queueStore(rs, q.shade3 + off, hit.tri, hit.inst, chain, tag);
        if (STATS) { nRays++; nBoxHits += count; }
        ref = kRefIdle;
      }

      // ---- (B) new rays for the lanes that hold none ----
      if (nPop) {
        const uint32_t rank = laneRank(wantMask);
        const bool take = (fin || vacant) && rank < nPop;
        const uint32_t pos = popBase + rank;
        const uint32_t off = (ps << q.ringLog2 | (pos & ringMask)) * 16u;
        const uint32_t tag = 0x80000000u | (pos >> q.ringLog2);
        const uint32_t tag8 = 0x80u | ((pos >> q.ringLog2) & 0x7Fu);
        u32x4 w0, w1;
        for (uint32_t spins = 0u;; spins++) {
          w0 = PTX_QLOAD(rs, q.ray0 + off);
          w1 = PTX_QLOAD(rs, q.ray1 + off);
          const bool ok = !take || (w0.w == tag && (w1.w >> 24) == tag8);
          if (__all(ok)) break;
          PTX_QDBG(8, 1u);
          if (spins > kQSpinLimit) {
            if (lane == 0u) __hip_atomic_store(ctl + kCtlError, 1u, PTX_RLX_AGENT);
            leave = true;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (take && !leave) {
          wo = vec3(bitsf(w0.x), bitsf(w0.y), bitsf(w0.z));
          wd = vec3(bitsf(w1.x), bitsf(w1.y), bitsf(w1.z));
          chain = w1.w & 0x00FFFFFFu;
        }
        // launch the segments: Scene::intersect starts at the TLAS root (pathKernel: S_LAUNCH)
        if (__any(take)) {
          vec3 ninv;
          bool nz;
          reciprocalDir(wd, ninv, nz);  // lanes that keep their ray compute their own values again and drop them
          if (take) {
            winv = ninv; wZeroDir = nz; zeroDir = nz;
            o = wo; d = wd; inv = winv;
            hit.t = FLT_MAX; hit.u = 0.0f; hit.v = 0.0f; hit.tri = 0xFFFFFFFFu; hit.inst = 0xFFFFFFFFu;
            tlasLeaf = 0u;
            count = 0u;
            inInst = false;
            stk.lds[0] = kRefDone;
            sp = 1u;
            ref = sc.tlasRootRefWide;
          }
        }
      }
      const bool starved = !births && nPop < nWant;
      if (starved) {
        PTX_QDBG(12, 1u);
        cooldown = PTX_QUEUE_COOLDOWN;
        popShard++;  // look into another shard next time
      }

      if ((births || nShade) && !leave) {
        // ================================ shading batch ================================
        // the wave's own rays wait in memory meanwhile (pathKernel: PTX_PARK)
        {
          const uint32_t pv[kQParkSlots] = {fbits(o.x),  fbits(o.y),  fbits(o.z),  fbits(d.x),    fbits(d.y),    fbits(d.z),    fbits(inv.x),
                                            fbits(inv.y), fbits(inv.z), fbits(hit.t), fbits(hit.u), fbits(hit.v), hit.tri,       hit.inst,
                                            tlasLeaf,    curInst,     fbits(wo.x), fbits(wo.y),   fbits(wo.z),   fbits(wd.x),   fbits(wd.y),
                                            fbits(wd.z), fbits(winv.x), fbits(winv.y), fbits(winv.z), chain};
          GlobalWords pg = (GlobalWords)parkGlobal;
          asm volatile("" : "+v"(pg));
#pragma unroll
          for (uint32_t i = 0; i < kQParkSlots; i++) { *pg = pv[i]; pg += stk.stride; }
        }

        const bool valid = births || lane < nShade;
        PTX_QDBG(6, 1u); PTX_QDBG(7, births ? 64u : nShade);
        bool birth = births;
        vec3 ro(0.0f), rd(0.0f, 0.0f, -1.0f);
        Hit h;
        h.t = FLT_MAX; h.u = 0.0f; h.v = 0.0f; h.tri = 0xFFFFFFFFu; h.inst = 0xFFFFFFFFu;
        uint32_t id = birthsDone * stk.stride + blockIdx.x * kBlockThreads + threadIdx.x;  // births: chain ids by thread
        if (!births) {
          const uint32_t pos = shadeBase + lane;
          const uint32_t off = (ps << q.ringLog2 | (pos & ringMask)) * 16u;
          const uint32_t tag = 0x80000000u | (pos >> q.ringLog2);
          u32x4 w0, w1, w2, w3;
          for (uint32_t spins = 0u;; spins++) {
            w0 = PTX_QLOAD(rs, q.shade0 + off);
            w1 = PTX_QLOAD(rs, q.shade1 + off);
            w2 = PTX_QLOAD(rs, q.shade2 + off);
            w3 = PTX_QLOAD(rs, q.shade3 + off);
            const bool ok = !valid || (w0.w == tag && w1.w == tag && w2.w == tag && w3.w == tag);
            if (__all(ok)) break;
            PTX_QDBG(9, 1u);
            if (spins > kQSpinLimit) {
              if (lane == 0u) __hip_atomic_store(ctl + kCtlError, 2u, PTX_RLX_AGENT);
              leave = true;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
          ro = vec3(bitsf(w0.x), bitsf(w0.y), bitsf(w0.z));
          rd = vec3(bitsf(w1.x), bitsf(w1.y), bitsf(w1.z));
          h.t = bitsf(w2.x); h.u = bitsf(w2.y); h.v = bitsf(w2.z);
          h.tri = w3.x; h.inst = w3.y;
          id = w3.z;
        }
        if (!leave) {
          const vec3 camPos(fc.camPos[0], fc.camPos[1], fc.camPos[2]);
          uint32_t rng = 0u, pix = 0u, sample = 0u, bounce = 0u;
          vec3 col(1.0f), sum(0.0f);
          bool havePixel = false, pathEnded = true, live = valid, sumDirty = false;
          if (valid && !birth) {
            const u32x4 a = PTX_QLOAD(rs, q.p0 + id * 16u);
            const u32x4 b = PTX_QLOAD(rs, q.p1 + id * 16u);
            col = vec3(bitsf(a.x), bitsf(a.y), bitsf(a.z));
            rng = a.w; pix = b.x; sample = b.y; bounce = b.z;
            havePixel = true;
            if (h.t < FLT_MAX) {
              if (STATS) nHits++;
              shadeHit(sc, h, ro, rd, rng, col);  // writes the next ray into ro / rd
              bounce++;
              pathEnded = bounce >= fc.maxBounces;  // gpuRender.cu:29
            } else {
              if (STATS) nMisses++;
              col *= environmentLookup(sc, fc, rd);
            }
            if (pathEnded) {
              const u32x4 c = PTX_QLOAD(rs, q.p2 + id * 16u);
              sum = vec3(bitsf(c.x), bitsf(c.y), bitsf(c.z)) + col;
              sumDirty = true;
              sample--;
            }
          }
          if (valid && pathEnded) {
            // ---- item finished? write it and pull the next one (pathKernel: S_PULL) ----
            while (sample == 0u && live) {
              if (havePixel) {
                const uint32_t p = pix & 0x00FFFFFFu, k = pix >> 24;
                const uint32_t n = fc.sppBase + (k < fc.sppExtra ? 1u : 0u);
                reinterpret_cast<float4*>(outRaw)[(size_t)k * nPixels + p] = make_float4(sum.x, sum.y, sum.z, (float)n);
                atomicAdd((uint32_t*)q.base + kCtlDone, 1u);
              }
              const uint32_t w = atomicAdd((uint32_t*)q.base + kCtlWork, 1u);
              if (w >= totalWork) {
                live = false;
                havePixel = false;
                break;
              }
              uint32_t slot = w, k = 0u;
              if (fc.subStreams > 1u) {
                k = (uint32_t)((double)w * fc.invPixels);
                slot = w - k * nPixels;
                if ((int32_t)slot < 0) { k--; slot += nPixels; }
                else if (slot >= nPixels) { k++; slot -= nPixels; }
              }
              uint32_t x, y;
              workPixel(fc, slot, x, y);
              const uint32_t p = y * fc.width + x;
              pix = p | (k << 24);
              sum = vec3(0.0f);
              sumDirty = true;
              rng = fc.seedBase + p + k * 0x9E3779B9u;  // gpuRender.cu:130 (+ stream offset)
              havePixel = true;
              sample = fc.sppBase + (k < fc.sppExtra ? 1u : 0u);
              if (fc.maxBounces == 0u) {
                sum = vec3((float)sample);  // gpuRender.cu:27-29: no segment is traced, every path returns white
                sample = 0u;
              }
            }
            if (live) {
              // ---- next camera ray: gpuRender.cu:132-143 ----
              const uint32_t p = pix & 0x00FFFFFFu;
              const uint32_t py = divByWidth(fc, p), px = p - py * fc.width;
              float fx = (float)px, fy = (float)py;
              if (fc.jitter) {
                const vec2 xi = pt::getRandomVec2(rng);
                fx += 0.5f * (2.0f * xi.x - 1.0f);
                fy += 0.5f * (2.0f * xi.y - 1.0f);
              }
              ro = camPos;
              rd = primaryDirection(fc, fx, fy);
              col = vec3(1.0f);
              bounce = 0u;
            }
          }
          // ---- the chain goes on: its record, then (stores drained) its next ray ----
          const unsigned long long liveMask = __ballot(live);
          const uint32_t nLive = (uint32_t)__popcll(liveMask);
          uint32_t base = 0u;
          if (nLive) {
            if (lane == 0u) base = __hip_atomic_fetch_add(ctl + ctlWord(home, 1u), nLive, PTX_RLX_AGENT);
            base = uniform(base);
          }
          if (live) {
            queueStore(rs, q.p0 + id * 16u, fbits(col.x), fbits(col.y), fbits(col.z), rng);
            queueStore(rs, q.p1 + id * 16u, pix, sample, bounce, 0u);
            if (sumDirty) queueStore(rs, q.p2 + id * 16u, fbits(sum.x), fbits(sum.y), fbits(sum.z), 0u);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (live) {
            const uint32_t pos = base + laneRank(liveMask);
            const uint32_t off = (home << q.ringLog2 | (pos & ringMask)) * 16u;
            const uint32_t lap = pos >> q.ringLog2;
            queueStore(rs, q.ray0 + off, fbits(ro.x), fbits(ro.y), fbits(ro.z), 0x80000000u | lap);
            queueStore(rs, q.ray1 + off, fbits(rd.x), fbits(rd.y), fbits(rd.z), (0x80u | (lap & 0x7Fu)) << 24 | id);
          }
          if (births) birthsDone++;
          cooldown = 0u;  // there are rays in the queue now
        }
        // the wave's own rays come back
        {
          uint32_t pv[kQParkSlots];
          GlobalWords pg = (GlobalWords)parkGlobal;
          asm volatile("" : "+v"(pg));
#pragma unroll
          for (uint32_t i = 0; i < kQParkSlots; i++) { pv[i] = *pg; pg += stk.stride; }
          o = vec3(bitsf(pv[0]), bitsf(pv[1]), bitsf(pv[2]));
          d = vec3(bitsf(pv[3]), bitsf(pv[4]), bitsf(pv[5]));
          inv = vec3(bitsf(pv[6]), bitsf(pv[7]), bitsf(pv[8]));
          hit.t = bitsf(pv[9]); hit.u = bitsf(pv[10]); hit.v = bitsf(pv[11]);
          hit.tri = pv[12]; hit.inst = pv[13];
          tlasLeaf = pv[14]; curInst = pv[15];
          wo = vec3(bitsf(pv[16]), bitsf(pv[17]), bitsf(pv[18]));
          wd = vec3(bitsf(pv[19]), bitsf(pv[20]), bitsf(pv[21]));
          winv = vec3(bitsf(pv[22]), bitsf(pv[23]), bitsf(pv[24]));
          chain = pv[25];
        }
      } else if (!leave && nFin + nVacant == 64u && nPop == 0u) {
        // ---- (D) nothing to trace, nothing to shade: finished, or wait for the others ----
        uint32_t doneItems = 0u, err = 0u, progress = 0u;
        PTX_QDBG(10, 1u);
        if (lane == 0u) {
          doneItems = __hip_atomic_load(ctl + kCtlDone, PTX_RLX_AGENT);
          err = __hip_atomic_load(ctl + kCtlError, PTX_RLX_AGENT);
        }
        if (lane < 4u * q.shards) progress = __hip_atomic_load(ctl + 32u * (1u + lane), PTX_RLX_AGENT);
        for (int off = 32; off > 0; off >>= 1) progress += __shfl_xor(progress, off);
        doneItems = uniform(doneItems);
        err = uniform(err);
        progress = uniform(progress);
        if (err != 0u || doneItems >= totalWork) {
          leave = true;
        } else {
          const unsigned long long now = wall_clock64();
          if (idleSince == 0ull || progress != idleMark) {
            idleSince = now;
            idleMark = progress;
          } else if (now - idleSince > kQIdleTicks) {
            if (lane == 0u) __hip_atomic_store(ctl + kCtlError, 3u, PTX_RLX_AGENT);
            leave = true;
          }
          __builtin_amdgcn_s_sleep(32);
          cooldown = 0u;
        }
      }
      if (!leave && (births || nShade || nPop)) idleSince = 0ull;
      if (leave) break;
    }

    // ============================== one traversal step per lane (pathKernel, wide radiance path) ==============================
    {
      const bool isNode = (int32_t)ref >= 0;
      const bool isSpecial = ref >= kRefIdle;
      const bool isTri = (ref >> 30) == 2u;
      bool isEnter = (ref >> 30) == 3u && !isSpecial;
      if (PTX_ENTER_THRESHOLD > 1) {
        const uint32_t nEnter = (uint32_t)__popcll(__ballot(isEnter));
        if (nEnter < PTX_ENTER_THRESHOLD && __any(isNode || isTri)) isEnter = false;
      }
      const bool steps = isNode || isTri || isEnter;
      const uint32_t first = ref & (isTri ? 0x007FFFFFu : 0x00FFFFFFu);
      const uint32_t nPrim = (isTri ? ((ref >> 23) & 0x7Fu) : ((ref >> 24) & 0x3Fu)) + 1u;

      const bool deep = __any(sp + 4u >= kQStack);
      const uint32_t top = stackRead<kQStack>(stk, sp ? sp - 1u : 0u, deep);

      const bool fetch3 = isNode || isTri || isEnter;
      const bool fetch4 = isNode || isEnter;
      const uint32_t off16 = isNode ? sc.wide16 + ref * 4u : (isTri ? sc.tris16 + first * 3u : sc.inst16 + first * 4u);
      const float4* rp = sc.arena + off16;
      float4 r0, r1, r2, r3;
      asm volatile("" : "=v"(r0.x), "=v"(r0.y), "=v"(r0.z), "=v"(r0.w), "=v"(r1.x), "=v"(r1.y), "=v"(r1.z), "=v"(r1.w));
      asm volatile("" : "=v"(r2.x), "=v"(r2.y), "=v"(r2.z), "=v"(r2.w), "=v"(r3.x), "=v"(r3.y), "=v"(r3.z), "=v"(r3.w));
      if (fetch3) { r0 = rp[0]; r1 = rp[1]; r2 = rp[2]; }
      if (fetch4) r3 = rp[3];

      uint32_t newRef = ref;
      uint32_t pushVal = 0u;
      bool doPush = false, doPop = false;
      uint32_t push2 = 0u, push3 = 0u, nPush = 0u;
      if (isEnter) {
        tlasLeaf = (nPrim > 1u) ? makeTlasLeafRef(first + 1u, nPrim - 1u) : 0u;
        curInst = first;
        inInst = true;
        o = xformRows(r0, r1, r2, wo, 1.0f);
        d = xformRows(r0, r1, r2, wd, 0.0f);
        reciprocalDir(d, inv, zeroDir);
        if (STATS) nInstEntries++;
        newRef = __float_as_uint(r3.y);
      }
      if (isTri) {
        TriRec tri;
        tri.v0 = r0; tri.e1 = r1; tri.e2 = r2;
        float u = 0.0f, v = 0.0f;
        const float dist = triangleTest(o, d, tri, u, v);
        if (dist < hit.t) { hit.t = dist; hit.u = u; hit.v = v; hit.tri = first; hit.inst = curInst; }
        if (STATS) nTriTests++;
        doPop = nPrim == 1u;
        newRef = doPop ? top : makeLeafRef(first + 1u, nPrim - 1u);
      }
      if (isNode) {
        uint32_t sr[4];
        uint32_t nHit;
        if (__any(zeroDir)) nHit = wideStep<true>(o, inv, hit.t, r0, r1, r2, r3, sr);
        else nHit = wideStep<false>(o, inv, hit.t, r0, r1, r2, r3, sr);
        if (STATS) { nNodeSteps++; count += nHit; }
        doPop = nHit == 0u;
        newRef = doPop ? top : sr[0];
        nPush = doPop ? 0u : nHit - 1u;
        pushVal = sr[1]; push2 = sr[2]; push3 = sr[3];
      }
      {
        const bool leaving = doPop && inInst && (top < sc.tlasNodeEnd[1] || top >= kTlasLeafFlag);
        if (__any(leaving)) {
          if (leaving) {
            o = wo; d = wd; inv = winv;
            zeroDir = wZeroDir;
            inInst = false;
            if (tlasLeaf != 0u) {
              newRef = tlasLeaf;
              doPop = false;
              tlasLeaf = 0u;
            }
          }
        }
      }
      if (steps) {
        if (!isNode) nPush = doPush ? 1u : 0u;
        const uint32_t v0 = nPush == 3u ? push3 : (nPush == 2u ? push2 : pushVal);
        const uint32_t v1 = nPush == 3u ? push2 : pushVal;
        stackWrite<kQStack>(stk, sp, v0, deep);
        if (__any(nPush >= 2u)) {
          stackWrite<kQStack>(stk, sp + 1u, v1, deep);
          stackWrite<kQStack>(stk, sp + 2u, pushVal, deep);
        }
        sp = sp + nPush - (doPop ? 1u : 0u);
        ref = newRef;
      }
    }
  }

#ifdef PTX_QUEUE_DEBUG
  if (lane == 0u) {
    const QueueLayout q = loadKernelArgument<QueueLayout>((uint32_t)offsetof(QueueKernelArgs, q));
    for (int k = 0; k < 16; k++) atomicAdd((uint32_t*)q.base + queueCtlBytes(q.shards) / 4u + k, dbg[k]);
  }
#endif
  if (STATS) {
    atomicAdd(&stats->rays, (unsigned long long)nRays);
    atomicAdd(&stats->nodeSteps, (unsigned long long)nNodeSteps);
    atomicAdd(&stats->triTests, (unsigned long long)nTriTests);
    atomicAdd(&stats->instEntries, (unsigned long long)nInstEntries);
    atomicAdd(&stats->hits, (unsigned long long)nHits);
    atomicAdd(&stats->misses, (unsigned long long)nMisses);
    atomicAdd(&stats->boxHits, nBoxHits);
  }
}

}  // namespace ptx
