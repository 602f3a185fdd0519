// poolKernel: the radiance path with a private ray pool per wave -- extend and shade as two roles of the same
// wave around two wave-local queues (north_star's "wavefront-style ray queue"; replaces the bounce loop of
// gpuRender.cu:24-116 and the per-pixel sample chain of :130-156).
//
// pathKernel (kernels.hpp) ties a pixel's chain of samples to one lane and shades inside the traversal loop with
// whatever lanes happen to wait: 16 of 64 on the bench frame, a fifth of the wave's time.  Here a wave owns
// W = 64 x chainsPerLane CHAINS -- a chain is one work item at a time: a pixel, or one RNG stream of a pixel,
// with its throughput, RNG state, sample and bounce counters and running sum -- whose state lives in the wave's
// own slice of a scratch buffer (word arrays indexed by chain slot), and two lists of slots in LDS:
//
//   ready    chains whose next ray waits to be traced        R0 {o.xyz, d.x}  R1 {d.y, d.z, -, -}
//   pending  chains whose ray is finished and waits for shading   H0 {t, u, v, triangle}  H1 {instance}
//
//   extend   every lane traces one ray, exactly as pathKernel's wide radiance path does (same state machine, same
//            arithmetic).  When a few lanes have finished, they write their hit records, their chains go to the
//            pending list, and they take the next rays from the ready list: a lane never waits for the shading of
//            its own chain, and a chain moves from lane to lane.
//   shade    when 64 chains are pending (or lanes run dry), the wave parks its traversal state and shades a batch,
//            one chain per lane: hit shading / environment lookup, accumulation, the next sample's camera ray, the
//            next work item from the global work counter.  The new rays go to the ready list.  Shading runs with
//            all 64 lanes instead of 16, a quarter as often.
//
// Everything a wave touches here is its own: no atomics but the work counter, nothing another workgroup reads during
// the launch, no waiting on anybody, so no spin and no time-out.  (The first form of this kernel kept the queues
// global and shared -- experiments/queue_kernel.hpp: images identical, 6 to 50 times slower, because thousands of
// waves claim from the same counters: profiles/r03/global_queues.txt.)
// Per chain the order of RNG draws, of the additions into the sum and every rounding are those of pathKernel, so the
// image is bit-identical to pathKernel's for equal seeds and streams.
#pragma once

#include "kernels.hpp"

namespace ptx {

#ifndef PTX_POOL_MIN_WAVES
#define PTX_POOL_MIN_WAVES 6
#endif
#ifndef PTX_POOL_LDS_STACK
#define PTX_POOL_LDS_STACK 24      // 24 KB of stack + 2 KB of lists per block: six blocks per CU as pathKernel
#endif
#ifndef PTX_POOL_SERVE_LANES
#define PTX_POOL_SERVE_LANES 12    // finished lanes that make a wave deliver and refill
#endif
#ifndef PTX_POOL_BATCH
#define PTX_POOL_BATCH 64          // pending chains that make a wave shade (a batch is at most 64: one chain per lane)
#endif
constexpr uint32_t kPStack = PTX_POOL_LDS_STACK;
constexpr uint32_t kPParkSlots = 14u;        // dwords of traversal state parked around a shading batch
constexpr uint32_t kPMaxChainsPerLane = 3u;  // slot ids are bytes; three lists of 192 bytes per wave fit beside the stack
constexpr uint32_t kPMaxSlots = 64u * kPMaxChainsPerLane;
constexpr uint32_t kPWordArrays = 6u;        // R0 R1 H0 P0 P1 P2 (16 bytes per slot each), then H1 (4 bytes per slot)
constexpr uint32_t kPSlotBytes = kPWordArrays * 16u + 4u;

// Scratch of one launch: `base` + wave * slots * kPSlotBytes is the slice of a wave (wave = its index in the grid).
struct PoolLayout {
  unsigned char* base;
  uint32_t chainsPerLane;  // 1 .. kPMaxChainsPerLane
  uint32_t staticPartition;  // 1: long items, and (nearly) all of them are taken in the first rounds: the waves keep pace by priority
};
struct PoolKernelArgs {  // the kernel arguments as the hardware lays them out (kernels.hpp: loadKernelArgument)
  DeviceScene sc;
  FrameConstants fc;
  PoolLayout q;
  uint32_t* workCounter;
  void* outRaw;
  DeviceStats* stats;
  uint32_t* stackOverflow;
};

// The traversal stack of this kernel: entries [0, kPStack) in LDS as in pathKernel; deeper entries and the park slots in
// the launch's overflow area, addressed as (uniform base) + 32-bit byte offset so that no lane holds a 64-bit address.
struct PoolStack {
  uint32_t* lds;            // &sStack[threadIdx.x]
  unsigned char* area;      // the overflow area (uniform)
  uint32_t laneBytes;       // 4 * global thread index
  uint32_t strideBytes;     // 4 * threads of the launch
};
__device__ __forceinline__ uint32_t* poolAreaWord(const PoolStack& m, uint32_t entry) {
  return reinterpret_cast<uint32_t*>(m.area + (size_t)(m.laneBytes + entry * m.strideBytes));
}
__device__ __forceinline__ uint32_t poolStackRead(const PoolStack& m, uint32_t slot, bool deep) {
  uint32_t v;
  if (!deep) {
    v = m.lds[slot * kBlockThreads];
  } else {
    const uint32_t a = m.lds[(slot < kPStack ? slot : kPStack - 1u) * kBlockThreads];
    const uint32_t b = *poolAreaWord(m, kPParkSlots + (slot < kPStack ? 0u : slot - kPStack));
    v = slot < kPStack ? a : b;
  }
  return v;
}
__device__ __forceinline__ void poolStackWrite(const PoolStack& m, uint32_t slot, uint32_t val, bool deep) {
  if (!deep) {
    m.lds[slot * kBlockThreads] = val;
  } else {
    if (slot < kPStack) m.lds[slot * kBlockThreads] = val;
    else *poolAreaWord(m, kPParkSlots + slot - kPStack) = val;
  }
}

// Pointers that reach the kernel through loadKernelArgument are generic to the compiler: these say "global memory"
// (global_load / global_store / global_atomic instead of the slower flat forms).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((address_space(1))) * GlobalQuads;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 __attribute__((address_space(1))) * GlobalPairs;
#ifndef PTX_POOL_NT
#define PTX_POOL_NT 0   // 1: chain records and parked state are read and written with the non-temporal hint
#endif
#if PTX_POOL_NT
#define PTX_PLD(p) __builtin_nontemporal_load(p)
#define PTX_PST(p, v) __builtin_nontemporal_store(v, p)
#else
#define PTX_PLD(p) (*(p))
#define PTX_PST(p, v) (*(p) = (v))
#endif
__device__ __forceinline__ u32x4 quad(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return (u32x4){a, b, c, d}; }
__device__ __forceinline__ u32x2 pair(uint32_t a, uint32_t b) { return (u32x2){a, b}; }

// The shading's copy of the scene description, re-read from the kernel-argument segment at a service so that its dozen
// pointers do not occupy SGPRs through the traversal loop.  Every array lives in the one arena: each pointer is rebuilt as
// (the kernel argument's arena pointer) + (its offset in the arena), which keeps it a GLOBAL pointer for the compiler
// (a pointer that comes out of loadKernelArgument alone is generic: flat loads).
template <typename T>
__device__ __forceinline__ const T* rebased(const float4* arena, const float4* arenaCopy, const T* p) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(arena) + (reinterpret_cast<const char*>(p) - reinterpret_cast<const char*>(arenaCopy)));
}
__device__ __forceinline__ DeviceScene shadingScene(const DeviceScene& sc, uint32_t argOffset) {
  DeviceScene s = loadKernelArgument<DeviceScene>(argOffset);
  const float4* a = s.arena;
  s.instShade = rebased(sc.arena, a, s.instShade); s.geomShade = rebased(sc.arena, a, s.geomShade);
  s.faceNormals = rebased(sc.arena, a, s.faceNormals); s.vtxNormals = rebased(sc.arena, a, s.vtxNormals);
  s.vtxUvs = rebased(sc.arena, a, s.vtxUvs); s.materials = rebased(sc.arena, a, s.materials);
  s.textures = rebased(sc.arena, a, s.textures); s.texels = rebased(sc.arena, a, s.texels);
  s.environment = rebased(sc.arena, a, s.environment);
  s.arena = sc.arena;
  return s;
}

__device__ __forceinline__ uint32_t laneRank(unsigned long long mask) {  // set bits of `mask` below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t uniformWord(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t fbits(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float bitsf(uint32_t u) { return __uint_as_float(u); }

// Profiling builds (-DPTX_POOL_PROFILE, read with ptx_debug_pool_profile): per wave, summed over the grid --
// [0..7] shader-clock ticks in: loop top, delivery, shading batch (with parking), refill, traversal step;
// [8..] counts: 8 iterations, 9 services, 10 batches, 11 chains shaded, 12 rays delivered, 13 rays taken,
// 14 lane-steps (lanes that step, summed over iterations), 15 lanes without a ray (summed over iterations);
// [16] the longest wave's ticks, [17] waves, [18] sum over waves of (ticks / 1024)^2
#ifdef PTX_POOL_PROFILE
__device__ unsigned long long gPoolProfile[32];
__device__ unsigned long long gPoolWaves[8192 * 8];  // per wave: ticks, iterations, lane-steps, batches, chains shaded, ticks in traversal, start tick, services
#define PTX_PSTAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); pprof[k] += now_ - plast; plast = now_; } while (0)
#define PTX_PCOUNT(k, n) do { pprof[k] += (n); } while (0)
#else
#define PTX_PSTAMP(k) do {} while (0)
#define PTX_PCOUNT(k, n) do {} while (0)
#endif

// PACED: the launch is a static partition (PoolLayout::staticPartition) and the waves keep pace by priority, see the shading
// batch; a kernel of its own (pacedPoolKernel) so that the other launches do not carry its registers through their service section.
template <bool STATS, bool PACED>
__device__ __forceinline__ void poolBody(const DeviceScene& sc, DeviceStats* stats, uint32_t* stackOverflow) {
  __shared__ uint32_t sStack[kPStack * kBlockThreads];
  __shared__ uint8_t sLists[3u * kPMaxSlots * (kBlockThreads / 64u)];  // per wave: ready (urgent), ready (others), pending
  PoolStack stk;
  stk.lds = sStack + threadIdx.x;
  stk.area = reinterpret_cast<unsigned char*>(stackOverflow);
  stk.laneBytes = 4u * (blockIdx.x * kBlockThreads + threadIdx.x);
  stk.strideBytes = 4u * gridDim.x * kBlockThreads;
  const uint32_t lane = threadIdx.x & 63u;
  uint8_t* const hotList = sLists + (threadIdx.x >> 6) * (3u * kPMaxSlots);
  uint8_t* const readyList = hotList + kPMaxSlots;
  uint8_t* const pendList = readyList + kPMaxSlots;

  // ---- lane state: the ray this lane traces ----
  uint32_t ref = kRefIdle;  // kRefIdle: the lane holds no ray; kRefDone: its ray is finished and waits for delivery
  uint32_t sp = 1u;
  uint32_t chain = 0u;      // slot of the chain the ray belongs to
  bool zeroDir = false, wZeroDir = false, inInst = false;
  uint32_t tlasLeaf = 0u, curInst = 0u;
  vec3 wo(0.0f), wd(0.0f, 0.0f, -1.0f), winv(0.0f);
  vec3 o(0.0f), d(0.0f, 0.0f, -1.0f), inv(0.0f);
  Hit hit;
  hit.t = FLT_MAX; hit.u = 0.0f; hit.v = 0.0f; hit.tri = 0xFFFFFFFFu; hit.inst = 0xFFFFFFFFu;

  uint32_t nRays = 0, nNodeSteps = 0, nTriTests = 0, nInstEntries = 0, nHits = 0, nMisses = 0;
  unsigned long long nBoxHits = 0;
  uint32_t count = 0;

  // ---- wave state (uniform; the seven list positions are bytes, packed into two words between services so that they hold two
  //      SGPRs through the traversal loop, not seven) ----
  //   birthsDone    rounds of 64 chains started so far (chainsPerLane of them)
  //   readyHead/Count, pendHead/Count, hotHead/Count    the three lists (rings of `slots` bytes)
  //   hotThreshold  Urgent chains.  With one RNG stream per pixel a pixel's samples are one serial chain, and the frame cannot
  //     end before its longest chain has: a chain that waits in the ready list while others are traced stretches exactly that.
  //     A chain whose expected remaining segments -- samples left x (segments so far + maxBounces) / (samples done + 1) --
  //     exceed the wave's threshold goes to the urgent list, which is served first; the threshold follows the share of urgent
  //     chains among those shaded, so that the urgent ones always fit the lanes.
  // LDS: 24 KB of stack + 3 x 192 bytes of lists per wave = 26,880 bytes per block, which is 21 allocation granules of
  // 1,280 bytes: six blocks per CU.  128 bytes more take a 22nd granule and the sixth block no longer fits, although the
  // occupancy query still answers six (measured: 301 -> 314 ms, the time of PTX_MAX_BLOCKS_PER_CU=5).
  // (PACED only: its extra state would otherwise push a lane mask of the traversal loop out of the SGPRs; the plain kernel is
  // better off with the seven as they are -- tools/kernel_resources.py --regions shows either choice)
  uint32_t listsA = 0u;  // PACED: readyHead | readyCount << 8 | pendHead << 16 | pendCount << 24
  uint32_t listsB = 0u;  // PACED: hotHead | hotCount << 8 | birthsDone << 16 | longest chain's remaining rays (of 256, decaying) << 24
  uint32_t kBirthsDone = 0u, kReadyHead = 0u, kReadyCount = 0u, kPendHead = 0u, kPendCount = 0u, kHotHead = 0u, kHotCount = 0u;  // !PACED
  float hotThreshold = 0.0f;
#ifndef PTX_POOL_PRIO_GAIN
#define PTX_POOL_PRIO_GAIN 25.0f   // priority levels per unit of lag (a lag of 4 % of the wave's work = one level)
#endif
#ifndef PTX_POOL_PRIO_PUBLISH
#define PTX_POOL_PRIO_PUBLISH 256u  // samples between two additions to the frame's count (a power of two)
#endif
#ifndef PTX_POOL_PRIO_C3
#define PTX_POOL_PRIO_C3 115u      // shares (of 256) of the longest possible item that the wave's longest chain has ahead: levels 3, 2, 1
#define PTX_POOL_PRIO_C2 77u
#define PTX_POOL_PRIO_C1 38u
#endif
  uint32_t waveProgress = 0u;  // PACED; bits 0-29: samples this wave has finished; bits 30-31: the priority its lag asks for (one SGPR for both)

#ifdef PTX_POOL_PROFILE
  unsigned long long pprof[16] = {};
  unsigned long long plast = __builtin_readcyclecounter();
  const unsigned long long pstart = plast;
#endif

  for (;;) {
    const bool fin = ref == kRefDone, vacant = ref == kRefIdle;
    const unsigned long long finMask = __ballot(fin), vacantMask = __ballot(vacant);
    const uint32_t nFin = (uint32_t)__popcll(finMask), nVacant = (uint32_t)__popcll(vacantMask);
    PTX_PCOUNT(8, 1u); PTX_PCOUNT(15, nVacant + nFin);
    PTX_PSTAMP(0);
    // deliver and refill when a few lanes have finished, or a quarter of the lanes that hold a ray (the wave's last chains)
    if (nFin >= PTX_POOL_SERVE_LANES || nFin * 4u >= 64u - nVacant) {
      PTX_MARK("SERVICE");
      const FrameConstants fc = loadKernelArgument<FrameConstants>((uint32_t)offsetof(PoolKernelArgs, fc));
      const PoolLayout q = loadKernelArgument<PoolLayout>((uint32_t)offsetof(PoolKernelArgs, q));
#ifndef PTX_POOL_RELOAD_SCENE
#define PTX_POOL_RELOAD_SCENE 0   // 1: the shading reads the scene description again at every service (fewer SGPRs held; measured 1 % slower)
#endif
#if PTX_POOL_RELOAD_SCENE
      const DeviceScene scs = shadingScene(sc, (uint32_t)offsetof(PoolKernelArgs, sc));
#else
      const DeviceScene& scs = sc;
#endif
      GlobalWords const workCounterS = (GlobalWords)loadKernelArgument<uint32_t*>((uint32_t)offsetof(PoolKernelArgs, workCounter));
      GlobalQuads const planes = (GlobalQuads)loadKernelArgument<void*>((uint32_t)offsetof(PoolKernelArgs, outRaw));
      const uint32_t slots = 64u * q.chainsPerLane;
      const uint32_t wave = uniformWord((blockIdx.x * kBlockThreads + threadIdx.x) >> 6);
      GlobalQuads const words = (GlobalQuads)(q.base + (size_t)wave * slots * kPSlotBytes);  // [array][slot]
      GlobalWords const instWords = (GlobalWords)(words + kPWordArrays * slots);               // H1 [slot]
      const uint32_t nPixels = fc.width * fc.height;
      const uint32_t totalWork = nPixels * fc.subStreams;
      uint32_t readyHead = kReadyHead, readyCount = kReadyCount, pendHead = kPendHead, pendCount = kPendCount;
      uint32_t hotHead = kHotHead, hotCount = kHotCount, birthsDone = kBirthsDone, longest = 0u;
      if (PACED) {
        readyHead = listsA & 255u; readyCount = (listsA >> 8) & 255u; pendHead = (listsA >> 16) & 255u; pendCount = listsA >> 24;
        hotHead = listsB & 255u; hotCount = (listsB >> 8) & 255u; birthsDone = (listsB >> 16) & 255u; longest = listsB >> 24;
      }
      const bool births = birthsDone < q.chainsPerLane;

      // ---- (A) the finished rays: hit record to the chain's slot, chain to the pending list ----
      if (fin) {
        PTX_PST(words + (2u * slots + chain), quad(fbits(hit.t), fbits(hit.u), fbits(hit.v), hit.tri));
        PTX_PST(instWords + chain, hit.inst);
        uint32_t at = pendHead + pendCount + laneRank(finMask);
        at = at >= slots ? at - slots : at;
        pendList[at] = (uint8_t)chain;
        if (STATS) { nRays++; nBoxHits += count; }
        ref = kRefIdle;
      }
      pendCount += nFin;
      const uint32_t nWant = nFin + nVacant;  // lanes without a ray now
      PTX_PCOUNT(9, 1u); PTX_PCOUNT(12, nFin);
      PTX_PSTAMP(1);

      // ---- (B) shading batch: 64 chains are pending, or lanes would stay without a ray ----
      const uint32_t nShade = births ? 64u : ((pendCount >= PTX_POOL_BATCH || readyCount + hotCount < nWant) ? min(pendCount, 64u) : 0u);
      if (nShade) {
        {
          // the wave's own rays wait in memory meanwhile (pathKernel: PTX_PARK; unconditional, or the compiler keeps the
          // registers); wo / wd are in the chain's slot already, the two reciprocal directions are computed again afterwards
          const uint32_t pv[kPParkSlots] = {fbits(o.x),   fbits(o.y),   fbits(o.z),   fbits(d.x), fbits(d.y), fbits(d.z), fbits(hit.t),
                                            fbits(hit.u), fbits(hit.v), hit.tri,      hit.inst,   tlasLeaf,   curInst,    chain};
          uint32_t off = stk.laneBytes;
          asm volatile("" : "+v"(off));  // the addresses are computed here, one after the other, not held across the loop
#pragma unroll
          for (uint32_t i = 0; i < kPParkSlots; i++) { PTX_PST(reinterpret_cast<uint32_t*>(stk.area + (size_t)off), pv[i]); off += stk.strideBytes; }
        }
        {
          const bool valid = lane < nShade;
          uint32_t slot = birthsDone * 64u + lane;  // births: the round's 64 new slots
          if (!births) {
            uint32_t at = pendHead + lane;
            at = at >= slots ? at - slots : at;
            slot = valid ? pendList[at] : 0u;
          }
          const vec3 camPos(fc.camPos[0], fc.camPos[1], fc.camPos[2]);
          vec3 ro(0.0f), rd(0.0f, 0.0f, -1.0f);
          uint32_t rng = 0u, pix = 0u, sample = 0u, bounce = 0u, segments = 0u;
          vec3 col(1.0f), sum(0.0f);
          bool havePixel = false, pathEnded = true, live = valid, sumDirty = false, endedSample = false;
          if (valid && !births) {
            const u32x4 a0 = PTX_PLD(words + (slot)), a1 = PTX_PLD(words + (slots + slot)), hh = PTX_PLD(words + (2u * slots + slot));
            const u32x4 pa = PTX_PLD(words + (3u * slots + slot)), pb = PTX_PLD(words + (4u * slots + slot));
            Hit h;
            h.t = bitsf(hh.x); h.u = bitsf(hh.y); h.v = bitsf(hh.z); h.tri = hh.w;
            h.inst = PTX_PLD(instWords + slot);
            ro = vec3(bitsf(a0.x), bitsf(a0.y), bitsf(a0.z));
            rd = vec3(bitsf(a0.w), bitsf(a1.x), bitsf(a1.y));
            col = vec3(bitsf(pa.x), bitsf(pa.y), bitsf(pa.z));
            rng = pa.w; pix = pb.x; sample = pb.y; bounce = pb.z; segments = pb.w + 1u;
            havePixel = true;
            if (h.t < FLT_MAX) {
              if (STATS) nHits++;
              shadeHit(scs, h, ro, rd, rng, col);  // writes the next ray into ro / rd
              bounce++;
              pathEnded = bounce >= fc.maxBounces;  // gpuRender.cu:29
            } else {
              if (STATS) nMisses++;
              col *= environmentLookup(scs, fc, rd);
            }
            endedSample = pathEnded;
            if (pathEnded) {
              const u32x4 pc = PTX_PLD(words + (5u * slots + slot));
              sum = vec3(bitsf(pc.x), bitsf(pc.y), bitsf(pc.z)) + col;
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
                planes[(size_t)k * nPixels + p] = quad(fbits(sum.x), fbits(sum.y), fbits(sum.z), fbits((float)n));
              }
              // one atomic per wave for the lanes that are here (the compiler's own aggregation does not see through the
              // reloaded pointer: 40.8 M single adds to one address took the sub-stream frame from 320 to 500 ms)
              const unsigned long long here = __ballot(true);
              uint32_t w = 0u;
              if (laneRank(here) == 0u) w = __hip_atomic_fetch_add(workCounterS, (uint32_t)__popcll(here), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              w = uniformWord(w) + laneRank(here);
              if (w >= totalWork) {
                live = false;
                havePixel = false;
                break;
              }
              uint32_t item = w, k = 0u;
              if (fc.subStreams > 1u) {
                k = (uint32_t)((double)w * fc.invPixels);
                item = w - k * nPixels;
                if ((int32_t)item < 0) { k--; item += nPixels; }
                else if (item >= nPixels) { k++; item -= nPixels; }
              }
              if (fc.scatterMul != 0u) item = (uint32_t)(((unsigned long long)item * fc.scatterMul) % nPixels);
              uint32_t x, y;
              workPixel(fc, item, x, y);
              const uint32_t p = y * fc.width + x;
              pix = p | (k << 24);
              sum = vec3(0.0f);
              sumDirty = true;
              rng = fc.seedBase + p + k * 0x9E3779B9u;  // gpuRender.cu:130 (+ stream offset)
              havePixel = true;
              sample = fc.sppBase + (k < fc.sppExtra ? 1u : 0u);
              segments = 0u;
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
          // ---- the chains that go on: record and next ray to the slot, slot to the ready list ----
          const uint32_t itemSamples = fc.sppBase + ((pix >> 24) < fc.sppExtra ? 1u : 0u);
          const float expectedLeft = (float)sample * (float)(segments + fc.maxBounces);      // x (samples done + 1)
          const bool urgent = live && expectedLeft > hotThreshold * (float)(itemSamples - sample + 1u);
          const unsigned long long liveMask = __ballot(live), urgentMask = __ballot(urgent), calmMask = liveMask & ~urgentMask;
          if (live) {
            PTX_PST(words + (slot), quad(fbits(ro.x), fbits(ro.y), fbits(ro.z), fbits(rd.x)));
            PTX_PST(words + (slots + slot), quad(fbits(rd.y), fbits(rd.z), 0u, 0u));
            PTX_PST(words + (3u * slots + slot), quad(fbits(col.x), fbits(col.y), fbits(col.z), rng));
            PTX_PST(words + (4u * slots + slot), quad(pix, sample, bounce, segments));
            if (sumDirty) PTX_PST(words + (5u * slots + slot), quad(fbits(sum.x), fbits(sum.y), fbits(sum.z), 0u));
            uint32_t at = urgent ? hotHead + hotCount + laneRank(urgentMask) : readyHead + readyCount + laneRank(calmMask);
            at = at >= slots ? at - slots : at;
            (urgent ? hotList : readyList)[at] = (uint8_t)slot;
          }
          {
            const uint32_t nUrgent = (uint32_t)__popcll(urgentMask), nLive = (uint32_t)__popcll(liveMask);
            hotCount += nUrgent;
            readyCount += nLive - nUrgent;
            // keep about a third of the shaded chains urgent
            if (nUrgent * 16u > nLive * 7u) hotThreshold = hotThreshold * 1.0625f + 0.25f;
            else if (nUrgent * 16u < nLive * 4u) hotThreshold *= 0.9375f;
          }
          if (PACED) {
            // Static partition (the reference's seeding at full size: every pixel is taken in the first rounds, nothing is left
            // to pull): all waves hold the same amount of work, but the issue arbiter favours the older waves of a SIMD, so the
            // young ones finish last on half-empty SIMDs (profiles/r03/wave_priority.txt: 380 ms against 336 ms).  Each wave
            // compares the share of its samples it has finished with the share the whole frame has finished (one counter next to
            // the work counter, added to every 256 samples, read here) and raises its priority (s_setprio) when it lags; the
            // four levels are dithered from batch to batch so that the average follows the lag.  Scheduling only: no result
            // depends on it.
            GlobalWords const progress = workCounterS + 32;
            const uint32_t nEnded = (uint32_t)__popcll(__ballot(endedSample));
            const uint32_t before = waveProgress & 0x3FFFFFFFu, waveDone = before + nEnded;
            const uint32_t publish = (waveDone & ~(PTX_POOL_PRIO_PUBLISH - 1u)) - (before & ~(PTX_POOL_PRIO_PUBLISH - 1u));  // the frame's count holds every wave's count rounded down
            uint32_t fair = waveProgress >> 30;
            if (publish != 0u) {
              // the addition returns what the other waves have added so far: no other read of the counter (a load of it at every
              // batch, from 6,144 waves, made the batches of scene 0 twice as long)
              uint32_t frameDone = 0u;
              if (lane == 0u) frameDone = __hip_atomic_fetch_add(progress, publish, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              frameDone = uniformWord(frameDone) + publish;
              const float mine = (float)waveDone / ((float)slots * (float)fc.sppBase);
              const float all = (float)frameDone / ((float)nPixels * (float)fc.sppBase);
              const float levelF = 1.5f + (all - mine) * PTX_POOL_PRIO_GAIN + 0.25f * (float)(pendHead & 3u);
              fair = levelF >= 3.0f ? 3u : (levelF <= 0.0f ? 0u : (uint32_t)levelF);
            }
            waveProgress = waveDone | (fair << 30);
            // The other way to be late: one chain with more rays ahead than the wave's lanes can take off it (scenes where a few
            // pixels cost many times the average).  Expected remaining rays of the wave's longest chain, as a share (of 256) of the
            // most an item can have; the maximum over the chains of this batch, decaying, stands for the maximum over all chains.
            float rem = live ? expectedLeft / (float)(itemSamples - sample + 1u) : 0.0f;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) rem = fmaxf(rem, __shfl_xor(rem, m, 64));
            const float most = (float)(fc.sppBase + 1u) * (float)(fc.maxBounces + 1u);
            const uint32_t now = min(255u, (uint32_t)(bitsf(uniformWord(fbits(rem))) * 256.0f / most));
            longest = max(longest - min(longest, 1u + (longest >> 5)), now);
          }
          if (births) {
            birthsDone++;
          } else {
            pendHead += nShade;
            pendHead = pendHead >= slots ? pendHead - slots : pendHead;
            pendCount -= nShade;
          }
        }
        {
          uint32_t pv[kPParkSlots];
          uint32_t off = stk.laneBytes;
          asm volatile("" : "+v"(off));
#pragma unroll
          for (uint32_t i = 0; i < kPParkSlots; i++) { pv[i] = PTX_PLD(reinterpret_cast<uint32_t*>(stk.area + (size_t)off)); off += stk.strideBytes; }
          o = vec3(bitsf(pv[0]), bitsf(pv[1]), bitsf(pv[2]));
          d = vec3(bitsf(pv[3]), bitsf(pv[4]), bitsf(pv[5]));
          hit.t = bitsf(pv[6]); hit.u = bitsf(pv[7]); hit.v = bitsf(pv[8]);
          hit.tri = pv[9]; hit.inst = pv[10];
          tlasLeaf = pv[11]; curInst = pv[12];
          chain = pv[13];
          const u32x4 a0 = PTX_PLD(words + (chain)), a1 = PTX_PLD(words + (slots + chain));  // a lane without a ray reads slot `chain` of old: unused
          wo = vec3(bitsf(a0.x), bitsf(a0.y), bitsf(a0.z));
          wd = vec3(bitsf(a0.w), bitsf(a1.x), bitsf(a1.y));
          // 1 / d as it was: the short and the long form of reciprocalDir agree bit for bit wherever the short one is used
          bool z0, z1;
          reciprocalDir(d, inv, z0);
          reciprocalDir(wd, winv, z1);
          (void)z0; (void)z1;
        }
      }

      if (nShade) { PTX_PCOUNT(10, 1u); PTX_PCOUNT(11, nShade); }
      PTX_PSTAMP(2);
      // ---- (C) the next rays for the lanes that hold none ----
      const uint32_t nTakeHot = min(nWant, hotCount);
      const uint32_t nTake = nTakeHot + min(nWant - nTakeHot, readyCount);
      PTX_PCOUNT(13, nTake);
      if (nTake) {
        const unsigned long long wantMask = finMask | vacantMask;
        const uint32_t rank = laneRank(wantMask);
        const bool take = (fin || vacant) && rank < nTake;
        if (take) {
          const bool fromHot = rank < nTakeHot;
          uint32_t at = fromHot ? hotHead + rank : readyHead + (rank - nTakeHot);
          at = at >= slots ? at - slots : at;
          chain = (fromHot ? hotList : readyList)[at];
          const u32x4 a0 = PTX_PLD(words + (chain)), a1 = PTX_PLD(words + (slots + chain));
          wo = vec3(bitsf(a0.x), bitsf(a0.y), bitsf(a0.z));
          wd = vec3(bitsf(a0.w), bitsf(a1.x), bitsf(a1.y));
        }
        hotHead += nTakeHot;
        hotHead = hotHead >= slots ? hotHead - slots : hotHead;
        hotCount -= nTakeHot;
        readyHead += nTake - nTakeHot;
        readyHead = readyHead >= slots ? readyHead - slots : readyHead;
        readyCount -= nTake - nTakeHot;
        // launch the segments: Scene::intersect starts at the TLAS root (pathKernel: S_LAUNCH)
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
      if (PACED) {
        const uint32_t chainLevel = longest > PTX_POOL_PRIO_C3 ? 3u : (longest > PTX_POOL_PRIO_C2 ? 2u : (longest > PTX_POOL_PRIO_C1 ? 1u : 0u));
        const uint32_t level = max(waveProgress >> 30, chainLevel);
        if (level == 3u) __builtin_amdgcn_s_setprio(3);
        else if (level == 2u) __builtin_amdgcn_s_setprio(2);
        else if (level == 1u) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
      PTX_PSTAMP(3);
      if (PACED) {
        listsA = readyHead | (readyCount << 8) | (pendHead << 16) | (pendCount << 24);
        listsB = hotHead | (hotCount << 8) | (birthsDone << 16) | (longest << 24);
      } else {
        kReadyHead = readyHead; kReadyCount = readyCount; kPendHead = pendHead; kPendCount = pendCount;
        kHotHead = hotHead; kHotCount = hotCount; kBirthsDone = birthsDone;
      }
      // no ray in any lane, none ready, none pending, every round of births made: the wave's chains are all finished
      if (!births && birthsDone >= q.chainsPerLane && nTake == 0u && nWant == 64u && pendCount == 0u && readyCount == 0u && hotCount == 0u) break;
    }

    // ============================== one traversal step per lane (pathKernel, wide radiance path) ==============================
    {
      PTX_MARK("TRAV_BEGIN");
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

      const bool deep = __any(sp + 4u >= kPStack);
      const uint32_t top = poolStackRead(stk, sp ? sp - 1u : 0u, deep);

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
      bool doPop = false;
      uint32_t push2 = 0u, push3 = 0u, nPush = 0u;
      PTX_MARK("ENTER");
      PTX_MARK("ENTER");
      if (isEnter) {
        // TLAS leaf: enter its first instance, keep the rest of the leaf (gpuFunctions.hpp:32-41)
        tlasLeaf = (nPrim > 1u) ? makeTlasLeafRef(first + 1u, nPrim - 1u) : 0u;
        curInst = first;
        inInst = true;
        o = xformRows(r0, r1, r2, wo, 1.0f);
        d = xformRows(r0, r1, r2, wd, 0.0f);  // not normalised: t stays world-space
        reciprocalDir(d, inv, zeroDir);
        if (STATS) nInstEntries++;
        newRef = __float_as_uint(r3.y);  // InstXform::wideRootRef
      }
      PTX_MARK("TRI");
      PTX_MARK("TRI");
      if (isTri) {
        // BLAS leaf, one triangle per step, in order, strict `<` (intersection.hpp:9-44, :112-122)
        TriRec tri;
        tri.v0 = r0; tri.e1 = r1; tri.e2 = r2;
        float u = 0.0f, v = 0.0f;
        const float dist = triangleTest(o, d, tri, u, v);
        if (dist < hit.t) { hit.t = dist; hit.u = u; hit.v = v; hit.tri = first; hit.inst = curInst; }
        if (STATS) nTriTests++;
        doPop = nPrim == 1u;
        newRef = doPop ? top : makeLeafRef(first + 1u, nPrim - 1u);
      }
      PTX_MARK("NODE");
      PTX_MARK("NODE");
      if (isNode) {
        // four-wide node (wide_bvh.hpp): nearest hit child next, the others deferred farthest first
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
      PTX_MARK("MERGE");
      PTX_MARK("MERGE");
      {
        // leaving an instance: the pop that reaches a TLAS-level entry (pathKernel, v13)
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
        const uint32_t v0 = nPush == 3u ? push3 : (nPush == 2u ? push2 : pushVal);
        const uint32_t v1 = nPush == 3u ? push2 : pushVal;
        poolStackWrite(stk, sp, v0, deep);
        if (__any(nPush >= 2u)) {
          poolStackWrite(stk, sp + 1u, v1, deep);
          poolStackWrite(stk, sp + 2u, pushVal, deep);
        }
        sp = sp + nPush - (doPop ? 1u : 0u);
        ref = newRef;
      }
      PTX_PCOUNT(14, (uint32_t)__popcll(__ballot(steps)));
      PTX_PSTAMP(4);
      PTX_MARK("TRAV_END");
    }
  }

#ifdef PTX_POOL_PROFILE
  if (lane == 0u) {
    for (int k = 0; k < 16; k++) atomicAdd(&gPoolProfile[k], pprof[k]);
    // [16] longest wave, [17] waves, [18] sum of (duration / 1024)^2, [19..28] waves by duration in tenths of 2^30 ticks
    const unsigned long long dur = __builtin_readcyclecounter() - pstart;
    atomicMax(&gPoolProfile[16], dur);
    atomicAdd(&gPoolProfile[17], 1ull);
    atomicAdd(&gPoolProfile[18], (dur >> 10) * (dur >> 10));
    const uint32_t wv = (blockIdx.x * kBlockThreads + threadIdx.x) >> 6;
    if (wv < 8192u) {
      unsigned long long* o = gPoolWaves + wv * 8u;
      o[0] = dur; o[1] = pprof[8]; o[2] = pprof[14]; o[3] = pprof[10]; o[4] = pprof[11]; o[5] = pprof[4]; o[6] = pstart; o[7] = pprof[9];
    }
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

template <bool STATS>
__global__ void __launch_bounds__(kBlockThreads, STATS ? 5 : PTX_POOL_MIN_WAVES) poolKernel(DeviceScene sc, FrameConstants fcArg, PoolLayout qArg,
                                                                                uint32_t* workCounter, void* __restrict__ outRaw,
                                                                                DeviceStats* stats, uint32_t* stackOverflow) {
  (void)fcArg; (void)qArg; (void)workCounter; (void)outRaw;  // read from the kernel-argument segment where they are needed (PoolKernelArgs)
  poolBody<STATS, false>(sc, stats, stackOverflow);
}
template <bool STATS>
__global__ void __launch_bounds__(kBlockThreads, STATS ? 5 : PTX_POOL_MIN_WAVES) pacedPoolKernel(DeviceScene sc, FrameConstants fcArg, PoolLayout qArg,
                                                                                     uint32_t* workCounter, void* __restrict__ outRaw,
                                                                                     DeviceStats* stats, uint32_t* stackOverflow) {
  (void)fcArg; (void)qArg; (void)workCounter; (void)outRaw;
  poolBody<STATS, true>(sc, stats, stackOverflow);
}

}  // namespace ptx
