// Kernels of the gfx950 backend.
//
// pathKernel is a persistent, wavefront-style state machine: the grid is sized to the chip (not to
// the image), every lane owns one work item at a time -- a pixel, or one RNG stream of a pixel
// (ptx_params.sub_streams) -- and pulls the next one from a global work counter when it is finished,
// and inside the main loop each lane advances its own ray by one step per iteration.  The lane's
// whole traversal state is the one-dword node reference it holds (device_scene.hpp):
//
//      internal node  : `-a` heat map: two slab tests on one 64-byte NodePair, order, push far child
//                       radiance    : up to four quantised child boxes of one 64-byte WideNode
//                                     (wide_bvh.hpp), sort, push up to three
//      BLAS leaf      : one Moeller-Trumbore test
//      TLAS leaf      : enter the next instance (ray -> object space); the instance is left, at no step
//                       of its own, by the pop that reaches a TLAS-level entry of the stack
//      kRefDone       : traversal finished, wait for the shading phase
//      kRefIdle       : no work left
//
// Every iteration a lane reads its stack top from LDS and fetches the one record its state needs
// (NodePair / WideNode, TriRec or InstXform: all live in one arena and are addressed from its base)
// -- only the 16-byte quarters its step uses, because the vector memory path charges per load
// instruction and active lane -- so an iteration pays one memory latency, not
// one per state; the state bodies then run from registers, predicated, and the iteration ends with
// the LDS write(s) of the slots above the stack top.  TLAS and BLAS nodes are the same records, so
// lanes walking the TLAS and lanes inside an instance share the node-step code.  Lanes whose ray has
// finished wait until a quarter of the wave (or PTX_SHADE_THRESHOLD lanes) is waiting; then the
// shading phase runs once for all of them: shade the hit / look up the environment, accumulate,
// write finished items, pull new ones, generate the next ray.  Because every lane always has a ray in
// flight, a wave is never held hostage by its slowest path, and an image with 100x cost differences
// between pixels (scene 1) load-balances through the counter.
//
// Registers follow the phases.  The shading phase is where the register pressure peaks, and the lanes that do not
// shade only carry their traversal state through it, so that state (ray, hit record, TLAS-leaf rest, instance:
// 16 dwords) is stored before the phase and read back after it; the bounce counter, touched in the phase only,
// stays in the same park area between phases; the work item's running sum lives in LDS; the frame constants are
// re-read from the kernel-argument segment by scalar loads in the phase instead of occupying SGPRs through the
// loop.  That is what lets the radiance kernel run at 6 waves per SIMD (80 VGPRs) with no scratch.
//
// `-a`: per-ray visit order, ray.t at every slab test and every rounding are those of the reference's
// nested traversal (intersection.hpp:99-233), so the counts are bit-exact against the CPU path.
// Radiance draws the same random numbers as the reference's GPU kernel and finds the same hits through
// the wide layout (same leaves, same triangle arithmetic; see wide_bvh.hpp for what may differ).
#pragma once

#include "../host/colors.hpp"
#include "device_scene.hpp"
#include "shade.hpp"
#include "trace.hpp"

namespace ptx {

constexpr uint32_t kBlockThreads = 256;  // 4 waves
#ifndef PTX_PARK
#define PTX_PARK 1   // traversal-only lane state waits in memory during the shading phase (0: stays in registers; then 5 waves/SIMD)
#endif
#ifndef PTX_SHADE_SHARE
#define PTX_SHADE_SHARE 4u   // ... or when one live lane in this many waits (the tail of the frame)
#endif
#ifndef PTX_MIN_WAVES
#define PTX_MIN_WAVES (PTX_PARK ? 6 : 5)  // waves per SIMD the register budget is cut for: 80 / 96 VGPRs
#endif
#ifndef PTX_OTHER_MIN_WAVES
#define PTX_OTHER_MIN_WAVES 5  // instrumented variants and the binary-layout radiance fallback (profiling builds may ask for 6)
#endif
#ifndef PTX_LDS_STACK
#define PTX_LDS_STACK (PTX_PARK ? 23 : 26)  // with the item's running sum (3 dwords per lane): 26 KB per block either way
#endif
constexpr uint32_t kLdsStack = PTX_LDS_STACK;  // stack entries per lane kept in LDS (1 KB per entry per block)
// Tunables (see DESIGN.md "Scheduling inside a wave").
#ifndef PTX_SHADE_THRESHOLD
#define PTX_SHADE_THRESHOLD 16  // waiting lanes that trigger the shading phase
#endif
#ifndef PTX_ENTER_THRESHOLD
#define PTX_ENTER_THRESHOLD 4   // lanes at a TLAS leaf before the (rare, long) instance-entry body is run
#endif
#ifndef PTX_RARE_PERIOD
#define PTX_RARE_PERIOD 1      // run the instance entry / exit bodies on every n-th iteration only
#endif
#ifndef PTX_INNER_ITERS
#define PTX_INNER_ITERS 1       // traversal iterations between two checks for the shading phase
#endif

// Optional markers in the generated assembly (tools/count_isa.py); no code is emitted for them.
#ifdef PTX_ISA_MARKERS
#define PTX_MARK(name) asm volatile("; PTXMARK " name)
#else
#define PTX_MARK(name) do {} while (0)
#endif

// Wave-level execution profile of the main loop (profiling builds only: -DPTX_WAVE_PROFILE; filled by the
// STATS instantiations, read with ptx_debug_wave_profile).  [2k] = iterations in which body k ran,
// [2k+1] = lanes that were active in it.
#ifdef PTX_WAVE_PROFILE
__device__ unsigned long long gWaveProfile[32];
#endif
#if defined(PTX_WAVE_PROFILE) && !defined(PTX_TIME_PROFILE)
#define PTX_PROF(k, lanes) do { if (STATS) { prof[2 * (k)] += 1u; prof[2 * (k) + 1] += (uint32_t)__popcll(__ballot(lanes)); } } while (0)
#else
#define PTX_PROF(k, lanes) do {} while (0)
#endif
// Wall-clock profile of the loop's regions (profiling builds only: -DPTX_TIME_PROFILE, with
// -DPTX_WAVE_PROFILE): shader-clock ticks per region summed over all waves into gWaveProfile[16 + k].
#if defined(PTX_WAVE_PROFILE) && defined(PTX_TIME_PROFILE)
#define PTX_STAMP(k) do { if (STATS) { const unsigned long long now_ = __builtin_readcyclecounter(); tprof[k] += now_ - tlast; tlast = now_; } } while (0)
#define PTX_WAIT_ALL() do { if (STATS) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define PTX_STAMP(k) do {} while (0)
#define PTX_WAIT_ALL() do {} while (0)
#endif

#ifdef PTX_VISIT_HIST
__device__ unsigned int gVisitHist[1u << 20];  // node steps per pair index (profiling builds only)
#endif

typedef uint32_t __attribute__((address_space(1))) * GlobalWords;  // a pointer the compiler knows to be global memory
constexpr uint32_t kParkRegs = 16u;                      // dwords of traversal state parked during the shading phase
constexpr uint32_t kParkSlots = PTX_PARK ? kParkRegs + 1u : 0u;  // entries of the global park area: those, and the bounce counter
constexpr int kModeCounts = 0;    // `-a`: gpuRender.cu:146-150
constexpr int kModeRadiance = 1;  // gpuRender.cu:132-169

// Per-lane traversal stack: entries [0, kLdsStack) live in LDS as sStack[entry][thread] (consecutive
// lanes -> consecutive banks); deeper entries go to a per-launch global overflow buffer laid out
// [entry][global thread] (coalesced; touched only by rays that really go that deep).  The main loop
// touches the stack exactly twice per iteration for every lane -- an early read of the top entry,
// overlapped with the global fetch, and one write of the slot above the top (a write that is not
// followed by sp++ leaves garbage above the top, which is harmless) -- so the common path has no
// branches; a wave-uniform test diverts to the LDS-or-overflow path only while some lane of the wave
// is near the LDS capacity.
struct StackMem {
  uint32_t* lds;       // &sStack[threadIdx.x]
  uint32_t* overflow;  // &overflowBase[global thread]
  uint32_t stride;     // total threads of the launch
};
template <uint32_t CAP = kLdsStack>  // entries in LDS
__device__ __forceinline__ uint32_t stackRead(const StackMem& m, uint32_t slot, bool deep) {
  uint32_t v;
  if (!deep) {
    v = m.lds[slot * kBlockThreads];
  } else {
    // both loads, then a value select: a pointer select would become a (slow, doubly counted) flat load
    const uint32_t a = m.lds[(slot < CAP ? slot : CAP - 1u) * kBlockThreads];
    const uint32_t b = m.overflow[(size_t)(slot < CAP ? 0u : slot - CAP) * m.stride];
    v = slot < CAP ? a : b;
  }
  return v;
}
template <uint32_t CAP = kLdsStack>
__device__ __forceinline__ void stackWrite(const StackMem& m, uint32_t slot, uint32_t val, bool deep) {
  if (!deep) {
    m.lds[slot * kBlockThreads] = val;
  } else {
    if (slot < CAP) m.lds[slot * kBlockThreads] = val;
    else m.overflow[(size_t)(slot - CAP) * m.stride] = val;
  }
}

// Work item -> pixel.  The image is cut into bands of 8 rows; inside a band items run down a column
// first, so 64 consecutive items (one wave's first fetch) form an 8x8 block of adjacent pixels for
// any image size, with no padding items.
__device__ __forceinline__ void workPixel(const FrameConstants& fc, uint32_t w, uint32_t& x, uint32_t& y) {
  const uint32_t bandItems = fc.width * 8u;
  const uint32_t band = __umulhi(w << 8, fc.bandMul) >> fc.bandShift;  // w / bandItems, w < 2^24 (see divByWidth)
  const uint32_t r = w - band * bandItems;
  const uint32_t rows = min(8u, fc.height - band * 8u);
  const uint32_t cx = r / rows;
  x = cx;
  y = band * 8u + (r - cx * rows);
}

// n / fc.width for n < 2^24 (a pixel index) without the compiler's division sequence, whose hoisted per-lane
// reciprocal would occupy a VGPR for the whole kernel: multiplier and shift come from the host (makeFrame),
// exact for every 24-bit numerator (M = ceil(2^(24+s) / width), s = ceil(log2 width)).
__device__ __forceinline__ uint32_t divByWidth(const FrameConstants& fc, uint32_t n) {
  return __umulhi(n << 8, fc.widthMul) >> fc.widthShift;
}

#ifndef PTX_RELOAD_FRAME
#define PTX_RELOAD_FRAME 1  // the frame constants are read from the kernel-argument segment at every shading phase
#endif
// The kernel arguments as the hardware lays them out (HIP: in order, each at its natural alignment).
struct PathKernelArgs {
  DeviceScene sc;
  FrameConstants fc;
};
// A copy of a kernel argument made by scalar loads from the kernel-argument segment where it is called, instead of
// SGPRs held from the kernel's entry: the frame constants (50 dwords) are used in the shading phase only, and holding
// them through the traversal loop made the compiler keep 42 SGPRs in the lanes of a VGPR (v_readlane / v_writelane
// around every use: 126 of them in the hit shading alone).
template <typename T>
__device__ __forceinline__ T loadKernelArgument(uint32_t byteOffset) {
  typedef const uint32_t __attribute__((address_space(4))) * KernargWords;
  KernargWords k = (KernargWords)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(k));  // not hoisted out of the loop
  T out;
  uint32_t* o = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
  for (uint32_t i = 0; i < sizeof(T) / 4u; i++) o[i] = k[byteOffset / 4u + i];
  return out;
}

// The register budget: 6 waves/SIMD (80 VGPRs) for the shipped radiance kernel and the heat map; the instrumented
// variants and the binary-layout fallback of the radiance path carry more state and keep 5 (96 VGPRs): no scratch anywhere.
template <int MODE, bool STATS, bool WIDE>
__global__ void __launch_bounds__(kBlockThreads, (WIDE && !STATS) || MODE == kModeCounts ? PTX_MIN_WAVES : PTX_OTHER_MIN_WAVES) pathKernel(DeviceScene sc, FrameConstants fcArg, uint32_t* workCounter,
                                                            void* __restrict__ outRaw, DeviceStats* stats,
                                                            uint32_t* stackOverflow) {
  constexpr bool kOneRay = MODE == kModeCounts;  // heat map: one primary ray per pixel
  __shared__ uint32_t sStack[kLdsStack * kBlockThreads];
#if PTX_PARK
  // the work item's running sum is touched when a path ends and when the item is written: LDS, not three registers
  __shared__ float sSum[3 * kBlockThreads];
  float* const sumLds = sSum + threadIdx.x;
#endif
  StackMem stk;
  stk.lds = sStack + threadIdx.x;
  stk.stride = gridDim.x * kBlockThreads;
  // the first kParkSlots entries of the area are the park slots of the shading phase (PTX_PARK)
  uint32_t* const parkGlobal = stackOverflow + (blockIdx.x * kBlockThreads + threadIdx.x);
  stk.overflow = parkGlobal + (size_t)kParkSlots * stk.stride;
  (void)parkGlobal;
  uint32_t sp = 1;

#if !PTX_RELOAD_FRAME
  const FrameConstants& fc = fcArg;
  const uint32_t nPixels = fc.width * fc.height;
  const uint32_t totalWork = nPixels * fc.subStreams;  // checked on the host to fit 32 bits
  const vec3 camPos(fc.camPos[0], fc.camPos[1], fc.camPos[2]);
#else
  (void)fcArg;
#endif

  // ---- lane state ----
  uint32_t ref = kRefDone;   // what the lane does next (see the table above)
  bool havePixel = false;    // the lane owns a work item; at the shading phase: its ray's result is waiting
  bool zeroDir = false;      // a component of the current ray's direction is exactly 0 (0*inf possible)
  bool wZeroDir = false;     // the same for the world-space ray (restored when an instance is left)
  uint32_t pix = 0;     // pixel index (bits 0..23) | RNG stream of this item (bits 24..31)
  uint32_t sample = 0;  // samples (rays, for the one-ray modes) this item still has to do; 0 = pull the next item
  uint32_t bounce = 0, rng = 0, count = 0;
  uint32_t tlasLeaf = 0, curInst = 0;
  bool inInst = false;  // the current ray is in an instance's object space
  vec3 sum(0.0f), col(1.0f);
  vec3 wo(0.0f), wd(0.0f, 0.0f, -1.0f), winv(0.0f);  // world-space ray
  vec3 o(0.0f), d(0.0f, 0.0f, -1.0f), inv(0.0f);     // current ray (world or object space)
  Hit hit;
  hit.t = FLT_MAX; hit.u = 0.0f; hit.v = 0.0f; hit.tri = 0xFFFFFFFFu; hit.inst = 0xFFFFFFFFu;

  uint32_t nRays = 0, nNodeSteps = 0, nTriTests = 0, nInstEntries = 0, nHits = 0, nMisses = 0;
  unsigned long long nBoxHits = 0;

#ifdef PTX_WAVE_PROFILE
  unsigned long long prof[16] = {};
#ifdef PTX_TIME_PROFILE
  unsigned long long tprof[8] = {};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif
#endif
  uint32_t loopCount = 0;
  for (;;) {
    loopCount++;
    // =========================== shading phase (postponed, wave-wide) ===========================
    const bool waiting = ref == kRefDone;
    const uint32_t nWaiting = (uint32_t)__popcll(__ballot(waiting));
    const uint32_t nLive = 64u - (uint32_t)__popcll(__ballot(ref == kRefIdle));
    if (nLive == 0u) break;
    PTX_PROF(0, ref != kRefIdle);
    // one ray per pixel (`-a`): the phase is short and the frame is all tail, so it pays to wait for half of the
    // live lanes (1.65 -> 1.29 ms at 1500x800 on scene 1); radiance: 16 lanes or a quarter (12: +3 %, 32 / half: +3.5 %)
    constexpr uint32_t kShadeLanes = kOneRay ? 32u : PTX_SHADE_THRESHOLD, kShadeShare = kOneRay ? 2u : PTX_SHADE_SHARE;
    if (nWaiting >= kShadeLanes || nWaiting * kShadeShare >= nLive) {
      PTX_PROF(1, waiting);
#if PTX_PARK
      // The shading phase is where the register pressure peaks, and the lanes that do not shade only carry their
      // traversal state through it: every lane stores it here and picks it up after the phase (a lane that shaded
      // takes its new segment's values instead), so the registers are free for the shading arithmetic.  Where to:
      // the LDS stack slots above the lane's top when every lane of the wave has kParkRegs of them free
      // (wave-uniform; shallow trees), else the lane's column of the global park area.  Stores and loads are
      // unconditional and the hand-over is a select per value: a conditional version makes the compiler keep
      // the registers (profiles/r02/park_6waves_ab.txt).
      const bool parkLds = !__any(sp + kParkRegs > kLdsStack);
      if (MODE == kModeRadiance) {
        const uint32_t pv[kParkRegs] = {__float_as_uint(o.x),   __float_as_uint(o.y),   __float_as_uint(o.z),   __float_as_uint(d.x),
                                        __float_as_uint(d.y),   __float_as_uint(d.z),   __float_as_uint(inv.x), __float_as_uint(inv.y),
                                        __float_as_uint(inv.z), __float_as_uint(hit.t), __float_as_uint(hit.u), __float_as_uint(hit.v),
                                        hit.tri,                hit.inst,               tlasLeaf,               curInst};
        if (parkLds) {
#pragma unroll
          for (uint32_t i = 0; i < kParkRegs; i++) stk.lds[(sp + i) * kBlockThreads] = pv[i];
        } else {
          GlobalWords pg = (GlobalWords)parkGlobal;  // global, not flat: the asm hides where the pointer came from
          asm volatile("" : "+v"(pg));            // the addresses are computed here, not kept in registers across the loop
#pragma unroll
          for (uint32_t i = 0; i < kParkRegs; i++) { *pg = pv[i]; pg += stk.stride; }
        }
      }
#endif
      if (waiting) {
        bool pathEnded = true;
#if PTX_RELOAD_FRAME
        const FrameConstants fc = loadKernelArgument<FrameConstants>((uint32_t)offsetof(PathKernelArgs, fc));
        const uint32_t nPixels = fc.width * fc.height;
        const uint32_t totalWork = nPixels * fc.subStreams;  // checked on the host to fit 32 bits
        const vec3 camPos(fc.camPos[0], fc.camPos[1], fc.camPos[2]);
#endif
#if PTX_PARK
        // The bounce counter is touched here and nowhere else: it lives in the park area between shading phases
        // (read now, needed a thousand instructions later; written back at the launch of the next segment).
        GlobalWords bounceSlot = (GlobalWords)parkGlobal;
        asm volatile("" : "+v"(bounceSlot));
        bounceSlot += (size_t)kParkRegs * stk.stride;
        if (MODE == kModeRadiance && havePixel) bounce = *bounceSlot;
#endif
        if (havePixel) {
          if (STATS) {
            nRays++;
            nBoxHits += count;
          }
          if (kOneRay) sample = 0u;
          if (MODE == kModeCounts) {
            reinterpret_cast<uint32_t*>(outRaw)[pix] = count;
          } else if (hit.t < FLT_MAX) {
            PTX_MARK("S_HIT");
            if (STATS) nHits++;
            shadeHit(sc, hit, wo, wd, rng, col);  // writes the next ray into wo / wd
            bounce++;
            pathEnded = bounce >= fc.maxBounces;  // gpuRender.cu:29: the loop ends, col is returned
          } else {
            PTX_MARK("S_MISS");
            if (STATS) nMisses++;
            col *= environmentLookup(sc, fc, wd);
          }
          PTX_MARK("S_AFTER");
          if (MODE == kModeRadiance && pathEnded) {
#if PTX_PARK
            sumLds[0] += col.x; sumLds[kBlockThreads] += col.y; sumLds[2 * kBlockThreads] += col.z;
#else
            sum += col;
#endif
            sample--;
          }
        }
        bool live = true;
        if (pathEnded) {
          PTX_MARK("S_PULL");
          // ---- item finished? write it and pull the next one ----
          while (sample == 0u && live) {
            if (MODE == kModeRadiance && havePixel) {
              const uint32_t p = pix & 0x00FFFFFFu, k = pix >> 24;
              // plane k of the accumulation buffer: (sum, samples) of stream k
              const uint32_t n = fc.sppBase + (k < fc.sppExtra ? 1u : 0u);
#if PTX_PARK
              sum = vec3(sumLds[0], sumLds[kBlockThreads], sumLds[2 * kBlockThreads]);
#endif
              reinterpret_cast<float4*>(outRaw)[(size_t)k * nPixels + p] = make_float4(sum.x, sum.y, sum.z, (float)n);
            }
            const uint32_t w = atomicAdd(workCounter, 1u);  // aggregated per wave by the compiler
            if (w >= totalWork) {
              live = false;
              havePixel = false;
              break;
            }
            // item -> (stream, pixel slot): all pixels of stream 0, then of stream 1, ...; slots walk the image in 8x8
            // blocks.  The streams that hold one sample more (k < sppExtra) come first, so the frame ends with its
            // shortest items (13 spp in 7 streams: 47.3 ms against 48.7 ms with a pixel's streams side by side;
            // profiles/r02/stream_major_ab.txt).  (Handing the pixels out most expensive first -- priced by the
            // heat-map kernel, counting sort over 256 cost classes -- was built and measured: 474 -> 470 ms with the
            // reference's seeding, 352.6 -> 351.7 ms with 25 streams, less than the pre-pass costs: the tail is the
            // length of the last items, not late starts of expensive pixels; profiles/r02/cost_order_ab.txt.)
            uint32_t slot = w, k = 0u;
            if (fc.subStreams > 1u) {
              k = (uint32_t)((double)w * fc.invPixels);  // w < 2^32: exact up to one unit either way
              slot = w - k * nPixels;
              if ((int32_t)slot < 0) { k--; slot += nPixels; }
              else if (slot >= nPixels) { k++; slot -= nPixels; }
            }
            uint32_t x, y;
            workPixel(fc, slot, x, y);
            const uint32_t p = y * fc.width + x;
            pix = p | (k << 24);
            sum = vec3(0.0f);
#if PTX_PARK
            sumLds[0] = 0.0f; sumLds[kBlockThreads] = 0.0f; sumLds[2 * kBlockThreads] = 0.0f;
#endif
            rng = fc.seedBase + p + k * 0x9E3779B9u;  // gpuRender.cu:130 (+ stream offset)
            havePixel = true;
            sample = kOneRay ? 1u : fc.sppBase + (k < fc.sppExtra ? 1u : 0u);
            if (kOneRay) {
              if (MODE == kModeCounts && fc.partCount > 1u) {
                const uint32_t tile = (y >> 3) * fc.tilesX + (x >> 3);
                if ((tile % fc.partCount) != fc.partIndex) {
                  reinterpret_cast<uint32_t*>(outRaw)[p] = 0u;
                  sample = 0u;  // not this part's tile: finished
                }
              }
            } else if (fc.maxBounces == 0u) {
              sum = vec3((float)sample);  // no segment is traced: every path returns white (gpuRender.cu:27-29)
#if PTX_PARK
              sumLds[0] = sum.x; sumLds[kBlockThreads] = sum.y; sumLds[2 * kBlockThreads] = sum.z;
#endif
              sample = 0u;
            }
          }
          if (live) {
            PTX_MARK("S_CAMERA");
            // ---- next camera ray: gpuRender.cu:132-143 ----
            const uint32_t p = pix & 0x00FFFFFFu;
            const uint32_t py = divByWidth(fc, p), px = p - py * fc.width;
            float fx = (float)px, fy = (float)py;
            if (MODE == kModeRadiance && fc.jitter) {
              const vec2 xi = pt::getRandomVec2(rng);
              fx += 0.5f * (2.0f * xi.x - 1.0f);
              fy += 0.5f * (2.0f * xi.y - 1.0f);
            }
            wo = camPos;
            wd = primaryDirection(fc, fx, fy);
            col = vec3(1.0f);
            bounce = 0u;
          }
        }
        if (live) {
          PTX_MARK("S_LAUNCH");
          // ---- launch the segment: Scene::intersect starts at the TLAS root ----
          reciprocalDir(wd, winv, wZeroDir);
          zeroDir = wZeroDir;
          if (!(PTX_PARK && MODE == kModeRadiance)) {
            o = wo; d = wd; inv = winv;
            hit.t = FLT_MAX; hit.u = 0.0f; hit.v = 0.0f; hit.tri = 0xFFFFFFFFu; hit.inst = 0xFFFFFFFFu;
            tlasLeaf = 0u;
          }
          count = 0u;
          inInst = false;
#if PTX_PARK
          if (MODE == kModeRadiance) *bounceSlot = bounce;
#endif
          stk.lds[0] = kRefDone;  // slot 0: bottom-of-stack marker
          sp = 1u;
          ref = WIDE ? sc.tlasRootRefWide : sc.tlasRootRef;
        } else {
          ref = kRefIdle;
        }
      }
#if PTX_PARK
      if (MODE == kModeRadiance) {
        // a lane that shaded starts its new segment (world space, no hit yet); the others get their state back
        uint32_t pv[kParkRegs];
        if (parkLds) {
#pragma unroll
          for (uint32_t i = 0; i < kParkRegs; i++) pv[i] = stk.lds[(sp + i) * kBlockThreads];
        } else {
          GlobalWords pg = (GlobalWords)parkGlobal;
          asm volatile("" : "+v"(pg));
#pragma unroll
          for (uint32_t i = 0; i < kParkRegs; i++) { pv[i] = *pg; pg += stk.stride; }
        }
        o.x = waiting ? wo.x : __uint_as_float(pv[0]); o.y = waiting ? wo.y : __uint_as_float(pv[1]); o.z = waiting ? wo.z : __uint_as_float(pv[2]);
        d.x = waiting ? wd.x : __uint_as_float(pv[3]); d.y = waiting ? wd.y : __uint_as_float(pv[4]); d.z = waiting ? wd.z : __uint_as_float(pv[5]);
        inv.x = waiting ? winv.x : __uint_as_float(pv[6]); inv.y = waiting ? winv.y : __uint_as_float(pv[7]); inv.z = waiting ? winv.z : __uint_as_float(pv[8]);
        hit.t = waiting ? FLT_MAX : __uint_as_float(pv[9]);
        hit.u = waiting ? 0.0f : __uint_as_float(pv[10]);
        hit.v = waiting ? 0.0f : __uint_as_float(pv[11]);
        hit.tri = waiting ? 0xFFFFFFFFu : pv[12];
        hit.inst = waiting ? 0xFFFFFFFFu : pv[13];
        tlasLeaf = waiting ? 0u : pv[14];
        curInst = waiting ? 0u : pv[15];
      }
#endif
    }

    PTX_STAMP(0);  // shading phase (and the loop-top checks)
    // ============================== traversal steps, one per lane each ===============================
#pragma unroll 1
    for (int it = 0; it < PTX_INNER_ITERS; it++) {
      PTX_MARK("TRAV_BEGIN");
      const bool isNode = (int32_t)ref >= 0;
      const bool isSpecial = ref >= kRefIdle;  // idle / done
      const bool isTri = (ref >> 30) == 2u;
      bool isEnter = (ref >> 30) == 3u && !isSpecial;
      if (PTX_ENTER_THRESHOLD > 1 && MODE == kModeRadiance) {
        // the instance-entry body is long and rarely needed: run it when a few lanes want it, or when
        // nothing else is left to do in this wave
        const uint32_t nEnter = (uint32_t)__popcll(__ballot(isEnter));
        if (nEnter < PTX_ENTER_THRESHOLD && __any(isNode || isTri)) isEnter = false;
      }
      if (PTX_RARE_PERIOD > 1) {
        // instance entry / exit bodies run on every PTX_RARE_PERIOD-th iteration only (a lane waits at
        // most PTX_RARE_PERIOD - 1 iterations for them), unless no lane has anything else to do
        if ((loopCount % PTX_RARE_PERIOD) != 0u && __any(isNode || isTri)) isEnter = false;
      }
      const bool steps = isNode || isTri || isEnter;
      const uint32_t first = ref & (isTri ? 0x007FFFFFu : 0x00FFFFFFu);
      const uint32_t nPrim = (isTri ? ((ref >> 23) & 0x7Fu) : ((ref >> 24) & 0x3Fu)) + 1u;

      // wave-uniform; false almost always (a wide step writes the three slots above the top)
      const bool deep = __any(sp + (WIDE ? 4u : 2u) >= kLdsStack);
      const uint32_t top = stackRead(stk, sp ? sp - 1u : 0u, deep);

      // Record fetch.  The vector memory path charges per load instruction and active lane (tools/ubench_gather2.hip:
      // ~45 cycles per divergent 64-lane load whatever its width); it bound the binary layout and still runs at
      // 0.6 of its rate on the wide one, so every lane fetches only what its step uses: nothing for lanes that
      // wait, three quarters for a triangle.  (Keeping the most visited NodePairs in LDS was tried and lost: the
      // hottest pairs are the ones many lanes of a wave read together, which the memory path already serves as
      // one access.)
      const bool fetch3 = isNode || isTri || isEnter;
      const bool fetch4 = isNode || isEnter;
      const uint32_t off16 = isNode ? (WIDE ? sc.wide16 : sc.pairs16) + ref * 4u : (isTri ? sc.tris16 + first * 3u : sc.inst16 + first * 4u);
      const float4* rp = sc.arena + off16;
      // lanes that fetch nothing never look at r0..r3: "any value" without the 16 moves of a zero fill
      float4 r0, r1, r2, r3;
      asm volatile("" : "=v"(r0.x), "=v"(r0.y), "=v"(r0.z), "=v"(r0.w), "=v"(r1.x), "=v"(r1.y), "=v"(r1.z), "=v"(r1.w));
      asm volatile("" : "=v"(r2.x), "=v"(r2.y), "=v"(r2.z), "=v"(r2.w), "=v"(r3.x), "=v"(r3.y), "=v"(r3.z), "=v"(r3.w));
      if (fetch3) { r0 = rp[0]; r1 = rp[1]; r2 = rp[2]; }
      if (fetch4) r3 = rp[3];

      uint32_t newRef = ref;
      uint32_t pushVal = 0u;
      bool doPush = false, doPop = false;
      uint32_t push2 = 0u, push3 = 0u, nPush = 0u;  // WIDE: a node step defers up to three children
      PTX_STAMP(1);  // decode, addresses, loads issued
      PTX_WAIT_ALL();
      PTX_STAMP(2);  // waiting for the record and the stack top

      PTX_PROF(2, steps);
      if (__any(isEnter)) PTX_PROF(4, isEnter);
      if (__any(isTri)) PTX_PROF(5, isTri);
      if (__any(isNode)) PTX_PROF(6, isNode);
      if (__any(isNode) && __any(zeroDir)) PTX_PROF(7, isNode);
      PTX_STAMP(3);
      PTX_MARK("ENTER");
      if (isEnter) {
        // TLAS leaf: enter its first instance, keep the rest of the leaf (gpuFunctions.hpp:32-41).
        tlasLeaf = (nPrim > 1u) ? makeTlasLeafRef(first + 1u, nPrim - 1u) : 0u;
        curInst = first;
        inInst = true;
        o = xformRows(r0, r1, r2, wo, 1.0f);
        d = xformRows(r0, r1, r2, wd, 0.0f);  // not normalised: t stays world-space
        reciprocalDir(d, inv, zeroDir);
        if (STATS) nInstEntries++;
        newRef = __float_as_uint(WIDE ? r3.y : r3.x);  // InstXform::rootRef / wideRootRef
      }
      PTX_STAMP(4);
      PTX_MARK("TRI");
      if (isTri) {
        // BLAS leaf, one triangle per step, in order, strict `<` (intersection.hpp:9-44, :112-122).
        TriRec tri;
        tri.v0 = r0; tri.e1 = r1; tri.e2 = r2;
        float u = 0.0f, v = 0.0f;
        const float dist = triangleTest(o, d, tri, u, v);
        if (dist < hit.t) {
          // the instance is recorded with the hit: intersectTLAS keeps the instance whose BLAS lowered t
          // (GPUMesh::intersect returns ray.t = transformedRay.t, intersection.hpp:184-189)
          hit.t = dist; hit.u = u; hit.v = v; hit.tri = first; hit.inst = curInst;
        }
        if (STATS) nTriTests++;
        doPop = nPrim == 1u;
        newRef = doPop ? top : makeLeafRef(first + 1u, nPrim - 1u);
      }
      PTX_STAMP(5);
      PTX_MARK("NODE");
      if (WIDE && isNode) {
        // Four-wide node of the radiance path (wide_bvh.hpp): nearest hit child next, the others are
        // deferred farthest first, so that they pop nearest first.
        uint32_t sr[4];
        uint32_t nHit;
        if (__any(zeroDir)) nHit = wideStep<true>(o, inv, hit.t, r0, r1, r2, r3, sr);
        else nHit = wideStep<false>(o, inv, hit.t, r0, r1, r2, r3, sr);
        if (STATS) nNodeSteps++;
        count += nHit;
        doPop = nHit == 0u;
        newRef = doPop ? top : sr[0];
        nPush = doPop ? 0u : nHit - 1u;
        // sr[i] (i >= 1) goes to slot sp + nHit - 1 - i; unused ones to the spare slot sp + 3
        pushVal = sr[1]; push2 = sr[2]; push3 = sr[3];
      }
      if (!WIDE && isNode) {
        // Internal node (intersection.hpp:133-161 / :201-230): one 64-byte record holds both children.
        float t1, t2;
        bool h1, h2;
        if (__any(zeroDir)) {
          h1 = slabHit<true>(o, inv, hit.t, r0, r1, t1);
          h2 = slabHit<true>(o, inv, hit.t, r2, r3, t2);
        } else {
          h1 = slabHit<false>(o, inv, hit.t, r0, r1, t1);
          h2 = slabHit<false>(o, inv, hit.t, r2, r3, t2);
        }
        const uint32_t rl = __float_as_uint(r0.w), rr = __float_as_uint(r2.w);
        if (STATS) nNodeSteps++;
#ifdef PTX_VISIT_HIST
        if (STATS && ref < (1u << 20)) atomicAdd(&gVisitHist[ref], 1u);
#endif
        // The reference compares d = hit ? tmin : FLT_MAX of the two children and visits the nearer one
        // first, ties keeping the left child (intersection.hpp:139-152); a hit's tmin is < ray.t <=
        // FLT_MAX, so "d1 > d2" is exactly: right hit, and left missed or farther.
#ifdef PTX_VALUE_ORDER
        const bool swap = (h1 ? t1 : FLT_MAX) > (h2 ? t2 : FLT_MAX);
#else
        const bool swap = h2 && (!h1 || t1 > t2);
#endif
        const uint32_t rn = swap ? rr : rl, rf = swap ? rl : rr;
        const bool hitNear = h1 || h2, hitFar = h1 && h2;
        count += (h1 ? 1u : 0u) + (h2 ? 1u : 0u);  // counted when pushed, not re-tested when popped
        pushVal = rf;
        doPush = hitFar;
        doPop = !hitNear;
        newRef = hitNear ? rn : top;
      }

      PTX_STAMP(6);
      PTX_MARK("MERGE");
      {
        // Leaving an instance costs no step of its own and no marker on the stack: a pop that would hand out
        // an entry of the TLAS level (a TLAS node or leaf, or the bottom marker) while the ray is in object
        // space means the BLAS is finished (the `return` of intersectBVH, intersection.hpp:99-163): the ray
        // goes back to world space and carries on with the rest of its TLAS leaf, or with that entry.
        const bool leaving = doPop && inInst && (top < sc.tlasNodeEnd[WIDE ? 1 : 0] || top >= kTlasLeafFlag);
        if (__any(leaving)) {
          PTX_PROF(3, leaving);
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
      if (WIDE) {
        if (steps) {
          // the one deferred reference of the other bodies is "child 1 of 1"
          if (!isNode) nPush = doPush ? 1u : 0u;
          // slots sp, sp+1, sp+2 in stack order: farthest deferred child lowest, nearest on top (slots at
          // or above the new top receive values that are never read)
          const uint32_t v0 = nPush == 3u ? push3 : (nPush == 2u ? push2 : pushVal);
          const uint32_t v1 = nPush == 3u ? push2 : pushVal;
          stackWrite(stk, sp, v0, deep);
          if (__any(nPush >= 2u)) {  // wave-uniform
            stackWrite(stk, sp + 1u, v1, deep);
            stackWrite(stk, sp + 2u, pushVal, deep);
          }
          sp = sp + nPush - (doPop ? 1u : 0u);
          ref = newRef;
        }
      } else if (steps) {
        stackWrite(stk, sp, pushVal, deep);
        sp = sp + (doPush ? 1u : 0u) - (doPop ? 1u : 0u);
        ref = newRef;
      }
      PTX_STAMP(7);
      PTX_MARK("TRAV_END");
    }
  }

#ifdef PTX_WAVE_PROFILE
  if (STATS && (threadIdx.x & 63u) == 0u)
    for (int k = 0; k < 16; k++) atomicAdd(&gWaveProfile[k], prof[k]);
#ifdef PTX_TIME_PROFILE
  if (STATS && (threadIdx.x & 63u) == 0u)
    for (int k = 0; k < 8; k++) atomicAdd(&gWaveProfile[16 + k], tprof[k]);
  if (STATS && (threadIdx.x & 63u) == 0u) atomicAdd(&gWaveProfile[0], (unsigned long long)loopCount);
#endif
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

// Adds the per-stream planes of a sub-stream launch in stream order; writes the accumulation buffer
// (sum, samples) or, with `resolve`, the finished pixel.
__global__ void __launch_bounds__(256) combineKernel(const float4* __restrict__ planes, uint32_t nStreams, uint32_t nPixels,
                                                     void* __restrict__ out, bool resolve) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nPixels) return;
  float4 a = planes[i];
  for (uint32_t k = 1; k < nStreams; k++) {
    const float4 b = planes[(size_t)k * nPixels + i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  if (resolve) {
    const vec3 c = resolvePixel(vec3(a.x, a.y, a.z), a.w);
    float* o = reinterpret_cast<float*>(out);
    o[3ull * i + 0] = c.x;
    o[3ull * i + 1] = c.y;
    o[3ull * i + 2] = c.z;
  } else {
    reinterpret_cast<float4*>(out)[i] = a;
  }
}

// Tile-partitioned heat map (ptx_render_counts_multi): every pixel is non-zero in at most one part, so the
// integer sum reassembles the frame exactly.
__global__ void __launch_bounds__(256) sumCountsKernel(const uint32_t* __restrict__ parts, uint32_t nParts, uint32_t nPixels,
                                                       uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nPixels) return;
  uint32_t c = parts[i];
  for (uint32_t k = 1; k < nParts; k++) c += parts[(size_t)k * nPixels + i];
  out[i] = c;
}

// ---- tight instance bounds (ptx_create; tight_bounds.hpp) ----
// World-space bounds of the transformed vertices of every instance: block b takes instances 4b .. 4b+3, and while
// they share a geometry (scene 1: always) every triangle is loaded once and sent through the four model matrices
// (60 GB instead of 240 GB of reads for 10,000 instances of a 505,848-triangle mesh).  Minimum and maximum do not
// depend on the order of the reduction, so the result is deterministic.
struct InstanceModel {  // rows of the model matrix (object -> world) and the instance's triangle range
  float m[12];
  uint32_t triOffset, triCount;
  uint32_t pad[2];
};
constexpr uint32_t kBoundsGroup = 4u;

__global__ void __launch_bounds__(256) instanceBoundsKernel(const float* __restrict__ vertices /* 9 floats per triangle */,
                                                            const InstanceModel* __restrict__ models, uint32_t nInstances,
                                                            float* __restrict__ out /* 6 floats per instance: min, max */) {
  __shared__ float part[4][kBoundsGroup][6];
  const uint32_t base = blockIdx.x * kBoundsGroup;
  uint32_t first = 0;
  while (first < kBoundsGroup && base + first < nInstances) {
    // a run of instances with the same triangle range
    const uint32_t off = models[base + first].triOffset, cnt = models[base + first].triCount;
    uint32_t run = 1;
    while (first + run < kBoundsGroup && base + first + run < nInstances && models[base + first + run].triOffset == off &&
           models[base + first + run].triCount == cnt)
      run++;
    float lo[kBoundsGroup][3], hi[kBoundsGroup][3];
    float M[kBoundsGroup][12];
#pragma unroll
    for (uint32_t r = 0; r < kBoundsGroup; r++) {
      for (int a = 0; a < 3; a++) { lo[r][a] = FLT_MAX; hi[r][a] = -FLT_MAX; }
      const uint32_t src = base + first + (r < run ? r : 0u);
      for (int k = 0; k < 12; k++) M[r][k] = models[src].m[k];
    }
    const float* v = vertices + (size_t)off * 9u;
    for (uint32_t t = threadIdx.x; t < cnt; t += 256u) {
      float p[9];
      for (int k = 0; k < 9; k++) p[k] = v[(size_t)t * 9u + k];
#pragma unroll
      for (uint32_t r = 0; r < kBoundsGroup; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float x = p[3 * c], y = p[3 * c + 1], z = p[3 * c + 2];
#pragma unroll
          for (int a = 0; a < 3; a++) {
            const float w = M[r][4 * a] * x + M[r][4 * a + 1] * y + M[r][4 * a + 2] * z + M[r][4 * a + 3];
            lo[r][a] = fminf(lo[r][a], w);
            hi[r][a] = fmaxf(hi[r][a], w);
          }
        }
      }
    }
#pragma unroll
    for (uint32_t r = 0; r < kBoundsGroup; r++)
      for (int a = 0; a < 3; a++) {
        float l = lo[r][a], h = hi[r][a];
        for (int o = 32; o > 0; o >>= 1) {
          l = fminf(l, __shfl_xor(l, o));
          h = fmaxf(h, __shfl_xor(h, o));
        }
        if ((threadIdx.x & 63u) == 0u) {
          part[threadIdx.x >> 6][r][a] = l;
          part[threadIdx.x >> 6][r][3 + a] = h;
        }
      }
    __syncthreads();
    if (threadIdx.x < run * 6u) {
      const uint32_t r = threadIdx.x / 6u, k = threadIdx.x % 6u;
      float x = part[0][r][k];
      for (int w = 1; w < 4; w++) x = k < 3u ? fminf(x, part[w][r][k]) : fmaxf(x, part[w][r][k]);
      out[(size_t)(base + first + r) * 6u + k] = x;
    }
    __syncthreads();
    first += run;
  }
}

// ---- device-side output stage (main.cpp:363-374, output.cpp:13-23) ----
__global__ void __launch_bounds__(256) maxCountKernel(const uint32_t* __restrict__ counts, uint32_t n, uint32_t* __restrict__ out) {
  __shared__ uint32_t part[4];
  uint32_t m = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = max(m, counts[i]);
  for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
  if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out, max(max(part[0], part[1]), max(part[2], part[3])));
}

__device__ __forceinline__ uint32_t quantise8(float c) {  // output.cpp:19-21: clamp, float * double, truncated
  const float v = pt::gclamp(c, 0.0f, 1.0f);
  return (uint32_t)(unsigned char)((double)v * 255.0);
}

__device__ __forceinline__ void storeBgr(uint8_t* __restrict__ rows, uint32_t rowBytes, uint32_t width, uint32_t i, const vec3& c) {
  const uint32_t y = i / width, x = i - y * width;
  uint8_t* p = rows + (size_t)y * rowBytes + 3u * x;
  p[0] = (uint8_t)quantise8(c.z);
  p[1] = (uint8_t)quantise8(c.y);
  p[2] = (uint8_t)quantise8(c.x);
}

__global__ void __launch_bounds__(256) heatmapBytesKernel(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ maxCount,
                                                          uint32_t width, uint32_t nPixels, uint32_t rowBytes, uint8_t* __restrict__ rows) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nPixels) return;
  const float inverseMax = 1.0f / (float)maxCount[0];  // main.cpp:364-369
  const uint32_t c = counts[i];
  vec3 col(0.0f);
  if (c > 0u) col = pt::afmhot((float)c * inverseMax);
  storeBgr(rows, rowBytes, width, i, col);
}

__global__ void __launch_bounds__(256) quantiseKernel(const float* __restrict__ rgb, uint32_t width, uint32_t nPixels, uint32_t rowBytes,
                                                      uint8_t* __restrict__ rows) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nPixels) return;
  storeBgr(rows, rowBytes, width, i, vec3(rgb[3ull * i], rgb[3ull * i + 1], rgb[3ull * i + 2]));
}

// ptx_selftest_reciprocal: every float with biased exponent 1..252, both signs (2 * 252 * 2^23 values).
__global__ void __launch_bounds__(256) reciprocalCheckKernel(unsigned long long* mismatches) {
  const uint32_t stride = gridDim.x * blockDim.x;
  unsigned long long bad = 0;
  for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < 2ull * 252ull * (1ull << 23); i += stride) {
    const uint32_t sign = (uint32_t)(i & 1ull) << 31;
    const uint32_t mag = (uint32_t)(i >> 1) + (1u << 23);  // exponent field 1 .. 252
    const float x = __uint_as_float(sign | mag);
    if (__float_as_uint(rcpNewton(x)) != __float_as_uint(1.0f / x)) bad++;
  }
  if (bad) atomicAdd(mismatches, bad);
}

// Tail of gpuRender.cu:158-169 on a summed accumulation buffer.
__global__ void __launch_bounds__(256) resolveKernel(const float4* __restrict__ accum, uint32_t nPixels,
                                                     float* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nPixels) return;
  const float4 a = accum[i];
  const vec3 c = resolvePixel(vec3(a.x, a.y, a.z), a.w);
  out[3ull * i + 0] = c.x;
  out[3ull * i + 1] = c.y;
  out[3ull * i + 2] = c.z;
}

}  // namespace ptx
