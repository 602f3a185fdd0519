// Kernels of the gfx950 backend.
//
// pathKernel is a persistent, wavefront-style state machine: the grid is sized to the chip (not to
// the image), every lane owns one pixel at a time and pulls the next one from a global work counter
// when its pixel is finished, and inside the main loop each lane advances its own ray by one step
// per iteration.  The lane's whole traversal state is the one-dword node reference it holds
// (device_scene.hpp):
//
//      internal node  : two slab tests on one 64-byte NodePair, order, push far child
//      BLAS leaf      : one Moeller-Trumbore test
//      TLAS leaf      : enter the next instance (ray -> object space, marker pushed)
//      kRefInstExit   : leave the instance (ray -> world space)
//      kRefDone       : traversal finished, wait for the shading phase
//      kRefIdle       : no pixels left
//
// Every iteration all lanes issue the same memory operations first -- an LDS read of their stack
// top and ONE 64-byte fetch of the record their state needs (NodePair, TriRec or InstXform live in
// one arena and are addressed from its base) -- so an iteration pays one memory latency, not one
// per state; the state bodies then run from registers, predicated, and the iteration ends with one
// LDS write of the slot above the stack top.  TLAS and BLAS nodes are the same records, so lanes
// walking the TLAS and lanes inside an instance share the node-step code.  Lanes whose ray has
// finished wait until a quarter of the wave (or PTX_SHADE_THRESHOLD lanes) is waiting; then the
// shading phase runs once for all of them: shade the hit / look up the environment, accumulate,
// write finished pixels, pull new pixels, generate the next ray.  Because every lane always has a
// ray in flight, a wave is never held hostage by its slowest path, and an image with 100x cost
// differences between pixels (scene 1) load-balances through the counter.
//
// Per-ray visit order, ray.t at every slab test and every rounding are those of the reference's
// nested traversal (intersection.hpp:99-233), so the `-a` counts are bit-exact against the CPU
// path and the radiance path draws the same random numbers as the reference's GPU kernel.
#pragma once

#include "device_scene.hpp"
#include "shade.hpp"
#include "trace.hpp"

namespace ptx {

constexpr uint32_t kBlockThreads = 256;  // 4 waves
#ifndef PTX_LDS_STACK
#define PTX_LDS_STACK 24
#endif
#ifndef PTX_MIN_WAVES
#define PTX_MIN_WAVES 5
#endif
constexpr uint32_t kLdsStack = PTX_LDS_STACK;  // stack entries per lane kept in LDS (1 KB per entry per block)
// Tunables (see DESIGN.md "Scheduling inside a wave").
#ifndef PTX_SHADE_THRESHOLD
#define PTX_SHADE_THRESHOLD 12  // waiting lanes that trigger the shading phase
#endif
#ifndef PTX_ENTER_THRESHOLD
#define PTX_ENTER_THRESHOLD 1   // lanes at a TLAS leaf before the (rare, long) instance-entry body is run
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
__device__ __forceinline__ uint32_t stackRead(const StackMem& m, uint32_t slot, bool deep) {
  uint32_t v;
  if (!deep) {
    v = m.lds[slot * kBlockThreads];
  } else {
    // both loads, then a value select: a pointer select would become a (slow, doubly counted) flat load
    const uint32_t a = m.lds[(slot < kLdsStack ? slot : kLdsStack - 1u) * kBlockThreads];
    const uint32_t b = m.overflow[(size_t)(slot < kLdsStack ? 0u : slot - kLdsStack) * m.stride];
    v = slot < kLdsStack ? a : b;
  }
  return v;
}
__device__ __forceinline__ void stackWrite(const StackMem& m, uint32_t slot, uint32_t val, bool deep) {
  if (!deep) {
    m.lds[slot * kBlockThreads] = val;
  } else {
    if (slot < kLdsStack) m.lds[slot * kBlockThreads] = val;
    else m.overflow[(size_t)(slot - kLdsStack) * m.stride] = val;
  }
}

// Work item -> pixel.  The image is cut into bands of 8 rows; inside a band items run down a column
// first, so 64 consecutive items (one wave's first fetch) form an 8x8 block of adjacent pixels for
// any image size, with no padding items.
__device__ __forceinline__ void workPixel(const FrameConstants& fc, uint32_t w, uint32_t& x, uint32_t& y) {
  const uint32_t bandItems = fc.width * 8u;
  const uint32_t band = w / bandItems;
  const uint32_t r = w - band * bandItems;
  const uint32_t rows = min(8u, fc.height - band * 8u);
  const uint32_t cx = r / rows;
  x = cx;
  y = band * 8u + (r - cx * rows);
}

template <int MODE, bool STATS, bool RESOLVE>
__global__ void __launch_bounds__(kBlockThreads, PTX_MIN_WAVES) pathKernel(DeviceScene sc, FrameConstants fc, uint32_t* workCounter,
                                                            void* __restrict__ outRaw, DeviceStats* stats,
                                                            uint32_t* stackOverflow) {
  __shared__ uint32_t sStack[kLdsStack * kBlockThreads];
  StackMem stk;
  stk.lds = sStack + threadIdx.x;
  stk.overflow = stackOverflow + (blockIdx.x * kBlockThreads + threadIdx.x);
  stk.stride = gridDim.x * kBlockThreads;
  uint32_t sp = 1;

  const uint32_t totalWork = fc.width * fc.height;
  const vec3 camPos(fc.camPos[0], fc.camPos[1], fc.camPos[2]);

  // ---- lane state ----
  uint32_t ref = kRefDone;   // what the lane does next (see the table above)
  bool pending = false;      // a traced ray's result waits for the shading phase
  bool havePixel = false;
  bool zeroDir = false;      // a component of the current ray's direction is exactly 0 (0*inf possible)
  uint32_t pix = 0, sample = 0, bounce = 0, rng = 0, count = 0;
  uint32_t tlasLeaf = 0, curInst = 0;
  float tBeforeInst = FLT_MAX;
  vec3 sum(0.0f), col(1.0f);
  vec3 wo(0.0f), wd(0.0f, 0.0f, -1.0f), winv(0.0f);  // world-space ray
  vec3 o(0.0f), d(0.0f, 0.0f, -1.0f), inv(0.0f);     // current ray (world or object space)
  Hit hit;
  hit.t = FLT_MAX; hit.u = 0.0f; hit.v = 0.0f; hit.tri = 0xFFFFFFFFu; hit.inst = 0xFFFFFFFFu;
  sample = (MODE == kModeCounts) ? 1u : fc.spp;  // "pixel finished": the first shading phase pulls work

  uint32_t nRays = 0, nNodeSteps = 0, nTriTests = 0, nInstEntries = 0, nHits = 0, nMisses = 0;
  unsigned long long nBoxHits = 0;

  uint32_t loopCount = 0;
  for (;;) {
    loopCount++;
    // =========================== shading phase (postponed, wave-wide) ===========================
    const bool waiting = ref == kRefDone;
    const uint32_t nWaiting = (uint32_t)__popcll(__ballot(waiting));
    const uint32_t nLive = 64u - (uint32_t)__popcll(__ballot(ref == kRefIdle));
    if (nLive == 0u) break;
    if (nWaiting >= PTX_SHADE_THRESHOLD || nWaiting * 4u >= nLive) {
      if (waiting) {
        bool pathEnded = true;
        if (pending) {
          pending = false;
          if (STATS) {
            nRays++;
            nBoxHits += count;
          }
          if (MODE == kModeCounts) {
            reinterpret_cast<uint32_t*>(outRaw)[pix] = count;
          } else if (hit.t < FLT_MAX) {
            if (STATS) nHits++;
            shadeHit(sc, hit, wo, wd, rng, col);  // writes the next ray into wo / wd
            bounce++;
            pathEnded = bounce >= fc.maxBounces;  // gpuRender.cu:29: the loop ends, col is returned
          } else {
            if (STATS) nMisses++;
            col *= environmentLookup(sc, fc, wd);
          }
          if (MODE == kModeRadiance && pathEnded) {
            sum += col;
            sample++;
          }
        }
        bool live = true;
        if (pathEnded) {
          // ---- pixel finished? write it and pull the next one ----
          const uint32_t perPixel = (MODE == kModeCounts) ? 1u : fc.spp;
          while (sample >= perPixel && live) {
            if (MODE == kModeRadiance && havePixel) {
              if (RESOLVE) {
                const vec3 c = resolvePixel(sum, (float)fc.spp);
                float* out = reinterpret_cast<float*>(outRaw);
                out[3ull * pix + 0] = c.x;
                out[3ull * pix + 1] = c.y;
                out[3ull * pix + 2] = c.z;
              } else {
                reinterpret_cast<float4*>(outRaw)[pix] = make_float4(sum.x, sum.y, sum.z, (float)fc.spp);
              }
            }
            const uint32_t w = atomicAdd(workCounter, 1u);  // aggregated per wave by the compiler
            if (w >= totalWork) {
              live = false;
              havePixel = false;
              break;
            }
            uint32_t x, y;
            workPixel(fc, w, x, y);
            pix = y * fc.width + x;
            sum = vec3(0.0f);
            rng = fc.seedBase + pix;  // gpuRender.cu:130
            havePixel = true;
            sample = 0u;
            if (MODE == kModeCounts) {
              if (fc.partCount > 1u) {
                const uint32_t tile = (y >> 3) * fc.tilesX + (x >> 3);
                if ((tile % fc.partCount) != fc.partIndex) {
                  reinterpret_cast<uint32_t*>(outRaw)[pix] = 0u;
                  sample = 1u;  // not this part's tile: finished
                }
              }
            } else if (fc.maxBounces == 0u) {
              sum = vec3((float)fc.spp);  // no segment is traced: every path returns white (gpuRender.cu:27-29)
              sample = fc.spp;
            }
          }
          if (live) {
            // ---- next camera ray: gpuRender.cu:132-143 ----
            const uint32_t py = pix / fc.width, px = pix - py * fc.width;
            float fx = (float)px, fy = (float)py;
            if (MODE == kModeRadiance && fc.spp > 1u) {
              const vec2 xi = pt::getRandomVec2(rng);
              fx += 0.5f * (2.0f * xi.x - 1.0f);
              fy += 0.5f * (2.0f * xi.y - 1.0f);
            }
            wo = camPos;
            wd = primaryDirection(fc, fx, fy);
            col = vec3(1.0f);
            bounce = 0u;
            if (MODE == kModeCounts) sample = 1u;  // one ray per pixel
          }
        }
        if (live) {
          // ---- launch the segment: Scene::intersect starts at the TLAS root ----
          winv = 1.0f / wd;
          o = wo; d = wd; inv = winv;
          zeroDir = (wd.x == 0.0f) || (wd.y == 0.0f) || (wd.z == 0.0f);
          hit.t = FLT_MAX; hit.u = 0.0f; hit.v = 0.0f; hit.tri = 0xFFFFFFFFu; hit.inst = 0xFFFFFFFFu;
          count = 0u;
          tlasLeaf = 0u;
          stk.lds[0] = kRefDone;  // slot 0: bottom-of-stack marker
          sp = 1u;
          ref = sc.tlasRootRef;
          pending = true;
        } else {
          ref = kRefIdle;
        }
      }
    }

    // ============================== traversal steps, one per lane each ===============================
#pragma unroll 1
    for (int it = 0; it < PTX_INNER_ITERS; it++) {
      PTX_MARK("TRAV_BEGIN");
      const bool isNode = (int32_t)ref >= 0;
      const bool isSpecial = ref >= kRefIdle;  // idle / done / instance exit
      bool isExit = ref == kRefInstExit;
      const bool isTri = (ref >> 30) == 2u;
      bool isEnter = (ref >> 30) == 3u && !isSpecial;
      if (PTX_ENTER_THRESHOLD > 1) {
        // the instance-entry body is long and rarely needed: run it when a few lanes want it, or when
        // nothing else is left to do in this wave
        const uint32_t nEnter = (uint32_t)__popcll(__ballot(isEnter));
        if (nEnter < PTX_ENTER_THRESHOLD && __any(isNode || isTri)) isEnter = false;
      }
      if (PTX_RARE_PERIOD > 1) {
        // instance entry / exit bodies run on every PTX_RARE_PERIOD-th iteration only (a lane waits at
        // most PTX_RARE_PERIOD - 1 iterations for them), unless no lane has anything else to do
        if ((loopCount % PTX_RARE_PERIOD) != 0u && __any(isNode || isTri)) {
          isEnter = false;
          isExit = false;
        }
      }
      const bool steps = isNode || isTri || isEnter || isExit;
      const uint32_t first = ref & (isTri ? 0x007FFFFFu : 0x00FFFFFFu);
      const uint32_t nPrim = (isTri ? ((ref >> 23) & 0x7Fu) : ((ref >> 24) & 0x3Fu)) + 1u;

      const bool deep = __any(sp + 2u >= kLdsStack);  // wave-uniform; false almost always
      const uint32_t top = stackRead(stk, sp ? sp - 1u : 0u, deep);

      // record address in 16-byte units from the arena base; lanes with nothing to fetch read the
      // first record of the arena (same line for all of them) so that the loads need no predication
      const uint32_t off16 = isNode ? sc.pairs16 + ref * 4u : (isTri ? sc.tris16 + first * 3u : (isEnter ? sc.inst16 + first * 4u : 0u));
      const float4* rp = sc.arena + off16;
      const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];

      uint32_t newRef = ref;
      uint32_t pushVal = 0u;
      bool doPush = false, doPop = false;

      PTX_MARK("EXIT");
      if (isExit) {
        // GPUMesh::intersect returns (ray.t = transformedRay.t); intersectTLAS keeps the instance if it
        // lowered t (intersection.hpp:184-189).
        if (hit.t < tBeforeInst) hit.inst = curInst;
        o = wo; d = wd; inv = winv;
        zeroDir = (wd.x == 0.0f) || (wd.y == 0.0f) || (wd.z == 0.0f);
        doPop = tlasLeaf == 0u;
        newRef = doPop ? top : tlasLeaf;
        tlasLeaf = 0u;
      }
      PTX_MARK("ENTER");
      if (isEnter) {
        // TLAS leaf: enter its first instance, keep the rest of the leaf (gpuFunctions.hpp:32-41).
        tlasLeaf = (nPrim > 1u) ? makeTlasLeafRef(first + 1u, nPrim - 1u) : 0u;
        curInst = first;
        o = xformRows(r0, r1, r2, wo, 1.0f);
        d = xformRows(r0, r1, r2, wd, 0.0f);  // not normalised: t stays world-space
        inv = 1.0f / d;
        zeroDir = (d.x == 0.0f) || (d.y == 0.0f) || (d.z == 0.0f);
        tBeforeInst = hit.t;
        if (STATS) nInstEntries++;
        pushVal = kRefInstExit;
        doPush = true;
        newRef = __float_as_uint(r3.x);  // InstXform::rootRef
      }
      PTX_MARK("TRI");
      if (isTri) {
        // BLAS leaf, one triangle per step, in order, strict `<` (intersection.hpp:9-44, :112-122).
        TriRec tri;
        tri.v0 = r0; tri.e1 = r1; tri.e2 = r2;
        float u = 0.0f, v = 0.0f;
        const float dist = triangleTest(o, d, tri, u, v);
        if (dist < hit.t) {
          hit.t = dist; hit.u = u; hit.v = v; hit.tri = first;
        }
        if (STATS) nTriTests++;
        doPop = nPrim == 1u;
        newRef = doPop ? top : makeLeafRef(first + 1u, nPrim - 1u);
      }
      PTX_MARK("NODE");
      if (isNode) {
        // Internal node (intersection.hpp:133-161 / :201-230): one 64-byte record holds both children.
        float d1, d2;
        if (__any(zeroDir)) {
          d1 = slabTest<true>(o, inv, hit.t, r0, r1);
          d2 = slabTest<true>(o, inv, hit.t, r2, r3);
        } else {
          d1 = slabTest<false>(o, inv, hit.t, r0, r1);
          d2 = slabTest<false>(o, inv, hit.t, r2, r3);
        }
        const uint32_t rl = __float_as_uint(r0.w), rr = __float_as_uint(r2.w);
        if (STATS) nNodeSteps++;
        const bool swap = d1 > d2;  // nearer child first; ties keep the left child
        const float dn = swap ? d2 : d1, df = swap ? d1 : d2;
        const uint32_t rn = swap ? rr : rl, rf = swap ? rl : rr;
        const bool hitNear = dn != FLT_MAX, hitFar = df != FLT_MAX;  // far hit implies near hit
        count += (hitNear ? 1u : 0u) + (hitFar ? 1u : 0u);  // counted when pushed, not re-tested when popped
        pushVal = rf;
        doPush = hitFar;
        doPop = !hitNear;
        newRef = hitNear ? rn : top;
      }

      PTX_MARK("MERGE");
      if (steps) {
        stackWrite(stk, sp, pushVal, deep);
        sp = sp + (doPush ? 1u : 0u) - (doPop ? 1u : 0u);
        ref = newRef;
      }
      PTX_MARK("TRAV_END");
    }
  }

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
