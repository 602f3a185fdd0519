// Radiance kernel with a block-level exchange of traversal steps (wide layout only).
//
// pathKernel (kernels.hpp) lets every lane take the step its own ray needs, so a wave64 pays for the node
// body, the triangle body and the instance-entry body in almost every iteration although each is wanted by a
// part of its lanes (profiles/r02: 27 / 20 / 2 lanes of 64; 42 % of the VALU lanes do useful work, and the
// kernel is bound by VALU issue).  Here a ray still lives in the registers of its lane -- state, stack,
// shading never move -- but the STEP is handed to another lane of the 256-thread block:
//
//   post    every lane writes what its step needs (8 dwords: origin, reciprocal direction / direction, ray.t,
//           the reference) into its own slot in LDS and enters itself into the list of its kind (node,
//           triangle, instance entry); ranks come from a ballot and one LDS atomic per wave and kind
//   ------- barrier
//   serve   thread j takes request j of the lists laid end to end -- nodes | triangles | entries -- so wave 0
//           runs 64 node steps, the last busy wave the entries, and only a wave on a boundary runs two bodies;
//           it reads the request, fetches the record (the one global access of the step), computes, and writes
//           the result (sorted children / t,u,v / object-space ray) back into the requester's slot
//   ------- barrier
//   merge   every lane reads its result and updates its ray: closest hit, stack, next reference
//
// The shading phase stays with the lane that owns the ray and is entered by the whole block at once (a wave
// that shaded alone would hold the other three at the barrier); it is triggered by the block's count of
// finished rays.  Same arithmetic as pathKernel on the wide layout, same order of a ray's steps: images are
// bit-identical to pathKernel's (checked: same CRC of the 1500x800 100-spp frame).
//
// MEASURED AND NOT SHIPPED (built only with -DPTX_EXPERIMENT_EXCHANGE, selected with PTX_EXCHANGE=1): 656 ms against
// pathKernel's 352 ms on the bench frame (profiles/r02/exchange_kernel_ab.txt).  The bodies do run at high lane
// utilisation, but an iteration is now a chain of two block barriers and five dependent LDS round trips (counters,
// list entry, request, result, next counters) around the record fetch, ~9,300 cycles per wave-iteration against
// ~4,800 (profiles/r02/exchange_kernel_time_profile.txt: posting alone 32 %, the fetch wait 1.4 %), and with 96
// VGPRs and 32 KB of LDS per block there are only five such chains per SIMD to overlap: the SIMDs idle.  What
// pathKernel pays in idle lanes is less than what the exchange pays in waiting waves.
#pragma once

#include "../path-tracer_amd/csrc/kernels.hpp"

namespace ptx {

#ifndef PTX_X_WAVES
#define PTX_X_WAVES 5
#endif
#ifndef PTX_X_STACK
#define PTX_X_STACK 19      // stack entries per lane in LDS (19 KB) beside the 12-KB exchange slots: 5 blocks per CU
#endif
#ifndef PTX_X_SHADE_T
#define PTX_X_SHADE_T 64    // finished rays in the block that trigger the shading phase
#endif
constexpr uint32_t kXStack = PTX_X_STACK;
constexpr uint32_t kXSlot = 12u;  // dwords per exchange slot; 12 = 4 * 3: b128 accesses of consecutive lanes are conflict-free

template <bool STATS>
__global__ void __launch_bounds__(kBlockThreads, PTX_X_WAVES) exchangeKernel(DeviceScene sc, FrameConstants fc, uint32_t* workCounter,
                                                                           void* __restrict__ outRaw, DeviceStats* stats,
                                                                           uint32_t* stackOverflow) {
  constexpr uint32_t CAP = kXStack;
  __shared__ uint32_t sStack[CAP * kBlockThreads];
  __shared__ __align__(16) uint32_t sSlot[kBlockThreads * kXSlot];
  __shared__ uint8_t sPerm[3 * kBlockThreads];   // [kind][rank] -> thread that posted the request
  __shared__ __align__(16) uint32_t sCnt[2][8];  // [iteration parity][nodes, triangles, entries, waiting, live, -, -, -]
  StackMem stk;
  stk.lds = sStack + threadIdx.x;
  stk.overflow = stackOverflow + (blockIdx.x * kBlockThreads + threadIdx.x);
  stk.stride = gridDim.x * kBlockThreads;
  uint32_t sp = 1;
  uint4* const mySlot = reinterpret_cast<uint4*>(sSlot + threadIdx.x * kXSlot);
  const uint32_t laneId = threadIdx.x & 63u;

  const uint32_t nPixels = fc.width * fc.height;
  const uint32_t totalWork = nPixels * fc.subStreams;  // checked on the host to fit 32 bits
  const vec3 camPos(fc.camPos[0], fc.camPos[1], fc.camPos[2]);

  // ---- lane state (as in pathKernel) ----
  uint32_t ref = kRefDone;
  bool havePixel = false;
  bool zeroDir = false, wZeroDir = false;
  uint32_t pix = 0, sample = 0, bounce = 0, rng = 0;
  uint32_t tlasLeaf = 0, curInst = 0;
  bool inInst = false;
  vec3 sum(0.0f), col(1.0f);
  vec3 wo(0.0f), wd(0.0f, 0.0f, -1.0f), winv(0.0f);
  vec3 o(0.0f), d(0.0f, 0.0f, -1.0f), inv(0.0f);
  Hit hit;
  hit.t = FLT_MAX; hit.u = 0.0f; hit.v = 0.0f; hit.tri = 0xFFFFFFFFu; hit.inst = 0xFFFFFFFFu;

  uint32_t nRays = 0, nNodeSteps = 0, nTriTests = 0, nInstEntries = 0, nHits = 0, nMisses = 0;
  unsigned long long nBoxHits = 0;
  uint32_t count = 0;
#ifdef PTX_WAVE_PROFILE
  unsigned long long prof[16] = {};
#ifdef PTX_TIME_PROFILE
  unsigned long long tprof[8] = {};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif
#endif

  if (threadIdx.x < 16u) (&sCnt[0][0])[threadIdx.x] = 0u;
  __syncthreads();

  uint32_t iter = 0;
  for (;; iter++) {
    const uint32_t par = iter & 1u;
    // ---- block-uniform: finished / live rays as counted when the previous iteration posted ----
    const uint32_t nWaitBlock = sCnt[par ^ 1u][3], nLiveBlock = sCnt[par ^ 1u][4];
    if (iter != 0u && nLiveBlock == 0u) break;
    const bool shadeNow = iter == 0u || nWaitBlock >= PTX_X_SHADE_T || nWaitBlock * 4u >= nLiveBlock;
    PTX_PROF(0, ref != kRefIdle);
    if (shadeNow) {
      const bool waiting = ref == kRefDone;
      PTX_PROF(1, waiting);
      if (waiting) {
        bool pathEnded = true;
        if (havePixel) {
          if (STATS) {
            nRays++;
            nBoxHits += count;
          }
          if (hit.t < FLT_MAX) {
            if (STATS) nHits++;
            shadeHit(sc, hit, wo, wd, rng, col);  // writes the next ray into wo / wd
            bounce++;
            pathEnded = bounce >= fc.maxBounces;  // gpuRender.cu:29
          } else {
            if (STATS) nMisses++;
            col *= environmentLookup(sc, fc, wd);
          }
          if (pathEnded) {
            sum += col;
            sample--;
          }
        }
        bool live = true;
        if (pathEnded) {
          while (sample == 0u && live) {
            if (havePixel) {
              const uint32_t p = pix & 0x00FFFFFFu, k = pix >> 24;
              const uint32_t n = fc.sppBase + (k < fc.sppExtra ? 1u : 0u);
              reinterpret_cast<float4*>(outRaw)[(size_t)k * nPixels + p] = make_float4(sum.x, sum.y, sum.z, (float)n);
            }
            const uint32_t w = atomicAdd(workCounter, 1u);
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
            rng = fc.seedBase + p + k * 0x9E3779B9u;  // gpuRender.cu:130 (+ stream offset)
            havePixel = true;
            sample = fc.sppBase + (k < fc.sppExtra ? 1u : 0u);
            if (fc.maxBounces == 0u) {
              sum = vec3((float)sample);
              sample = 0u;
            }
          }
          if (live) {
            const uint32_t p = pix & 0x00FFFFFFu;
            const uint32_t py = divByWidth(fc, p), px = p - py * fc.width;
            float fx = (float)px, fy = (float)py;
            if (fc.jitter) {
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
          reciprocalDir(wd, winv, wZeroDir);
          zeroDir = wZeroDir;
          o = wo; d = wd; inv = winv;
          hit.t = FLT_MAX; hit.u = 0.0f; hit.v = 0.0f; hit.tri = 0xFFFFFFFFu; hit.inst = 0xFFFFFFFFu;
          count = 0u;
          tlasLeaf = 0u;
          inInst = false;
          stk.lds[0] = kRefDone;
          sp = 1u;
          ref = sc.tlasRootRefWide;
        } else {
          ref = kRefIdle;
        }
      }
    }

    PTX_STAMP(0);  // loop top + shading phase
    // =============================================== post ===============================================
    const bool isNode = (int32_t)ref >= 0;
    const bool isTri = (ref >> 30) == 2u;
    const bool isEnter = (ref >> 30) == 3u && ref < kRefIdle;
    const uint32_t first = ref & (isTri ? 0x007FFFFFu : 0x00FFFFFFu);
    const uint32_t nPrim = (isTri ? ((ref >> 23) & 0x7Fu) : ((ref >> 24) & 0x3Fu)) + 1u;
    const unsigned long long mN = __ballot(isNode), mT = __ballot(isTri), mE = __ballot(isEnter);
    const uint32_t nN = (uint32_t)__popcll(mN), nT = (uint32_t)__popcll(mT), nE = (uint32_t)__popcll(mE);
    const uint32_t nW = (uint32_t)__popcll(__ballot(ref == kRefDone)), nL = 64u - (uint32_t)__popcll(__ballot(ref == kRefIdle));
    uint32_t bN = 0u, bT = 0u, bE = 0u;
    if (laneId == 0u) {
      bN = atomicAdd(&sCnt[par][0], nN);
      bT = atomicAdd(&sCnt[par][1], nT);
      bE = atomicAdd(&sCnt[par][2], nE);
      atomicAdd(&sCnt[par][3], nW);
      atomicAdd(&sCnt[par][4], nL);
    }
    const bool deep = __any(sp + 4u >= CAP);
    const uint32_t top = stackRead<CAP>(stk, sp ? sp - 1u : 0u, deep);
    // request: 8 dwords.  node: o, inv, ray.t, reference | zero-direction flag in bit 31; triangle: o, d, -, first
    // triangle; entry: world o, world d, -, first instance
    {
      const vec3 a = isEnter ? wo : o;
      const vec3 b = isNode ? inv : (isTri ? d : wd);
      const uint32_t r = isNode ? (ref | (zeroDir ? 0x80000000u : 0u)) : first;
      mySlot[0] = make_uint4(__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(b.x));
      mySlot[1] = make_uint4(__float_as_uint(b.y), __float_as_uint(b.z), __float_as_uint(hit.t), r);
    }
    bN = (uint32_t)__builtin_amdgcn_readfirstlane((int)bN);
    bT = (uint32_t)__builtin_amdgcn_readfirstlane((int)bT);
    bE = (uint32_t)__builtin_amdgcn_readfirstlane((int)bE);
    {
      const unsigned long long below = (1ull << laneId) - 1ull;
      const uint32_t dest = isNode ? bN + (uint32_t)__popcll(mN & below)
                                   : (isTri ? kBlockThreads + bT + (uint32_t)__popcll(mT & below) : 2u * kBlockThreads + bE + (uint32_t)__popcll(mE & below));
      if (isNode || isTri || isEnter) sPerm[dest] = (uint8_t)threadIdx.x;
    }
    PTX_STAMP(1);  // post
    __syncthreads();
    PTX_STAMP(2);  // barrier after post

    // =============================================== serve ==============================================
    {
      if (threadIdx.x < 8u) sCnt[par ^ 1u][threadIdx.x] = 0u;  // the other parity's counters: read by everybody before this barrier
      const uint4 cnt = *reinterpret_cast<const uint4*>(&sCnt[par][0]);
      const uint32_t Nn = cnt.x, Nt = cnt.y, Ne = cnt.z;
      const uint32_t j = threadIdx.x;
      const bool sNode = j < Nn;
      const bool sTri = !sNode && j < Nn + Nt;
      const bool sEnter = !sNode && !sTri && j < Nn + Nt + Ne;
      if (sNode || sTri || sEnter) {
        const uint32_t src = sPerm[sNode ? j : (sTri ? kBlockThreads + (j - Nn) : 2u * kBlockThreads + (j - Nn - Nt))];
        uint4* const slot = reinterpret_cast<uint4*>(sSlot + src * kXSlot);
        const uint4 q0 = slot[0], q1 = slot[1];
        const vec3 ra(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
        const vec3 rb(__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y));
        const float rt = __uint_as_float(q1.z);
        const uint32_t rr = q1.w;
        const uint32_t off16 = sNode ? sc.wide16 + (rr & 0x7FFFFFFFu) * 4u : (sTri ? sc.tris16 + rr * 3u : sc.inst16 + rr * 4u);
        const float4* rp = sc.arena + off16;
        float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3;
        asm volatile("" : "=v"(r3.x), "=v"(r3.y), "=v"(r3.z), "=v"(r3.w));
        if (!sTri) r3 = rp[3];
        PTX_STAMP(3);  // serve: counters, list entry, request, fetch issued
        PTX_WAIT_ALL();
        PTX_STAMP(4);  // waiting for the record
        PTX_PROF(6, sNode);
        PTX_PROF(5, sTri);
        PTX_PROF(4, sEnter);
        if (sNode) {
          uint32_t sr[4];
          uint32_t nHit;
          if (__any((int32_t)rr < 0)) nHit = wideStep<true>(ra, rb, rt, r0, r1, r2, r3, sr);
          else nHit = wideStep<false>(ra, rb, rt, r0, r1, r2, r3, sr);
          slot[0] = make_uint4(sr[0], sr[1], sr[2], sr[3]);
          reinterpret_cast<uint32_t*>(slot)[4] = nHit;
        }
        if (sTri) {
          TriRec tri;
          tri.v0 = r0; tri.e1 = r1; tri.e2 = r2;
          float u = 0.0f, v = 0.0f;
          const float dist = triangleTest(ra, rb, tri, u, v);
          slot[0] = make_uint4(__float_as_uint(dist), __float_as_uint(u), __float_as_uint(v), 0u);
        }
        if (sEnter) {
          const vec3 no = xformRows(r0, r1, r2, ra, 1.0f);
          const vec3 nd = xformRows(r0, r1, r2, rb, 0.0f);  // not normalised: t stays world-space
          vec3 ninv;
          bool nz;
          reciprocalDir(nd, ninv, nz);
          slot[0] = make_uint4(__float_as_uint(no.x), __float_as_uint(no.y), __float_as_uint(no.z), __float_as_uint(nd.x));
          slot[1] = make_uint4(__float_as_uint(nd.y), __float_as_uint(nd.z), __float_as_uint(ninv.x), __float_as_uint(ninv.y));
          slot[2] = make_uint4(__float_as_uint(ninv.z), nz ? 1u : 0u, __float_as_uint(r3.y), 0u);  // InstXform::wideRootRef
        }
      }
    }
    PTX_STAMP(5);  // serve: compute, result written
    __syncthreads();
    PTX_STAMP(6);  // barrier after serve

    // =============================================== merge ==============================================
    {
      uint32_t newRef = ref;
      bool doPop = false;
      uint32_t nPush = 0u, pushVal = 0u, push2 = 0u, push3 = 0u;
      const bool steps = isNode || isTri || isEnter;
      if (isEnter) {
        const uint4 a = mySlot[0], b = mySlot[1], c = mySlot[2];
        tlasLeaf = (nPrim > 1u) ? makeTlasLeafRef(first + 1u, nPrim - 1u) : 0u;
        curInst = first;
        inInst = true;
        o = vec3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
        d = vec3(__uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y));
        inv = vec3(__uint_as_float(b.z), __uint_as_float(b.w), __uint_as_float(c.x));
        zeroDir = c.y != 0u;
        if (STATS) nInstEntries++;
        newRef = c.z;
      }
      if (isTri) {
        const uint4 a = mySlot[0];
        const float dist = __uint_as_float(a.x);
        if (dist < hit.t) {
          hit.t = dist; hit.u = __uint_as_float(a.y); hit.v = __uint_as_float(a.z); hit.tri = first; hit.inst = curInst;
        }
        if (STATS) nTriTests++;
        doPop = nPrim == 1u;
        newRef = doPop ? top : makeLeafRef(first + 1u, nPrim - 1u);
      }
      if (isNode) {
        const uint4 a = mySlot[0];
        const uint32_t nHit = reinterpret_cast<const uint32_t*>(mySlot)[4];
        if (STATS) nNodeSteps++;
        count += nHit;
        doPop = nHit == 0u;
        newRef = doPop ? top : a.x;
        nPush = doPop ? 0u : nHit - 1u;
        pushVal = a.y; push2 = a.z; push3 = a.w;
      }
      {
        // leaving an instance: see pathKernel
        const bool leaving = doPop && inInst && (top < sc.tlasNodeEnd[1] || top >= kTlasLeafFlag);
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
      if (steps) {
        const uint32_t v0 = nPush == 3u ? push3 : (nPush == 2u ? push2 : pushVal);
        const uint32_t v1 = nPush == 3u ? push2 : pushVal;
        stackWrite<CAP>(stk, sp, v0, deep);
        if (__any(nPush >= 2u)) {
          stackWrite<CAP>(stk, sp + 1u, v1, deep);
          stackWrite<CAP>(stk, sp + 2u, pushVal, deep);
        }
        sp = sp + nPush - (doPop ? 1u : 0u);
        ref = newRef;
      }
    }
    PTX_STAMP(7);  // merge
  }

#ifdef PTX_WAVE_PROFILE
  if (STATS && (threadIdx.x & 63u) == 0u)
    for (int k = 0; k < 16; k++) atomicAdd(&gWaveProfile[k], prof[k]);
#ifdef PTX_TIME_PROFILE
  if (STATS && (threadIdx.x & 63u) == 0u) {
    for (int k = 0; k < 8; k++) atomicAdd(&gWaveProfile[16 + k], tprof[k]);
    atomicAdd(&gWaveProfile[0], (unsigned long long)iter);
  }
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

}  // namespace ptx
