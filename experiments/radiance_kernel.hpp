// Radiance kernel of the gfx950 backend: wave-uniform phases over a persistent grid, K rays per lane.
//
// pathKernel (kernels.hpp) lets every lane take whatever step its one ray needs in every iteration --
// node, triangle, instance entry, instance exit -- so a wave64 pays for all four bodies almost every
// time although each is wanted by a part of its lanes only (round 1: 40 % of the VALU lanes did useful
// work; the instance bodies ran in 80 % of the iterations for 2.5 lanes of 64).  Two changes here:
//
//  1. An iteration runs ONE kind of step, chosen for the whole wave from a census of its lanes
//     (ballots, scalar code), as a wave-uniform branch:
//        NODE    one WideNode (wide_bvh.hpp): up to four quantised child boxes, sort, push up to three
//        TRI     one Moeller-Trumbore test of a PARKED LEAF
//        ENTER   enter the next instance of a TLAS leaf (ray -> object space, exit marker pushed)
//        SHADE   shade the hit / look up the environment, accumulate, pull work, generate the next ray
//  2. Every lane owns PTX_RAD_RAYS rays (registers), not one, and takes part in an iteration with
//     whichever of its rays wants that kind of step.  With one ray per lane about half the lanes wait in
//     every iteration; with two independent rays a lane is idle only if neither wants the step.  This is
//     the wave64 form of a ray queue: instead of moving rays between lanes (their state is ~25 dwords;
//     moving it through LDS every step costs more than the step), the pool a wave picks from is made
//     larger than the wave.
//
// A ray that reaches a BLAS leaf parks it in a register (`leaf`) and goes on walking nodes -- the
// next-nearest child, or what its stack holds -- until a second leaf turns up; TRI iterations drain the
// parked leaves.  So a ray can usually take part in either kind, at the price of node steps made with
// a ray.t that a parked triangle would have shortened.  The order in which a ray visits its leaves now
// depends on what the other rays of its wave do, so the closest hit must not depend on that order:
// equal distances are resolved towards the smaller triangle index (the reference keeps the first one
// it meets; radiance only -- the `-a` heat map keeps pathKernel and the reference's exact visit order).
// Leaving an instance costs no iteration of its own: a ray whose stack top is the exit marker is
// restored to world space at the end of any iteration it did not step in.
//
// Traversal stacks: PTX_RAD_STACK entries per ray in LDS, [ray][entry][thread] (conflict-free); deeper
// entries (0.05 % of the steps on scene 1) go to the global overflow area, [entry][ray][thread], by a
// per-lane branch.
#pragma once

#include "../path-tracer_amd/csrc/kernels.hpp"

namespace ptx {

#ifndef PTX_RAD_WAVES
#define PTX_RAD_WAVES 4   // waves per SIMD the register budget is cut for (512 / 4 = 128 VGPRs)
#endif
#ifndef PTX_RAD_RAYS
#define PTX_RAD_RAYS 2    // rays per lane
#endif
#ifndef PTX_RAD_STACK
// LDS per CU is 160 KB = 40 dwords per lane at 4 blocks of 256 lanes
#define PTX_RAD_STACK ((160u / PTX_RAD_WAVES) / PTX_RAD_RAYS)
#endif
constexpr uint32_t kRadRays = PTX_RAD_RAYS;
constexpr uint32_t kRadStack = PTX_RAD_STACK;
#ifndef PTX_RAD_SHADE_T
#define PTX_RAD_SHADE_T 32  // lanes holding a finished ray before a SHADE iteration is run
#endif
#ifndef PTX_RAD_ENTER_T
#define PTX_RAD_ENTER_T 16  // lanes holding a ray at a TLAS leaf before an ENTER iteration is run
#endif
#ifndef PTX_RAD_TRI_W
#define PTX_RAD_TRI_W 1     // TRI is chosen over NODE when nTri * PTX_RAD_TRI_W > nNode * PTX_RAD_NODE_W
#endif
#ifndef PTX_RAD_NODE_W
#define PTX_RAD_NODE_W 1
#endif

#if defined(PTX_WAVE_PROFILE)
__device__ unsigned long long gSpHist[128];  // [sp] at node steps, [64 + sp] at triangle steps (profiling builds only)
#define PTX_RPROF(k, lanes) do { if (STATS) { prof[2 * (k)] += 1u; prof[2 * (k) + 1] += (uint32_t)__popcll(__ballot(lanes)); } } while (0)
#else
#define PTX_RPROF(k, lanes) do {} while (0)
#endif

// a[sel] for a per-lane index into a register array (K - 1 selects per dword)
template <typename T, uint32_t K>
__device__ __forceinline__ T pick(const T (&a)[K], uint32_t sel) {
  T v = a[0];
#pragma unroll
  for (uint32_t k = 1; k < K; k++) v = sel == k ? a[k] : v;
  return v;
}
template <typename T, uint32_t K>
__device__ __forceinline__ void put(T (&a)[K], uint32_t sel, const T& v) {
#pragma unroll
  for (uint32_t k = 0; k < K; k++) a[k] = sel == k ? v : a[k];
}

// vec3: selected per component (a select between two structs becomes a select between their ADDRESSES,
// which would push the arrays out of the registers)
template <uint32_t K>
__device__ __forceinline__ vec3 pick(const vec3 (&a)[K], uint32_t sel) {
  float x = a[0].x, y = a[0].y, z = a[0].z;
#pragma unroll
  for (uint32_t k = 1; k < K; k++) {
    x = sel == k ? a[k].x : x;
    y = sel == k ? a[k].y : y;
    z = sel == k ? a[k].z : z;
  }
  return vec3(x, y, z);
}
template <uint32_t K>
__device__ __forceinline__ void put(vec3 (&a)[K], uint32_t sel, const vec3& v) {
#pragma unroll
  for (uint32_t k = 0; k < K; k++) {
    a[k].x = sel == k ? v.x : a[k].x;
    a[k].y = sel == k ? v.y : a[k].y;
    a[k].z = sel == k ? v.z : a[k].z;
  }
}

struct RadStack {
  uint32_t* lds;       // &sStack[threadIdx.x]
  uint32_t* overflow;  // &overflowBase[global thread]
  uint32_t stride;     // total threads of the launch
};
__device__ __forceinline__ uint32_t radStackRead(const RadStack& m, uint32_t ray, uint32_t slot) {
  uint32_t v;
  if (slot < kRadStack) v = m.lds[(ray * kRadStack + slot) * kBlockThreads];
  else v = m.overflow[(size_t)((slot - kRadStack) * kRadRays + ray) * m.stride];
  return v;
}
__device__ __forceinline__ void radStackWrite(const RadStack& m, uint32_t ray, uint32_t slot, uint32_t val) {
  if (slot < kRadStack) m.lds[(ray * kRadStack + slot) * kBlockThreads] = val;
  else m.overflow[(size_t)((slot - kRadStack) * kRadRays + ray) * m.stride] = val;
}

template <bool STATS>
__global__ void __launch_bounds__(kBlockThreads, PTX_RAD_WAVES) radianceKernel(DeviceScene sc, FrameConstants fc, uint32_t* workCounter,
                                                                              void* __restrict__ outRaw, DeviceStats* stats,
                                                                              uint32_t* stackOverflow) {
  constexpr uint32_t K = kRadRays;
  __shared__ uint32_t sStack[K * kRadStack * kBlockThreads];
  RadStack stk;
  stk.lds = sStack + threadIdx.x;
  stk.overflow = stackOverflow + (blockIdx.x * kBlockThreads + threadIdx.x);
  stk.stride = gridDim.x * kBlockThreads;

  const uint32_t nPixels = fc.width * fc.height;
  const uint32_t totalWork = nPixels * fc.subStreams;  // checked on the host to fit 32 bits
  const vec3 camPos(fc.camPos[0], fc.camPos[1], fc.camPos[2]);
  float4* const out = reinterpret_cast<float4*>(outRaw);

  // ---- per-ray state (registers; every array is indexed by constants or through pick / put) ----
  uint32_t ref[K];    // what the node side of the ray does next (device_scene.hpp)
  uint32_t leaf[K];   // parked BLAS leaf (a leaf reference; 0 = none)
  uint32_t sp[K];
  vec3 o[K], d[K], inv[K];    // current ray (world or object space)
  float ht[K], hu[K], hv[K];  // closest hit so far: t (FLT_MAX = none), barycentrics
  uint32_t htri[K], hinst[K];
  vec3 wo[K], wd[K];          // world-space ray
  uint32_t curInst[K], tlasLeaf[K];
  bool zeroDir[K];            // a component of the current ray's direction is exactly 0 (0 * inf possible)
  bool pending[K];            // a traced ray's result waits for the shading phase
  bool havePixel[K];
  // shading-only state
  vec3 sum[K], col[K];
  uint32_t pix[K], rng[K], sample[K], bounce[K];
  uint32_t count[K];          // child boxes hit by the current segment (statistics only)
#pragma unroll
  for (uint32_t k = 0; k < K; k++) {
    ref[k] = kRefDone; leaf[k] = 0u; sp[k] = 1u;
    o[k] = vec3(0.0f); d[k] = vec3(0.0f, 0.0f, -1.0f); inv[k] = vec3(0.0f);
    ht[k] = FLT_MAX; hu[k] = 0.0f; hv[k] = 0.0f; htri[k] = 0xFFFFFFFFu; hinst[k] = 0xFFFFFFFFu;
    wo[k] = vec3(0.0f); wd[k] = vec3(0.0f, 0.0f, -1.0f);
    curInst[k] = 0u; tlasLeaf[k] = 0u;
    zeroDir[k] = false; pending[k] = false; havePixel[k] = false;
    sum[k] = vec3(0.0f); col[k] = vec3(1.0f);
    pix[k] = 0u; rng[k] = 0u; sample[k] = 0u; bounce[k] = 0u; count[k] = 0u;
  }

  uint32_t nRays = 0, nNodeSteps = 0, nTriTests = 0, nInstEntries = 0, nHits = 0, nMisses = 0;
  unsigned long long nBoxHits = 0;
#ifdef PTX_WAVE_PROFILE
  unsigned long long prof[16] = {};
#endif

  for (;;) {
    // ------------------------------------ census of the wave ------------------------------------
    PTX_MARK("R_CENSUS");
    uint32_t doneSel = K, nodeSel = K, triSel = K, enterSel = K;  // first ray of the lane that wants each kind (K = none)
    bool allIdle = true;
#pragma unroll
    for (uint32_t k = K; k-- > 0;) {
      doneSel = ref[k] == kRefDone ? k : doneSel;
      nodeSel = (int32_t)ref[k] >= 0 ? k : nodeSel;
      triSel = leaf[k] != 0u ? k : triSel;
      enterSel = ((ref[k] >> 30) == 3u && ref[k] < kRefIdle) ? k : enterSel;
      allIdle = allIdle && ref[k] == kRefIdle;
    }
    const unsigned long long mIdle = __ballot(allIdle);
    if (mIdle == ~0ull) break;
    const uint32_t nLive = 64u - (uint32_t)__popcll(mIdle);
    const uint32_t nDone = (uint32_t)__popcll(__ballot(doneSel < K));
    const uint32_t nNode = (uint32_t)__popcll(__ballot(nodeSel < K));
    const uint32_t nTri = (uint32_t)__popcll(__ballot(triSel < K));
    const uint32_t nEnter = (uint32_t)__popcll(__ballot(enterSel < K));
    PTX_RPROF(0, !allIdle);

    if (nDone >= PTX_RAD_SHADE_T || nDone >= nLive || (nDone != 0u && nNode == 0u && nTri == 0u && nEnter == 0u)) {
      // =========================== SHADE (postponed, wave-wide) ===========================
      PTX_RPROF(1, doneSel < K);
      PTX_MARK("R_SHADE");
      if (doneSel < K) {
        const uint32_t s = doneSel;
        vec3 sSum = pick(sum, s), sCol = pick(col, s), sWo = pick(wo, s), sWd = pick(wd, s);
        uint32_t sPix = pick(pix, s), sRng = pick(rng, s), sSample = pick(sample, s), sBounce = pick(bounce, s);
        bool sHave = pick(havePixel, s);
        bool pathEnded = true;
        if (pick(pending, s)) {
          Hit hit;
          hit.t = pick(ht, s); hit.u = pick(hu, s); hit.v = pick(hv, s); hit.tri = pick(htri, s); hit.inst = pick(hinst, s);
          if (STATS) {
            nRays++;
            nBoxHits += pick(count, s);
          }
          if (hit.t < FLT_MAX) {
            if (STATS) nHits++;
            shadeHit(sc, hit, sWo, sWd, sRng, sCol);  // writes the next ray into sWo / sWd
            sBounce++;
            pathEnded = sBounce >= fc.maxBounces;  // gpuRender.cu:29: the loop ends, col is returned
          } else {
            if (STATS) nMisses++;
            sCol *= environmentLookup(sc, fc, sWd);
          }
          if (pathEnded) {
            sSum += sCol;
            sSample--;
          }
        }
        bool live = true;
        if (pathEnded) {
          // ---- item finished? write it and pull the next one ----
          while (sSample == 0u && live) {
            if (sHave) {
              const uint32_t p = sPix & 0x00FFFFFFu, k = sPix >> 24;
              const uint32_t n = fc.sppBase + (k < fc.sppExtra ? 1u : 0u);
              out[(size_t)k * nPixels + p] = make_float4(sSum.x, sSum.y, sSum.z, (float)n);  // plane k: (sum, samples)
            }
            const uint32_t w = atomicAdd(workCounter, 1u);  // aggregated per wave by the compiler
            if (w >= totalWork) {
              live = false;
              sHave = false;
              break;
            }
            const uint32_t k = fc.subStreams > 1u ? w / nPixels : 0u;
            uint32_t x, y;
            workPixel(fc, w - k * nPixels, x, y);
            const uint32_t p = y * fc.width + x;
            sPix = p | (k << 24);
            sSum = vec3(0.0f);
            sRng = fc.seedBase + p + k * 0x9E3779B9u;  // gpuRender.cu:130 (+ stream offset)
            sHave = true;
            sSample = fc.sppBase + (k < fc.sppExtra ? 1u : 0u);
            if (fc.maxBounces == 0u) {
              sSum = vec3((float)sSample);  // no segment is traced: every path returns white (gpuRender.cu:27-29)
              sSample = 0u;
            }
          }
          if (live) {
            // ---- next camera ray: gpuRender.cu:132-143 ----
            const uint32_t p = sPix & 0x00FFFFFFu;
            const uint32_t py = p / fc.width, px = p - py * fc.width;
            float fx = (float)px, fy = (float)py;
            if (fc.jitter) {
              const vec2 xi = pt::getRandomVec2(sRng);
              fx += 0.5f * (2.0f * xi.x - 1.0f);
              fy += 0.5f * (2.0f * xi.y - 1.0f);
            }
            sWo = camPos;
            sWd = primaryDirection(fc, fx, fy);
            sCol = vec3(1.0f);
            sBounce = 0u;
          }
        }
        // ---- launch the segment: Scene::intersect starts at the TLAS root ----
        vec3 sInv;
        bool sZero;
        reciprocalDir(sWd, sInv, sZero);
        radStackWrite(stk, s, 0u, kRefDone);  // slot 0: bottom-of-stack marker
        put(sum, s, sSum); put(col, s, sCol); put(wo, s, sWo); put(wd, s, sWd);
        put(pix, s, sPix); put(rng, s, sRng); put(sample, s, sSample); put(bounce, s, sBounce);
        put(havePixel, s, sHave);
        put(pending, s, live);
        put(o, s, sWo); put(d, s, sWd); put(inv, s, sInv); put(zeroDir, s, sZero);
        put(ht, s, FLT_MAX); put(hu, s, 0.0f); put(hv, s, 0.0f); put(htri, s, 0xFFFFFFFFu); put(hinst, s, 0xFFFFFFFFu);
        put(count, s, 0u);
        put(tlasLeaf, s, 0u);
        put(sp, s, 1u);
        put(ref, s, live ? sc.tlasRootRefWide : kRefIdle);
      }
      continue;
    }

    // ------------------------------------ which kind of step ------------------------------------
    PTX_MARK("R_KIND");
    enum { kNone, kNode, kTri, kEnter };
    int kind;
    if (nEnter >= PTX_RAD_ENTER_T || (nEnter != 0u && nNode == 0u && nTri == 0u)) kind = kEnter;
    else if (nTri * PTX_RAD_TRI_W > nNode * PTX_RAD_NODE_W) kind = kTri;
    else if (nNode != 0u) kind = kNode;
    else kind = nTri != 0u ? kTri : kNone;

    // every ray's stack top, early: overlaps the record fetch; used by pops, exits and parking
    uint32_t top[K];
#pragma unroll
    for (uint32_t k = 0; k < K; k++) top[k] = radStackRead(stk, k, sp[k] ? sp[k] - 1u : 0u);  // sp is 0 once the bottom marker was popped
    uint32_t touched = K;  // the ray of this lane whose stack changed in this iteration: its `top` is stale

    if (kind == kNode) {
      PTX_RPROF(2, nodeSel < K);
      PTX_MARK("R_NODE");
      if (nodeSel < K) {
        const uint32_t s = nodeSel;
        const uint32_t r = pick(ref, s);
        const float4* rp = sc.arena + (sc.wide16 + r * 4u);
        const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
        const vec3 so = pick(o, s), sinv = pick(inv, s);
        const float st = pick(ht, s);
        const uint32_t ssp = pick(sp, s);
        const bool hasLeaf = pick(leaf, s) != 0u;
        uint32_t sr[4];
        uint32_t nHit;
        if (__any(pick(zeroDir, s))) nHit = wideStep<true>(so, sinv, st, r0, r1, r2, r3, sr);
        else nHit = wideStep<false>(so, sinv, st, r0, r1, r2, r3, sr);
        if (STATS) {
          nNodeSteps++;
          put(count, s, pick(count, s) + nHit);
#ifdef PTX_WAVE_PROFILE
          atomicAdd(&gSpHist[ssp < 63u ? ssp : 63u], 1ull);
#endif
        }
        // the nearest child is a BLAS leaf and no leaf is parked: park it, go on with the next child
        if (!hasLeaf && nHit != 0u && (sr[0] >> 30) == 2u) {
          put(leaf, s, sr[0]);
          sr[0] = sr[1]; sr[1] = sr[2]; sr[2] = sr[3];
          nHit--;
        }
        const bool doPop = nHit == 0u;
        const uint32_t nPush = doPop ? 0u : nHit - 1u;
        // sr[i] (i >= 1) goes to slot sp + nHit - 1 - i: farthest lowest, nearest on top; slots at or above the
        // new top receive values that are never read
        const uint32_t v0 = nPush == 3u ? sr[3] : (nPush == 2u ? sr[2] : sr[1]);
        const uint32_t v1 = nPush == 3u ? sr[2] : sr[1];
        radStackWrite(stk, s, ssp, v0);
        if (__any(nPush >= 2u)) {  // wave-uniform
          radStackWrite(stk, s, ssp + 1u, v1);
          radStackWrite(stk, s, ssp + 2u, sr[1]);
        }
        put(sp, s, ssp + nPush - (doPop ? 1u : 0u));
        put(ref, s, doPop ? pick(top, s) : sr[0]);
        touched = s;
      }
    } else if (kind == kTri) {
      PTX_RPROF(3, triSel < K);
      PTX_MARK("R_TRI");
      if (triSel < K) {
        // one triangle of the parked leaf per step (intersection.hpp:9-44, :112-122)
        const uint32_t s = triSel;
        const uint32_t lf = pick(leaf, s);
        const uint32_t first = lf & 0x007FFFFFu;
        const uint32_t left = (lf >> 23) & 0x7Fu;  // triangles after this one
        const float4* rp = sc.arena + (sc.tris16 + first * 3u);
        TriRec tri;
        tri.v0 = rp[0]; tri.e1 = rp[1]; tri.e2 = rp[2];
        const vec3 so = pick(o, s), sd = pick(d, s);
        const float cur = pick(ht, s);
        float u = 0.0f, v = 0.0f;
        const float dist = triangleTest(so, sd, tri, u, v);
        // closer, or as close and of smaller index: independent of the order in which leaves are visited
        if (dist < cur || (dist == cur && dist < FLT_MAX && first < pick(htri, s))) {
          // the instance is recorded with the hit (intersection.hpp:184-189 keeps the instance whose BLAS lowered t)
          put(ht, s, dist); put(hu, s, u); put(hv, s, v); put(htri, s, first); put(hinst, s, pick(curInst, s));
        }
        if (STATS) {
          nTriTests++;
#ifdef PTX_WAVE_PROFILE
          const uint32_t ssp = pick(sp, s);
          atomicAdd(&gSpHist[64u + (ssp < 63u ? ssp : 63u)], 1ull);
#endif
        }
        put(leaf, s, left == 0u ? 0u : makeLeafRef(first + 1u, left));
      }
    } else if (kind == kEnter) {
      PTX_RPROF(4, enterSel < K);
      PTX_MARK("R_ENTER");
      if (enterSel < K) {
        // TLAS leaf: enter its first instance, keep the rest of the leaf (gpuFunctions.hpp:32-41)
        const uint32_t s = enterSel;
        const uint32_t r = pick(ref, s);
        const uint32_t first = r & 0x00FFFFFFu;
        const uint32_t left = (r >> 24) & 0x3Fu;
        const float4* rp = sc.arena + (sc.inst16 + first * 4u);
        const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
        const vec3 sWo = pick(wo, s), sWd = pick(wd, s);
        const uint32_t ssp = pick(sp, s);
        const vec3 no = xformRows(r0, r1, r2, sWo, 1.0f);
        const vec3 nd = xformRows(r0, r1, r2, sWd, 0.0f);  // not normalised: t stays world-space
        vec3 ninv;
        bool nz;
        reciprocalDir(nd, ninv, nz);
        if (STATS) nInstEntries++;
        radStackWrite(stk, s, ssp, kRefInstExit);
        put(o, s, no); put(d, s, nd); put(inv, s, ninv); put(zeroDir, s, nz);
        put(tlasLeaf, s, left != 0u ? makeTlasLeafRef(first + 1u, left) : 0u);
        put(curInst, s, first);
        put(sp, s, ssp + 1u);
        put(ref, s, __float_as_uint(r3.y));  // InstXform::wideRootRef
        touched = s;
      }
    } else {
      PTX_RPROF(5, true);
    }

    // ---- rays that did not touch their stack: leave a finished instance, park a leaf that `ref` holds ----
    PTX_MARK("R_TAIL");
#pragma unroll
    for (uint32_t k = 0; k < K; k++) {
      const bool doExit = touched != k && ref[k] == kRefInstExit && leaf[k] == 0u;
      if (__any(doExit)) {
        PTX_RPROF(7, doExit);
        if (doExit) {
          o[k] = wo[k]; d[k] = wd[k];
          reciprocalDir(wd[k], inv[k], zeroDir[k]);
          const bool more = tlasLeaf[k] != 0u;
          ref[k] = more ? tlasLeaf[k] : top[k];
          sp[k] -= more ? 0u : 1u;
          tlasLeaf[k] = 0u;
        }
      }
      const bool doPark = touched != k && !doExit && leaf[k] == 0u && (ref[k] >> 30) == 2u;
      if (doPark) {
        leaf[k] = ref[k];
        ref[k] = top[k];
        sp[k]--;
      }
    }
  }

  PTX_MARK("R_END");
#ifdef PTX_WAVE_PROFILE
  if (STATS && (threadIdx.x & 63u) == 0u)
    for (int k = 0; k < 16; k++) atomicAdd(&gWaveProfile[k], prof[k]);
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
