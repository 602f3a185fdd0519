// Kernels of the gfx950 backend.  One wave64 owns one 8x8 pixel tile (lane l -> x = l & 7, y = l >> 3),
// so the 64 primary rays of a wave are spatially adjacent; tiles are handed to workgroups so that the
// tiles an XCD works on are neighbours (shared BVH nodes stay in that XCD's L2).
#pragma once

#include "device_scene.hpp"
#include "shade.hpp"
#include "trace.hpp"

namespace ptx {

// Workgroups are dealt round-robin to the 8 XCDs (blocks b and b+8 share one): give XCD x the x-th
// contiguous band of tiles.  Bijective for any tile count.
__device__ __forceinline__ uint32_t xcdBandedTile(uint32_t block, uint32_t nTiles) {
  const uint32_t xcd = block & 7u, slot = block >> 3;
  const uint32_t q = nTiles >> 3, r = nTiles & 7u;
  const uint32_t base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
  return base + slot;
}

__device__ __forceinline__ bool tilePixel(const FrameConstants& fc, uint32_t tile, uint32_t lane, uint32_t& x, uint32_t& y) {
  const uint32_t ty = tile / fc.tilesX, tx = tile - ty * fc.tilesX;
  x = tx * 8u + (lane & 7u);
  y = ty * 8u + (lane >> 3);
  return x < fc.width && y < fc.height;
}

template <bool STATS>
__device__ __forceinline__ void flushStats(DeviceStats* stats, uint64_t rays, const TraceCounters& tc, uint64_t hits,
                                           uint64_t misses, uint64_t boxHits) {
  if (!STATS) return;
  atomicAdd(&stats->rays, (unsigned long long)rays);
  atomicAdd(&stats->nodeSteps, (unsigned long long)tc.nodeSteps);
  atomicAdd(&stats->triTests, (unsigned long long)tc.triTests);
  atomicAdd(&stats->instEntries, (unsigned long long)tc.instEntries);
  atomicAdd(&stats->hits, (unsigned long long)hits);
  atomicAdd(&stats->misses, (unsigned long long)misses);
  atomicAdd(&stats->boxHits, (unsigned long long)boxHits);
}

// `-a` heat map: gpuRender.cu:146-150.  One un-jittered primary ray per pixel; the number of child
// boxes hit is written as an exact integer.
template <bool STATS>
__global__ void __launch_bounds__(64) countsKernel(DeviceScene sc, FrameConstants fc, uint32_t* __restrict__ out,
                                                   DeviceStats* stats) {
  const uint32_t nTiles = fc.tilesX * fc.tilesY;
  const uint32_t tile = xcdBandedTile(blockIdx.x, nTiles);
  uint32_t x, y;
  if (!tilePixel(fc, tile, threadIdx.x, x, y)) return;
  const uint32_t idx = y * fc.width + x;
  if (fc.partCount > 1u && (tile % fc.partCount) != fc.partIndex) {
    out[idx] = 0u;
    return;
  }
  const vec3 o(fc.camPos[0], fc.camPos[1], fc.camPos[2]);
  const vec3 d = primaryDirection(fc, (float)x, (float)y);
  const vec3 inv = 1.0f / d;
  uint32_t count = 0;
  TraceCounters tc = {0, 0, 0};
  traceRay<STATS>(sc, o, d, inv, count, tc);
  out[idx] = count;
  flushStats<STATS>(stats, 1, tc, 0, 0, count);
}

// Radiance: gpuRender.cu:118-170 + getIllumination :24-116.  RESOLVE = true writes the tonemapped
// pixel (3 floats); false writes the un-normalised sum and the sample count (4 floats) for a
// multi-GPU reduce.
template <bool STATS, bool RESOLVE>
__global__ void __launch_bounds__(64) radianceKernel(DeviceScene sc, FrameConstants fc, float* __restrict__ out,
                                                     DeviceStats* stats) {
  const uint32_t nTiles = fc.tilesX * fc.tilesY;
  const uint32_t tile = xcdBandedTile(blockIdx.x, nTiles);
  uint32_t x, y;
  if (!tilePixel(fc, tile, threadIdx.x, x, y)) return;
  const uint32_t idx = y * fc.width + x;

  uint32_t rng = fc.seedBase + idx;  // gpuRender.cu:130
  const vec3 camPos(fc.camPos[0], fc.camPos[1], fc.camPos[2]);
  vec3 sum(0.0f);
  TraceCounters tc = {0, 0, 0};
  uint64_t rays = 0, hits = 0, misses = 0, boxHits = 0;

  for (uint32_t s = 0; s < fc.spp; s++) {
    float fx = (float)x, fy = (float)y;
    if (fc.spp > 1u) {  // gpuRender.cu:133-137
      const vec2 xi = pt::getRandomVec2(rng);
      fx += 0.5f * (2.0f * xi.x - 1.0f);
      fy += 0.5f * (2.0f * xi.y - 1.0f);
    }
    vec3 o = camPos;
    vec3 d = primaryDirection(fc, fx, fy);
    vec3 col(1.0f);
    for (uint32_t b = 0; b < fc.maxBounces; b++) {
      const vec3 inv = 1.0f / d;
      uint32_t count = 0;
      const Hit hit = traceRay<STATS>(sc, o, d, inv, count, tc);
      if (STATS) {
        rays++;
        boxHits += count;
      }
      if (hit.t < FLT_MAX) {
        if (STATS) hits++;
        shadeHit(sc, hit, o, d, rng, col);
      } else {
        if (STATS) misses++;
        col *= environmentLookup(sc, fc, d);
        break;
      }
    }
    sum += col;
  }

  if (RESOLVE) {
    const vec3 c = resolvePixel(sum, (float)fc.spp);
    out[3ull * idx + 0] = c.x;
    out[3ull * idx + 1] = c.y;
    out[3ull * idx + 2] = c.z;
  } else {
    reinterpret_cast<float4*>(out)[idx] = make_float4(sum.x, sum.y, sum.z, (float)fc.spp);
  }
  flushStats<STATS>(stats, rays, tc, hits, misses, boxHits);
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
