// Surface shading, BRDF importance sampling, environment lookup and tonemapping for one lane.
// Restates gpuRender.cu:14-116 / main.cpp:54-171 (getEnvironment, getIllumination's hit and miss
// branches), brdf.hpp:7-74, gpuFunctions.hpp:5-25,45-56, material.hpp:64-76, image.hpp:59-67 and
// tonemapping.hpp:6-8 on the flat DeviceScene.  Float types follow the reference's C++ expressions:
// M_PI / INV2PI / INVPI are double macros, so `phi`, the environment u/v scaling and nothing else are
// evaluated in double before narrowing.
#pragma once

#include "../host/ptmath.hpp"
#include "../host/rng.hpp"
#include "device_scene.hpp"
#include "trace.hpp"

// Test builds only (-DPTX_SHARED_LIBM=\"<path>/shared_libm.h\", Makefile target `sharedlibm`): the five libm calls below
// (cosf, sinf, atan2f, acosf, powf) come from a header that the radiance parity test also compiles into its build of the CPU
// checker, so that the two sides can be compared bit for bit (tests/test_gpu_shared_libm.py).  The shipped library calls the
// device library.
#ifdef PTX_SHARED_LIBM
#include PTX_SHARED_LIBM
#endif

namespace ptx {

using pt::vec2;
using pt::vec3;

__device__ __forceinline__ float dotClamped(const vec3& a, const vec3& b) { return pt::gmax(pt::dot(a, b), 1e-5f); }

// gpuFunctions.hpp:45-47
__device__ __forceinline__ vec3 quatRotate(const vec3& p, const vec3& q, float qw) {
  return 2.0f * pt::cross(q, p * qw + pt::cross(q, p)) + p;
}

// gpuRender.cu:14-22
__device__ __forceinline__ vec3 environmentLookup(const DeviceScene& sc, const FrameConstants& fc, const vec3& direction) {
  const vec3 sampleDir = pt::normalize(quatRotate(direction, vec3(0.0f, fc.envQy, 0.0f), fc.envQw));
  const double inv2pi = 1.0f / (2.0f * 3.14159265358979323846);
  const double invpi = 1.0f / 3.14159265358979323846;
  const uint32_t W = sc.envWidth, H = sc.envHeight;
  const uint32_t u = (uint32_t)(W * (atan2f(sampleDir.z, sampleDir.x) * inv2pi + 0.5f));
  const uint32_t v = (uint32_t)(sc.envHeightF * acosf(sampleDir.y) * invpi);  // (float)H * ...
  const uint32_t idx = pt::gmin(u + v * W, (W * H) - 1u);
  return f4xyz(sc.environment[idx]);
}

// image.hpp:59-67
__device__ __forceinline__ vec3 textureLookup(const DeviceScene& sc, uint32_t tex, const vec2& uv) {
  const TextureRec t = sc.textures[tex];
  const float ux = pt::fract(uv.x), uy = pt::fract(uv.y);
  const uint32_t cx = (uint32_t)(ux * (float)t.width);
  const uint32_t cy = (uint32_t)(uy * (float)t.height);
  const uint32_t idx = cy * t.width + cx;
  if (idx >= t.width * t.height) return vec3(0.0f);
  return f4xyz(sc.texels[t.texelBase + idx]);
}

// brdf.hpp:7-28
__device__ __forceinline__ vec3 rotateToNormal(const vec3& L, const vec3& N) {
  const float sign_ = N.z >= 0.0f ? 1.0f : -1.0f;
  const float a = -1.0f / (sign_ + N.z);
  const float b = N.x * N.y * a;
  vec3 tangent(1.0f + sign_ * N.x * N.x * a, sign_ * b, -sign_ * N.x);
  vec3 bitangent(b, sign_ + N.y * N.y * a, -N.y);
  tangent = pt::normalize(tangent);
  bitangent = pt::normalize(bitangent);
  return pt::normalize(tangent * L.x + bitangent * L.y + N * L.z);
}

// brdf.hpp:35-38 / :48-51
__device__ __forceinline__ vec3 tangentSample(float cosTheta, float xiY) {
  const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
  const float phi = (float)((double)(xiY * 2.0f) * 3.14159265358979323846);
  return pt::normalize(vec3(cosf(phi) * sinTheta, sinf(phi) * sinTheta, cosTheta));
}

__device__ __forceinline__ float geometryTerm(float cosTheta, float k) { return cosTheta / (cosTheta * (1.0f - k) + k); }

// One bounce of getIllumination's hit branch (gpuRender.cu:35-108).  Updates throughput `col`, writes
// the next ray.
__device__ __forceinline__ void shadeHit(const DeviceScene& sc, const Hit& hit, vec3& rayO, vec3& rayD, uint32_t& rng,
                                         vec3& col) {
  const InstShade is = sc.instShade[hit.inst];
  const GeomShade gs = sc.geomShade[is.geometryId];
  const MaterialRec mat = sc.materials[is.materialId];
  const uint32_t local = hit.tri - gs.triBase;
  const float bx = hit.u, by = hit.v;
  const float bw = 1.0f - (bx + by);

  // gpuFunctions.hpp:16-25
  vec3 n;
  if (gs.vtxNormalBase != 0xFFFFFFFFu) {
    const float4* vn = sc.vtxNormals + (gs.vtxNormalBase + 3u * local);
    n = bx * f4xyz(vn[1]) + by * f4xyz(vn[2]) + bw * f4xyz(vn[0]);
  } else {
    n = f4xyz(sc.faceNormals[gs.faceNormalBase + local]);
  }
  // gpuFunctions.hpp:5-14
  vec2 uv(0.0f, 0.0f);
  if (gs.vtxUvBase != 0xFFFFFFFFu) {
    const float2* t = sc.vtxUvs + (gs.vtxUvBase + 3u * local);
    const float2 t0 = t[0], t1 = t[1], t2 = t[2];
    uv = vec2((bx * t1.x + by * t2.x) + bw * t0.x, (bx * t1.y + by * t2.y) + bw * t0.y);
  }

  const vec3 p = rayO + rayD * hit.t;
  vec3 N = pt::normalize(xformRows(is.row0, is.row1, is.row2, n, 0.0f));
  if (pt::dot(rayD, N) > 0.0f) N = -N;
  const vec3 V = -rayD;

  const float metalness = mat.albedoMetal.w;
  const float roughness = mat.emissiveRough.w;
  vec3 albedo = f4xyz(mat.albedoMetal);
  if (mat.albedoTex != 0xFFFFFFFFu) albedo = albedo * textureLookup(sc, mat.albedoTex, uv);
  vec3 emissive = f4xyz(mat.emissiveRough);
  if (mat.emissiveTex != 0xFFFFFFFFu) emissive = emissive * textureLookup(sc, mat.emissiveTex, uv);

  vec3 localCol, sampleDir;
  const vec2 Xi = pt::getRandomVec2(rng);
  // Both branches of gpuRender.cu:56-105 turn (cosTheta, Xi.y) into a direction around N with the same code
  // (brdf.hpp:32-41 / :45-54): a wave that holds diffuse and metallic hits runs it once, not twice.
  const bool diffuse = metalness == 0.0f;
  const float a = roughness;
  const float cosTheta = diffuse ? sqrtf(1.0f - Xi.x) : sqrtf((1.0f - Xi.x) / (1.0f + (a * a - 1.0f) * Xi.x));
  const vec3 H = rotateToNormal(tangentSample(cosTheta, Xi.y), N);
  if (diffuse) {
    // Lambert: cosine-weighted direction, weight = albedo (gpuRender.cu:56-69)
    sampleDir = H;
    localCol = albedo;
  } else {
    // GGX half vector, reflect, weight = F*G*VdotH / (NdotV*NdotH) (gpuRender.cu:71-105, brdf.hpp:62-74)
    const vec3 F0 = pt::mix(f4xyz(mat.f0), albedo, metalness);
    sampleDir = pt::normalize(pt::reflect(-V, H));
    const float NdotL = dotClamped(N, sampleDir);
    const float NdotV = dotClamped(N, V);
    const float NdotH = dotClamped(N, H);
    const float VdotH = dotClamped(V, H);
    const vec3 F = F0 + (vec3(1.0f) - F0) * powf(1.0f - VdotH, 5.0f);
    const float k = roughness * 0.5f;
    const float G = geometryTerm(NdotV, k) * geometryTerm(NdotL, k);
    localCol = (F * G * VdotH) / (NdotV * NdotH);
  }
  col *= localCol + emissive;
  rayO = p + 1e-4f * N;
  rayD = sampleDir;
}

// gpuRender.cu:158-166 per channel
__device__ __forceinline__ float gradeChannel(float c) {
  c *= pt::smoothstep(-0.75f, 1.45f, c);
  const float x = 0.275f * c;
  c = pt::gclamp((x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f), 0.0f, 1.0f);
  return powf(c, 1.0f / 2.2f);
}
__device__ __forceinline__ vec3 resolvePixel(const vec3& sum, float samples) {
  const vec3 c = sum / samples;
  return vec3(gradeChannel(c.x), gradeChannel(c.y), gradeChannel(c.z));
}

// camera.hpp:8-20 + gpuRender.cu:139-143 with the per-frame parts hoisted into FrameConstants.
__device__ __forceinline__ vec3 primaryDirection(const FrameConstants& fc, float fx, float fy) {
  const vec3 dir = pt::normalize(vec3(fx - fc.halfResX, fy - fc.halfResY, fc.negZ));
  const vec3 w(fc.viewC0[0] * dir.x + fc.viewC1[0] * dir.y + fc.viewC2[0] * dir.z,
               fc.viewC0[1] * dir.x + fc.viewC1[1] * dir.y + fc.viewC2[1] * dir.z,
               fc.viewC0[2] * dir.x + fc.viewC1[2] * dir.y + fc.viewC2[2] * dir.z);
  return pt::normalize(w);
}

}  // namespace ptx
