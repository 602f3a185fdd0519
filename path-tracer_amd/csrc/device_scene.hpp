// GPU-side scene layout of the gfx950 backend (what ptx_create builds from ptx_scene_desc).
//
// The reference mirrors its C++ object graph on the device: scene -> GPUMesh[] (144 B) ->
// GPUGeometry* -> {BVHNode[] 32 B, GPUTriangle[] 36 B, normals, uvs}, GPUMaterial* -> GPUImage* ->
// texels (gpuRender.cu:188-262): four dependent pointer loads before the first node, 12-byte packed
// vec3 everywhere.  Here everything lives in ONE arena as flat arrays of 16-byte-aligned records,
// sized so one lane fetches a whole record with dwordx4 loads and records never straddle 64-B lines:
//
//   NodePair  64 B   the two children of one internal node (the reference tests bvh[idx1] and
//                    bvh[idx1+1] together at every step, intersection.hpp:133-136); one array holds the
//                    TLAS pairs followed by every BLAS's pairs
//   TriRec    48 B   v0, e1 = v1 - v0, e2 = v2 - v0 (the same IEEE subtractions intersection.hpp:11-12
//                    performs per test, done once on the host: bit-identical)
//   InstXform 64 B   rows of the inverse model matrix + the BLAS root reference
//   InstShade 64 B   rows of the normal matrix + geometry / material ids (touched only on a hit)
//   MaterialRec 64 B, GeomShade 16 B, float4 normals / texels / environment
//
// A node reference is one dword that also encodes what a lane has to do next:
//   0x00000000..0x7FFFFFFF  internal node: index of its NodePair
//   0x80000000..0xBFFFFFFF  BLAS leaf:  bits 29..23 = triangle count - 1, bits 22..0 = first triangle (absolute)
//   0xC0000000..0xFFFFFFFC  TLAS leaf:  bits 29..24 = instance count - 1, bits 23..0 = first instance
//   0xFFFFFFFD / E          lane idle / traversal done (the bottom-of-stack marker)
//   (0xFFFFFFFF: instance-exit marker of the experimental phase kernel only)
#pragma once

#include <stdint.h>

#include <hip/hip_runtime.h>

namespace ptx {

constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr uint32_t kTlasLeafFlag = 0xC0000000u;
constexpr uint32_t kRefIdle = 0xFFFFFFFDu;      // the lane has no more pixels
constexpr uint32_t kRefDone = 0xFFFFFFFEu;      // bottom-of-stack marker: traversal finished
constexpr uint32_t kRefInstExit = 0xFFFFFFFFu;  // pushed when a BLAS is entered: instance finished
constexpr uint32_t kMaxBlasLeafPrims = 128u, kMaxTlasLeafPrims = 64u;
constexpr uint32_t kMaxTriIndex = 0x007FFFFFu, kMaxInstIndex = 0x00FFFFF0u;
// Far children deferred on the stack never exceed the depth of the tree being walked; TLAS depth +
// BLAS depth + the two markers bound the stack (the reference uses two unchecked 32-entry stacks,
// intersection.hpp:97).  Entries beyond the LDS part go to a global overflow area sized from the bound.
constexpr uint32_t kStackEntries = 256u;

__host__ __device__ inline uint32_t makeLeafRef(uint32_t first, uint32_t count) {
  return kLeafFlag | ((count - 1u) << 23) | first;
}
__host__ __device__ inline uint32_t makeTlasLeafRef(uint32_t first, uint32_t count) {
  return kTlasLeafFlag | ((count - 1u) << 24) | first;
}

struct NodePair {   // 64 B
  float4 lmin;      // xyz = left child's aabb.min, w = bits of the left child's reference
  float4 lmax;      // xyz = left child's aabb.max
  float4 rmin;      // xyz = right child's aabb.min, w = bits of the right child's reference
  float4 rmax;      // xyz = right child's aabb.max
};

struct TriRec {     // 48 B
  float4 v0, e1, e2;
};

struct InstXform {  // 64 B
  float4 row0, row1, row2;  // row r = (m[0][r], m[1][r], m[2][r], m[3][r]) of glm's column-major inverse model
  uint32_t rootRef;         // BLAS root reference (binary layout: NodePair index or leaf)
  uint32_t wideRootRef;     // the same root in the wide layout (WideNode index or leaf)
  uint32_t pad[2];
};

struct InstShade {  // 64 B
  float4 row0, row1, row2;  // rows of the normal matrix
  uint32_t geometryId, materialId, pad[2];
};

struct GeomShade {  // 16 B: bases for the per-triangle attribute arrays (absolute triangle index - triBase = local)
  uint32_t triBase;
  uint32_t faceNormalBase;  // index into faceNormals
  uint32_t vtxNormalBase;   // index into vtxNormals (3 per triangle) or 0xFFFFFFFF
  uint32_t vtxUvBase;       // index into vtxUvs (3 per triangle) or 0xFFFFFFFF
};

struct MaterialRec {  // 64 B
  float4 albedoMetal;     // albedo.xyz, metalness
  float4 emissiveRough;   // emissive.xyz, roughness
  float4 f0;              // F0.xyz
  uint32_t albedoTex, emissiveTex, pad[2];
};

struct TextureRec {  // 16 B
  uint32_t width, height, texelBase, pad;
};

struct DeviceScene {
  const float4* arena;   // base of the one allocation; the three traversal arrays are also addressed
  uint32_t pairs16, tris16, inst16;  // from here, as offsets in 16-byte units (unified 64-byte fetch)
  uint32_t wide16;                   // WideNode array of the radiance path (wide_bvh.hpp)
  const NodePair* pairs;
  const TriRec* tris;
  const InstXform* instXform;
  const InstShade* instShade;
  const GeomShade* geomShade;
  const float4* faceNormals;
  const float4* vtxNormals;
  const float2* vtxUvs;
  const MaterialRec* materials;
  const TextureRec* textures;
  const float4* texels;
  const float4* environment;
  uint32_t tlasRootRef;
  uint32_t tlasRootRefWide;  // the TLAS root in the wide layout
  uint32_t tlasNodeEnd[2];   // internal-node references below this belong to the TLAS ([0] NodePair, [1] WideNode index)
  uint32_t envWidth, envHeight;
  float envHeightF;  // (float)envHeight, converted on the host: a uniform operand, not a per-lane register
};

// Per-frame constants, computed once on the host (the reference recomputes viewMatrix, tan(fov) and
// the environment rotation quaternion per sample per pixel: gpuRender.cu:140-141, camera.hpp:8-20,
// gpuFunctions.hpp:51-52).  tanf / sin / cos are evaluated by the host libm exactly as the reference's
// CPU path evaluates them, so the -a primary rays are bit-identical to the CPU path's.
struct FrameConstants {
  float camPos[3];
  float viewC0[3], viewC1[3], viewC2[3];  // columns of mat3(xaxis, yaxis, -zaxis)
  float halfResX, halfResY;               // 0.5f * resolution
  float negZ;                             // -(0.5f*res.y) / tanf(0.5f*radians(fov))
  float envQy, envQw;                     // quaternion (0, qy, 0, qw) of rotateY(dir, -M_PI)
  uint32_t width, height;
  uint32_t spp, maxBounces;
  uint32_t jitter;      // the FRAME has more than one sample per pixel (gpuRender.cu:134)
  uint32_t seedBase;    // 1023u (+ stream offset)
  uint32_t partIndex, partCount;
  uint32_t tilesX, tilesY;
  uint32_t subStreams;          // RNG streams per pixel in this launch (>= 1), each a separate work item
  uint32_t sppBase, sppExtra;   // stream k renders sppBase + (k < sppExtra) samples
  uint32_t widthMul, widthShift;  // n / width for n < 2^24: umulhi(n << 8, widthMul) >> widthShift
  uint32_t bandMul, bandShift;    // the same for n / (8 * width)
  double invPixels;               // 1.0 / (width * height)
  uint32_t scatterMul;            // poolKernel: item w renders pixel slot (w * scatterMul) mod pixels (0: slot w); see makeFrame
  uint32_t padTail;
};

struct DeviceStats {  // accumulated with atomics by the instrumented kernel variants
  unsigned long long rays, nodeSteps, triTests, instEntries, hits, misses, boxHits;
};

}  // namespace ptx
