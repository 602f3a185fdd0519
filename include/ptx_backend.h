/*
 * ptx_backend.h -- C ABI of the MI355X (gfx950) path-tracing backend.
 *
 * This is the drop-in boundary for the one device path of al-ro/path-tracer: the host function
 *
 *     void renderGPU(const Scene&, const std::vector<std::shared_ptr<Geometry>>&,
 *                    const std::vector<std::shared_ptr<Material>>&, const Camera&, Image&,
 *                    const Image& environment, uint samples, int maxBounces, bool renderBVH);
 *
 * declared at reference gpuRender.hpp:6-10, defined at gpuRender.cu:172-311 and called from exactly
 * one place, main.cpp:337.  The reference has no FFI layer; a maintainer replaces the body of
 * renderGPU with: flatten the C++ scene into the POD arrays of ptx_scene_desc, then
 * ptx_create -> ptx_render / ptx_render_counts -> ptx_destroy (INTEGRATION.md shows the binding;
 * path-tracer_amd/host/gpu_render.cpp is that binding for this repository's own host program).
 *
 * Conventions
 *  - Plain C, plain pointers and sizes.  No C++ or torch types cross this boundary.
 *  - All arrays in ptx_scene_desc are HOST pointers in the reference's own memory layout
 *    (dataTypes.hpp:26-99, packed 12-byte vec3, column-major 4x4 matrices).  The backend copies and
 *    re-lays them out for the GPU inside ptx_create; the caller may free them afterwards.
 *  - Every entry point returns 0 on success and a negative PTX_E* code on failure;
 *    ptx_last_error() returns a message for the calling thread.  The reference prints and exits on
 *    any CUDA error (error.hpp:10-26); the caller keeps that behaviour by checking the code.
 *  - Image row 0 is the BOTTOM scan line (gpuRender.cu:120-126 with output.cpp:15).
 *  - There is no CPU fallback: without a GPU / without the HIP library every call fails.
 */
#ifndef PTX_BACKEND_H
#define PTX_BACKEND_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTX_OK 0
#define PTX_EINVAL (-1)  /* bad argument / inconsistent scene description */
#define PTX_EHIP (-2)    /* a HIP runtime call failed (message has file:line) */
#define PTX_ELIMIT (-3)  /* scene exceeds a backend limit (traversal stack depth, 32-bit indices) */

/* dataTypes.hpp:72-78  BVHNode {AABB{vec3 min; vec3 max}; uint leftFirst; uint count}, 32 bytes.
 * count > 0: leaf, primitives [left_first, left_first+count); count == 0: children are nodes
 * left_first and left_first+1 (bvh.hpp:165-178). */
typedef struct ptx_bvh_node {
  float aabb_min[3];
  float aabb_max[3];
  uint32_t left_first;
  uint32_t count;
} ptx_bvh_node;

/* dataTypes.hpp:59-63  GPUTriangle {vec3 v0, v1, v2}, 36 bytes, in BLAS leaf order
 * (geometry.cpp:54). */
typedef struct ptx_triangle {
  float v0[3], v1[3], v2[3];
} ptx_triangle;

#define PTX_NONE 0xFFFFFFFFu

/* One Geometry (geometry.hpp / GPUGeometry, geometry.cpp:110-138): a BLAS plus per-triangle
 * attributes.  All offsets index the pooled arrays of ptx_scene_desc. */
typedef struct ptx_geometry {
  uint32_t node_offset;        /* BLAS root = blas_nodes[node_offset]; node indices inside the BLAS
                                  are relative to node_offset */
  uint32_t node_count;
  uint32_t tri_offset;         /* triangle i of this geometry = triangles[tri_offset + i] */
  uint32_t tri_count;
  uint32_t face_normal_offset; /* face_normals[face_normal_offset + i] (vec3), always present */
  uint32_t vtx_normal_offset;  /* vtx_normals[vtx_normal_offset + 3*i + c] (vec3) or PTX_NONE */
  uint32_t vtx_uv_offset;      /* vtx_uvs[vtx_uv_offset + 3*i + c] (vec2) or PTX_NONE */
  uint32_t reserved;
} ptx_geometry;

/* One Mesh instance (mesh.hpp:59-78 GPUMesh): inverse model matrix, normal matrix
 * (= transpose(inverse(model)), mesh.cpp:29-30), both column-major as glm::mat4, plus pool ids. */
typedef struct ptx_instance {
  float inv_model[16];
  float normal_matrix[16];
  uint32_t geometry_id;
  uint32_t material_id;
} ptx_instance;

/* material.hpp:35-56 GPUMaterial.  Texture ids index ptx_scene_desc.textures, PTX_NONE = none. */
typedef struct ptx_material {
  float albedo[3];
  float metalness;
  float roughness;
  float emissive[3];
  float f0[3];
  uint32_t albedo_texture;
  uint32_t emissive_texture;
} ptx_material;

/* image.hpp:38-67 GPUImage: nearest-texel RGB float image; texel (x,y) is
 * texels[texel_offset + y*width + x] (3 floats each). */
typedef struct ptx_texture {
  uint32_t width, height;
  uint64_t texel_offset;
} ptx_texture;

typedef struct ptx_scene_desc {
  const ptx_bvh_node* tlas_nodes;  uint32_t tlas_node_count;  /* scene.cpp:21-41 */
  const ptx_instance* instances;   uint32_t instance_count;   /* in TLAS leaf order (scene.cpp:36) */
  const ptx_geometry* geometries;  uint32_t geometry_count;
  const ptx_bvh_node* blas_nodes;  uint32_t blas_node_total;
  const ptx_triangle* triangles;   uint32_t triangle_total;
  const float* face_normals;       uint32_t face_normal_total; /* vec3 each */
  const float* vtx_normals;        uint32_t vtx_normal_total;  /* vec3 each, may be NULL/0 */
  const float* vtx_uvs;            uint32_t vtx_uv_total;      /* vec2 each, may be NULL/0 */
  const ptx_material* materials;   uint32_t material_count;
  const ptx_texture* textures;     uint32_t texture_count;
  const float* texels;             uint64_t texel_total;       /* vec3 each */
  const float* environment;        uint32_t env_width, env_height; /* vec3 each, main.cpp:33 */
} ptx_scene_desc;

/* dataTypes.hpp:101-108 Camera. */
typedef struct ptx_camera {
  float position[3];
  float target[3];
  float up[3];
  float fov_deg;
} ptx_camera;

/* seed_mode values */
#define PTX_SEED_REFERENCE_GPU 0u /* rngState = 1023 + pixel index, per pixel (gpuRender.cu:130) */
#define PTX_SEED_STREAM 1u        /* as above, decorrelated by part_index: sample-slice split (see
                                     DESIGN.md "multi-GPU") */

typedef struct ptx_params {
  uint32_t width, height;
  uint32_t spp;          /* samples per pixel rendered BY THIS CALL */
  uint32_t max_bounces;
  uint32_t seed_mode;
  uint32_t part_index;   /* this call's share of a partitioned frame: */
  uint32_t part_count;   /*   counts: 8x8 tiles with (tile % part_count) == part_index, other pixels
                              are written as 0;  radiance: RNG stream id (sample-slice split).
                              part_count <= 1 means the whole frame. */
  uint32_t sub_streams;  /* radiance, PTX_SEED_STREAM only: split this call's spp over this many RNG streams
                              per pixel (stream id = part_index * sub_streams + k, k-th stream gets
                              spp / sub_streams samples, the first spp % sub_streams one more), each a
                              separate work item; partial sums are added in stream order.  The reference
                              has one serial stream per pixel, i.e. W*H independent chains; a chip with
                              more lanes than that idles through the tail of every frame.  0 or 1 = one
                              stream (required for PTX_SEED_REFERENCE_GPU); at most 256.
                              ALL PARTS OF ONE FRAME MUST PASS THE SAME VALUE (derive it from the largest share): stream ids
                              are part_index * sub_streams + k, so parts that pass different values overlap and trace
                              the same random streams twice.  A part whose share is smaller than sub_streams simply runs
                              fewer streams. */
  uint32_t frame_spp;    /* samples per pixel of the WHOLE frame when this call renders a share of it (sample-slice
                              split: the sum of spp over all parts); 0 = this call is the whole frame (= spp).
                              The reference jitters the camera ray iff the frame has more than one sample
                              (gpuRender.cu:134 `samples > 1`): a share of one sample of an N-sample frame must still
                              be jittered, and a share of zero samples (more parts than samples; allowed only with
                              frame_spp > 0, radiance device entry points) yields an all-zero accumulation buffer. */
} ptx_params;

typedef struct ptx_stats {
  double upload_s;        /* ptx_create: host re-layout + H2D, seconds                        */
  double render_s;        /* kernel-only time of the last render call (HIP events, seconds)   */
  uint64_t rays;          /* calls of Scene::intersect (gpuRender.cu:32,149): one per segment */
  uint64_t node_steps;    /* internal-node steps (two child boxes fetched), TLAS + BLAS: S    */
  uint64_t tri_tests;     /* Moeller-Trumbore tests: T                                        */
  uint64_t inst_entries;  /* GPUMesh::intersect calls (instance transforms): I                */
  uint64_t hits;          /* shaded hits                                                      */
  uint64_t misses;        /* environment lookups                                              */
  uint64_t box_hits;      /* sum of the reference `count` (child boxes hit)                   */
} ptx_stats;

typedef struct ptx_scene ptx_scene; /* opaque handle */

/* Number of visible HIP devices (0 and an error code if the runtime is unusable). */
int ptx_device_count(int* count);

/* Replaces the upload half of renderGPU (gpuRender.cu:188-267: GPUImage/GPUGeometry/GPUMaterial/
 * GPUMesh/GPUScene constructors): validate, re-lay-out, one arena allocation, H2D copies onto
 * `device`.  Fails with PTX_ELIMIT if a BVH is deeper than the traversal stack. */
int ptx_create(const ptx_scene_desc* desc, int device, ptx_scene** out);

/* Replaces render<<<>>> + the D2H copy for renderBVH == true (gpuRender.cu:146-150, 277, 286):
 * one un-jittered primary ray per pixel, out_counts[y*width+x] = number of child boxes hit
 * (exact integers; the reference stores float(count) in all three channels). */
int ptx_render_counts(ptx_scene* scene, const ptx_camera* camera, const ptx_params* params,
                      uint32_t* out_counts, ptx_stats* stats /* may be NULL */);

/* Replaces render<<<>>> + the D2H copy for renderBVH == false (gpuRender.cu:132-169, 277, 286):
 * spp paths per pixel, averaged, colour-graded, ACES tonemapped, gamma corrected.
 * out_rgb is width*height*3 floats. */
int ptx_render(ptx_scene* scene, const ptx_camera* camera, const ptx_params* params,
               float* out_rgb, ptx_stats* stats /* may be NULL */);

/* Device-side output stage: what main() and output.cpp do to the rendered image before the BMP
 * container is written, done on the GPU so that 3 bytes per pixel come back instead of 12 (or 4).
 *   ptx_render_counts_bgr8 : ptx_render_counts, then the heat-map colouring of main.cpp:363-374 (maximum
 *                            over the frame, afmhot(count / max), black where count == 0) and the
 *                            quantisation of output.cpp:13-23 (clamp to [0,1], (unsigned char)(c * 255.0));
 *                            *max_count receives the frame's maximum ("Maximum BVH tests").
 *   ptx_render_bgr8        : ptx_render, then the same quantisation.
 * out_rows is the pixel array of a 24-bit BMP: image row 0 (the bottom scan line, as the kernel's
 * rows are ordered) first, 3 bytes per pixel in B, G, R order, every row padded with zeros to row_bytes
 * (>= 3 * width; a BMP wants a multiple of 4).  Bit-identical to the host path on the same image. */
int ptx_render_counts_bgr8(ptx_scene* scene, const ptx_camera* camera, const ptx_params* params,
                           uint8_t* out_rows, uint32_t row_bytes, uint32_t* max_count, ptx_stats* stats /* may be NULL */);
int ptx_render_bgr8(ptx_scene* scene, const ptx_camera* camera, const ptx_params* params,
                    uint8_t* out_rows, uint32_t row_bytes, ptx_stats* stats /* may be NULL */);

/* Multi-GPU building blocks (no counterpart in the reference, which is single-GPU; BASELINE.json
 * asks for a sample-slice split with an RCCL reduce of float accumulation buffers).
 * d_accum / d_out are DEVICE pointers on the scene's device (e.g. a torch tensor's data_ptr):
 *   ptx_accumulate_device : d_accum[4*pixel + c] = un-normalised sum over this call's spp samples
 *                           (c = 0..2; channel 3 = number of samples), launched on `hip_stream`
 *                           (a hipStream_t, NULL = default stream); asynchronous.
 *   ptx_resolve_device    : d_out_rgb[3*pixel + c] = tonemap(accum / accum[3]) -- the tail of
 *                           gpuRender.cu:158-169 -- after the caller has summed the buffers of all
 *                           parts (RCCL reduce).
 *   ptx_counts_device     : heat-map counts for this part's tiles into d_counts (uint32 per pixel,
 *                           0 outside the part), so a sum-reduce reassembles the frame exactly.
 * A handle owns one set of per-launch scratch (work counter, stack overflow area, stream planes, timing
 * events): at most ONE launch per handle may be in flight.  Launches through one handle on one stream
 * are ordered and fine; for concurrent launches on several streams create one handle per stream. */
int ptx_accumulate_device(ptx_scene* scene, const ptx_camera* camera, const ptx_params* params,
                          float* d_accum, void* hip_stream);
int ptx_resolve_device(ptx_scene* scene, const float* d_accum, uint32_t width, uint32_t height,
                       float* d_out_rgb, void* hip_stream);
int ptx_counts_device(ptx_scene* scene, const ptx_camera* camera, const ptx_params* params,
                      uint32_t* d_counts, void* hip_stream);

/* One process, several GPUs (no counterpart in the reference; SURVEY.md section 5: "-g <ngpus>"):
 * scenes[i] is a handle that ptx_create made on the i-th device to use (the scene is replicated).
 *   ptx_render_multi        : `params` describes the WHOLE frame (spp = samples per pixel of the frame).  Part i
 *                             traces spp / count (+1 for the first spp % count) samples of every pixel with RNG
 *                             stream set i (sample-slice split; count > 1 requires PTX_SEED_STREAM) on its own
 *                             stream, all devices at once; the W*H*4 f32 accumulation buffers are copied
 *                             device-to-device (hipMemcpyPeerAsync, xGMI) into the first handle's device, added
 *                             there in part order, divided by the summed sample count, graded and tonemapped.
 *                             Same result as N processes with one RCCL reduce (multigpu.py); count = 1 equals
 *                             ptx_render.
 *   ptx_render_counts_multi : heat map, 8x8-tile partition (tile % count == i), integer sum on the first device:
 *                             bit-identical to ptx_render_counts.
 * stats->render_s is the wall time of the whole job (slowest part + gather + resolve + copy back). */
int ptx_render_multi(ptx_scene* const* scenes, uint32_t count, const ptx_camera* camera, const ptx_params* params,
                     float* out_rgb, ptx_stats* stats /* may be NULL */);
int ptx_render_counts_multi(ptx_scene* const* scenes, uint32_t count, const ptx_camera* camera, const ptx_params* params,
                            uint32_t* out_counts, ptx_stats* stats /* may be NULL */);

/* ptx_accumulate_device for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE (renderGPU
 * allocates everything per call, gpuRender.cu:188-262): this is for code under HIP graph capture, tight
 * serving loops and harnesses that forbid hipMalloc.  Same arguments, same kernels and bit-identical
 * output, but the device memory the launch needs beyond the scene, the input and d_accum -- the work
 * counter, the traversal-stack overflow area, and (sub_streams > 1) the per-stream accumulation planes
 * -- comes from ONE static array of PTX_NOALLOC_SCRATCH_BYTES that the library holds on every GPU it
 * runs on until the process ends; the call makes no hipMalloc / hipFree.  A launch that needs more
 * (planes = sub_streams * width * height * 16 bytes; the stack overflow and park area is at most 123 MB for the
 * preset scenes: (stack bound 84 - 23 entries in LDS + 17 park slots) * 4 bytes per lane of the 393,216-lane
 * persistent grid, scene 1 on the wide layout)
 * fails with PTX_ELIMIT and a message that names the numbers.  All calls on one GPU share that GPU's
 * array, whatever the scene handle: calls that overlap on one GPU (two streams, two threads) race;
 * calls ordered on one stream do not.  Requires a handle from ptx_create on the same device (that is
 * where the array's device address is resolved). */
#define PTX_NOALLOC_SCRATCH_BYTES (256u << 20)
int ptx_accumulate_device_noalloc(ptx_scene* scene, const ptx_camera* camera, const ptx_params* params,
                                  float* d_accum, void* hip_stream);

/* Statistics of the most recent render on this scene, gathered by an instrumented (slower) variant
 * of the kernels: call with the same camera/params as the render it describes; paths are
 * deterministic in (pixel, seed), so the numbers apply to the un-instrumented launch exactly. */
int ptx_measure(ptx_scene* scene, const ptx_camera* camera, const ptx_params* params,
                int heatmap, ptx_stats* stats);

/* Kernel-only time of the most recent launch made through this handle, seconds (HIP events on the
 * launch stream); synchronises that stream. */
int ptx_last_kernel_seconds(ptx_scene* scene, double* seconds);

/* Self-checks of two things the radiance path relies on (no counterpart in the reference).
 *   ptx_check_wide_layout  : HOST ONLY, needs no GPU.  Builds the four-wide quantised re-layout of the
 *                            scene's BVHs (what ptx_create builds for radiance calls) and verifies that
 *                            every child's quantised box covers its exact box with the device's own
 *                            arithmetic, and that every leaf of every tree is referenced exactly once.
 *                            report[0] = wide nodes, [1] = child slots checked, [2] = slots whose
 *                            quantised box does NOT cover (must be 0), [3] = leaves missing or
 *                            duplicated (must be 0), [4] = deepest wide tree, [5] = binary nodes folded.
 *   ptx_selftest_reciprocal: on `device`, compares the 3-instruction reciprocal used for ray
 *                            directions with the IEEE 1.0f / x for EVERY float whose biased exponent is
 *                            in [1, 252] (the guarded range); *mismatches must come back 0. */
int ptx_check_wide_layout(const ptx_scene_desc* desc, uint64_t report[6]);
/* Self-check of the tight TLAS boxes of the radiance path (csrc/tight_bounds.hpp; no counterpart in the reference, which
 * bounds an instance by the transformed corners of its BLAS box, mesh.cpp:18-41).  Computes the instance bounds on
 * `device` as ptx_create does and reports: [0] instances, [1] instances whose bounds were recomputed on the host,
 * [2] of those, how many differ from the device's in any bit -- or, when the scene is so large that BLAS box corners
 *     stand in for the vertices, are not contained in the device's (must be 0), [3] TLAS nodes whose tight box is not
 * inside the reference's box or whose links changed (must be 0), [4] leaf boxes that do not hold the padded bounds
 * of one of their instances, plus recomputed instances whose padded box does not hold the bounds of their vertices taken in
 * double through the exact inverse of inv_model -- the points the device's object-space test stands for (must be 0), [5] mean leaf volume, tight / reference, in millionths.  device < 0: no GPU
 * is touched, the bounds come from the host loop (small scenes: CPU tests of the refit). */
int ptx_check_tight_tlas(const ptx_scene_desc* desc, int device, uint64_t report[6]);
int ptx_selftest_reciprocal(int device, uint64_t* mismatches);

/* Replaces the frees at gpuRender.cu:289-310. */
void ptx_destroy(ptx_scene* scene);

const char* ptx_last_error(void);

/* Backend identification string: "ptx_backend <version> gfx950 HIP <runtime>" */
const char* ptx_version(void);

/* Hash of the sources and compiler flags this library was built from (tools/source_hash.py; "unstamped" for a build that
 * did not pass it).  No counterpart in the reference: measurements kept beside the code (profiles/traffic.json) name the
 * build they were taken on, and bench.py reports them only for that build. */
const char* ptx_build_id(void);

/* PTX_NOALLOC_SCRATCH_BYTES of the library that is loaded (callers that size requests from a binding, not from this header). */
uint64_t ptx_noalloc_scratch_bytes(void);

#ifdef __cplusplus
}
#endif
#endif /* PTX_BACKEND_H */
