/*
 * pt_host.h -- C entry points of the host scene pipeline (libpt_host.so), for callers that are not
 * C++ (the Python tests and bench.py load it with ctypes).  It wraps the host mirror of the
 * reference's scene code -- sampleScenes.cpp:9-143 (presets), geometry.cpp:11-67 / scene.cpp:16-41
 * (BLAS / TLAS build with the bvh.hpp builder), input.cpp:12-147 (loaders) -- and hands out the
 * flattened ptx_scene_desc that ptx_create (ptx_backend.h) consumes.  No GPU code is involved.
 */
#ifndef PT_HOST_H
#define PT_HOST_H

#include <stdint.h>

#include "ptx_backend.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pth_scene pth_scene;

typedef struct pth_info {
  uint32_t triangles_per_mesh;   /* README.md:56-60 "triangles per mesh" (geometry 0) */
  uint32_t blas_nodes;           /* README.md:56-60 "BLAS" (geometry 0) */
  uint32_t tlas_nodes;           /* README.md:56-60 "TLAS" */
  uint32_t instances;
  uint32_t blas_depth, tlas_depth;
  uint32_t blas_max_leaf, tlas_max_leaf;
  uint32_t stand_in_geometry;    /* 1: procedural bust instead of models/bust-of-menelaus.stl */
  uint32_t stand_in_environment; /* 1: analytic sky instead of the HDR */
  double blas_build_s, tlas_build_s;
} pth_info;

/* getScene(sampleScene, ...) + scene.completeScene() + flatten.  preset: 0 THREE_STL, 1 SCATTER_STL,
 * 2 OBJ (sampleScenes.hpp:6-8).  models_dir holds bust-of-menelaus.stl and viking-room/;
 * env_path is the Radiance HDR (either may be absent: deterministic stand-ins are used and
 * reported in pth_info).  scatter_instances: 0 = 10,000 as the reference.  Returns 0 / -1. */
int pth_scene_create(int preset, const char* models_dir, const char* env_path,
                     uint32_t scatter_instances, uint32_t sky_width, uint32_t sky_height,
                     pth_scene** out);
/* Writes the stand-in bust (the geometry used when models/bust-of-menelaus.stl is absent) as a binary STL
 * file, with one degenerate facet (a vertex named twice) inserted after every `degenerate_every` facets
 * (0 = none), so that the STL ingest path -- input.cpp:12-33 incl. stl_reader's removal of such facets --
 * can be exercised and timed on a file of the scan's size.  Returns the number of facets written or -1. */
long pth_write_standin_stl(const char* path, uint32_t degenerate_every);

/* The same, through an on-disk cache of the FLATTENED scene: the key is a hash of the
 * preset, its parameters and the CONTENT of every asset file the preset reads (or the stand-in
 * generator's version when a file is absent); a hit skips asset parsing, the BVH builds and the
 * flattening and hands back byte-identical arrays.  cache_dir is created if needed; *cache_hit
 * (may be NULL) tells what happened.  A cache file that is damaged or was written by another
 * format version is ignored and rewritten. */
int pth_scene_create_cached(int preset, const char* models_dir, const char* env_path,
                            uint32_t scatter_instances, uint32_t sky_width, uint32_t sky_height,
                            const char* cache_dir, int* cache_hit, pth_scene** out);
const ptx_scene_desc* pth_scene_desc(const pth_scene* s);
void pth_scene_camera(const pth_scene* s, ptx_camera* out);
void pth_scene_info(const pth_scene* s, pth_info* out);
void pth_scene_destroy(pth_scene* s);

/* main.cpp:363-374: max over counts, afmhot(count / max) for count > 0, black otherwise.
 * rgb is width*height*3 floats; returns the maximum. */
uint32_t pth_heatmap_colour(const uint32_t* counts, uint32_t n_pixels, float* rgb);

/* output.cpp:8-28: clamp, (uint8)(c*255.0), vertical flip, 24-bit BMP.  Returns 0 / -1. */
int pth_write_bmp(const char* path, const float* rgb, uint32_t width, uint32_t height);

/* The two halves of pth_write_bmp: the quantised, flipped, padded pixel array (what the device-side
 * output stage of ptx_backend.h returns ready-made) and the 54-byte container around it.
 * rows holds pth_bmp_row_bytes(width) * height bytes. */
uint32_t pth_bmp_row_bytes(uint32_t width);
void pth_quantise_rows(const float* rgb, uint32_t width, uint32_t height, unsigned char* rows);
int pth_write_bmp_rows(const char* path, const unsigned char* rows, uint32_t width, uint32_t height);

/* Small probes of host arithmetic for the tests (random.hpp:12-29, tiny_obj_loader.h:890-1027,
 * glm::inverse as used by mesh.cpp:26-29). */
uint32_t pth_pcg_hash(uint32_t* state);
float pth_random_float(uint32_t* state);
int pth_parse_decimal(const char* text, double* out);
void pth_mat4_inverse(const float m[16], float out[16]);
void pth_mesh_matrices(const float centroid[3], const float rotation[3], float scale,
                       const float translation[3], float model[16], float inv_model[16]);

const char* pth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
