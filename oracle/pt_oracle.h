/* pt_oracle.h -- entry points of the CPU oracle (TEST INFRASTRUCTURE ONLY; see pt_oracle.c). */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#include "../include/ptx_backend.h"

#ifdef __cplusplus
extern "C" {
#endif

/* seed modes: PTX_SEED_REFERENCE_GPU (0) and PTX_SEED_STREAM (1) as in ptx_backend.h, plus the
 * reference CPU path's own seeding: one state 1031 per worker thread carried across the pixels that
 * thread happens to draw from the shared atomic counter (main.cpp:186,191-193) -- deterministic
 * only with threads == 1. */
#define ORACLE_SEED_REFERENCE_CPU 100u

/* `-a`: main.cpp:213-218.  threads = worker threads sharing an atomic pixel counter. */
int oracle_render_counts(const ptx_scene_desc* sc, const ptx_camera* cam, const ptx_params* prm, int threads,
                         uint32_t* out_counts, ptx_stats* stats);
/* radiance: main.cpp:201-236 (out_rgb: width*height*3, row 0 = bottom). */
int oracle_render(const ptx_scene_desc* sc, const ptx_camera* cam, const ptx_params* prm, uint32_t seed_mode,
                  int threads, float* out_rgb, ptx_stats* stats);
/* un-normalised sums, 4 floats per pixel (r, g, b, samples): the multi-part building block. */
int oracle_accumulate(const ptx_scene_desc* sc, const ptx_camera* cam, const ptx_params* prm, uint32_t seed_mode,
                      int threads, float* out_accum4, ptx_stats* stats);
void oracle_resolve(const float* accum4, uint32_t n_pixels, float* out_rgb);

/* single-function probes */
float oracle_slab(const float* origin, const float* direction, float t, const float* bmin, const float* bmax);
float oracle_triangle(const float* origin, const float* direction, const float* tri9, float* bary2);
uint32_t oracle_pcg_hash(uint32_t* state);
float oracle_random_float(uint32_t* state);
void oracle_random_vec2(uint32_t* state, float* out2);
void oracle_primary_ray(const ptx_camera* cam, uint32_t width, uint32_t height, float fx, float fy, float* dir3);
void oracle_sample_cosine(const float* xi2, const float* n3, float* out3);
void oracle_sample_ggx(const float* xi2, const float* n3, float a, float* out3);
uint32_t oracle_environment_index(const ptx_scene_desc* sc, const float* dir3);
void oracle_tonemap(const float* sum3, float samples, float* out3);
uint32_t oracle_trace(const ptx_scene_desc* sc, const float* origin, const float* direction, float* t_out,
                      uint32_t* tri_out, float* bary2, uint32_t* count_out);

#ifdef __cplusplus
}
#endif
#endif
