/*
 * pt_oracle.c -- CPU restatement of the reference's render path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker for the HIP backend (path-tracer_amd/csrc): only tests/, the smoke test in
 * __graft_entry__.py and the cpu_baseline leg of bench.py may load it.  No product code path links,
 * loads or calls anything in oracle/.
 *
 * It restates, in plain C and function by function, the reference's CPU render path (`-d 1`):
 *   render loop ............ main.cpp:175-238        getIllumination ...... main.cpp:77-171
 *   getEnvironment ......... main.cpp:54-75          TLAS/BLAS traversal .. intersection.hpp:99-233
 *   slab test .............. intersection.hpp:48-75  Moeller-Trumbore ..... intersection.hpp:9-44
 *   instance transform ..... mesh.cpp:7-16           attribute fetch ...... geometry.cpp:69-87
 *   material / texture ..... material.cpp:13-25, image.hpp:27-35
 *   BRDF sampling .......... brdf.hpp:7-74           RNG .................. random.hpp:12-29
 *   camera ................. camera.hpp:8-20         tonemap .............. tonemapping.hpp:6-8
 * operating on the same flattened scene arrays (ptx_scene_desc, reference element layouts) that the
 * HIP backend receives, so both sides see bit-identical inputs.
 *
 * Vector arithmetic in the reference is GLM (un-vendored, version unpinned, absent here); the
 * operation orders below are GLM 0.9.9's scalar paths.  glm::sqrt/cos/sin/
 * tan/pow are `using std::...`, so with float arguments they are sqrtf/cosf/sinf/tanf/powf; M_PI and
 * the INV2PI/INVPI macros are double, so expressions that contain them are evaluated in double
 * exactly where C's usual arithmetic conversions say so.  Build: -O2/-O3, -ffp-contract=off, no
 * -ffast-math (the reference CPU path is plain g++ -O3 on x86-64: no FMA contraction).
 *
 * Pinning: the reference has no tests.  The only known answers are README.md:56-60 (scene 2: 3,828
 * triangles -> 4,899 BLAS nodes, 1 TLAS node, heat-map maximum 123), which tests/test_pins.py checks
 * through this oracle, and the two pictures of scene 2 under images/ (tests/test_reference_images.py: the `-a`
 * screen shot correlates 0.9993 with this oracle's heat map; the render agrees in structure, its light comes from
 * an HDR that the checkout lacks).  Radiance VALUES have no published numbers: against outputs of the reference
 * "parity unpinned" beyond that; against its formulas they are checked by the closed forms of
 * tests/test_oracle_analytic.py (Lambert and GGX lobes by quadrature, the mirror limit, environment addressing,
 * textures, instance transform).
 */
#define _GNU_SOURCE
#include <float.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/ptx_backend.h"
#include "pt_oracle.h"

/* Test build libpt_oracle_sharedlibm.so (-DPT_SHARED_LIBM): cosf, sinf, atan2f, acosf and powf come from the header that the
 * kernels' test build uses too, so that the radiance parity test can ask for identical bits (tests/shared_libm/shared_libm.h). */
#ifdef PT_SHARED_LIBM
#include "../tests/shared_libm/shared_libm.h"
float oracle_shared_libm_probe(int fn, float x, float y) {
  switch (fn) {
    case 0: return cosf(x);
    case 1: return sinf(x);
    case 2: return atan2f(x, y);
    case 3: return acosf(x);
    default: return powf(x, y);
  }
}
#endif

typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;

/* ---- GLM scalar semantics ---- */
static inline float gminf(float x, float y) { return (y < x) ? y : x; }
static inline float gmaxf(float x, float y) { return (x < y) ? y : x; }
static inline uint32_t gminu(uint32_t x, uint32_t y) { return (y < x) ? y : x; }
static inline float gclampf(float x, float lo, float hi) { return gminf(gmaxf(x, lo), hi); }

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 muls(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 smul(float s, v3 a) { return V3(s * a.x, s * a.y, s * a.z); }
static inline v3 divs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 sdiv(float s, v3 a) { return V3(s / a.x, s / a.y, s / a.z); }
static inline v3 neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross3(v3 x, v3 y) {
  return V3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline v3 normalize3(v3 v) { return muls(v, 1.0f / sqrtf(dot3(v, v))); }
static inline float dot_c(v3 a, v3 b) { return gmaxf(dot3(a, b), 1e-5f); } /* dataTypes.hpp:129-131 */

/* glm mat4 (column-major float[16]) * vec4: (m0*x + m1*y) + (m2*z + m3*w), xyz only */
static inline v3 mat4_mul_xyz(const float* m, v3 v, float w) {
  v3 r;
  r.x = (m[0] * v.x + m[4] * v.y) + (m[8] * v.z + m[12] * w);
  r.y = (m[1] * v.x + m[5] * v.y) + (m[9] * v.z + m[13] * w);
  r.z = (m[2] * v.x + m[6] * v.y) + (m[10] * v.z + m[14] * w);
  return r;
}

/* ---- RNG: random.hpp:12-29 ---- */
static inline uint32_t pcg_hash(uint32_t* seed) {
  *seed = *seed * 747796405u + 2891336453u;
  uint32_t state = *seed;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
static inline float random_float(uint32_t* s) { return (float)((double)pcg_hash(s) * (1.0 / 4294967296.0)); }
/* dvec2(pcg_hash(s), pcg_hash(s)): g++ (the reference CPU path's compiler) evaluates the arguments
 * right to left, so the first draw lands in y.  See path-tracer_amd/host/rng.hpp. */
static inline v2 random_vec2(uint32_t* s) {
  v2 r;
  r.y = random_float(s);
  r.x = random_float(s);
  return r;
}

/* ---- ray ---- */
typedef struct {
  v3 origin, direction, invDirection;
  float t;
} Ray;
typedef struct {
  v2 barycentric;
  uint32_t hitIndex;
} HitRecord;

typedef struct {
  uint64_t rays, node_steps, tri_tests, inst_entries, hits, misses, box_hits;
} Counters;

/* intersection.hpp:48-75 */
static inline float slab(const Ray* ray, const float* bmin, const float* bmax) {
  float t1x = (bmin[0] - ray->origin.x) * ray->invDirection.x;
  float t1y = (bmin[1] - ray->origin.y) * ray->invDirection.y;
  float t1z = (bmin[2] - ray->origin.z) * ray->invDirection.z;
  float t2x = (bmax[0] - ray->origin.x) * ray->invDirection.x;
  float t2y = (bmax[1] - ray->origin.y) * ray->invDirection.y;
  float t2z = (bmax[2] - ray->origin.z) * ray->invDirection.z;
  float tmin = gmaxf(gmaxf(gminf(t1x, t2x), gminf(t1y, t2y)), gminf(t1z, t2z));
  float tmax = gminf(gminf(gmaxf(t1x, t2x), gmaxf(t1y, t2y)), gmaxf(t1z, t2z));
  if (tmax >= tmin && tmin < ray->t && tmax > 0) return tmin;
  return FLT_MAX;
}

/* intersection.hpp:9-44 */
static inline float moller_trumbore(const Ray* ray, const ptx_triangle* tri, v2* bary) {
  v3 v0 = V3(tri->v0[0], tri->v0[1], tri->v0[2]);
  v3 edge1 = sub(V3(tri->v1[0], tri->v1[1], tri->v1[2]), v0);
  v3 edge2 = sub(V3(tri->v2[0], tri->v2[1], tri->v2[2]), v0);
  v3 h = cross3(ray->direction, edge2);
  float a = dot3(edge1, h);
  if (a > -1e-4f && a < 1e-4f) return FLT_MAX;
  float f = 1.0f / a;
  v3 s = sub(ray->origin, v0);
  float u = f * dot3(s, h);
  if (u < 0.0f || u > 1.0f) return FLT_MAX;
  v3 q = cross3(s, edge1);
  float v = f * dot3(ray->direction, q);
  if (v < 0.0f || u + v > 1.0f) return FLT_MAX;
  float t = f * dot3(edge2, q);
  if (t > 0.0f) {
    bary->x = u;
    bary->y = v;
    return t;
  }
  return FLT_MAX;
}

#define ORACLE_STACK 64 /* reference: 32, unchecked (intersection.hpp:97); the depth is checked here */

/* intersection.hpp:99-163 */
static void intersect_blas(Ray* ray, const ptx_bvh_node* bvh, const ptx_triangle* prims, HitRecord* hit,
                           uint32_t* count, Counters* c, int* overflow) {
  const ptx_bvh_node* node = &bvh[0];
  uint32_t stack[ORACLE_STACK];
  uint32_t sp = 0;
  for (;;) {
    if (node->count > 0) {
      for (uint32_t i = 0; i < node->count; i++) {
        uint32_t idx = node->left_first + i;
        v2 bary = {0, 0};
        float d = moller_trumbore(ray, &prims[idx], &bary);
        c->tri_tests++;
        if (d < ray->t) {
          ray->t = d;
          hit->barycentric = bary;
          hit->hitIndex = idx;
        }
      }
      if (sp == 0) break;
      node = &bvh[stack[--sp]];
      continue;
    }
    uint32_t idx1 = node->left_first, idx2 = idx1 + 1u;
    float d1 = slab(ray, bvh[idx1].aabb_min, bvh[idx1].aabb_max);
    float d2 = slab(ray, bvh[idx2].aabb_min, bvh[idx2].aabb_max);
    c->node_steps++;
    if (d1 > d2) {
      float td = d1; d1 = d2; d2 = td;
      uint32_t ti = idx1; idx1 = idx2; idx2 = ti;
    }
    if (d1 == FLT_MAX) {
      if (sp == 0) break;
      node = &bvh[stack[--sp]];
      continue;
    }
    node = &bvh[idx1];
    (*count)++;
    if (d2 != FLT_MAX) {
      (*count)++;
      if (sp >= ORACLE_STACK) { *overflow = 1; return; }
      stack[sp++] = idx2;
    }
  }
}

/* mesh.cpp:7-16 */
static void intersect_instance(const ptx_scene_desc* sc, uint32_t inst, Ray* ray, HitRecord* hit, uint32_t* count,
                               Counters* c, int* overflow) {
  const ptx_instance* m = &sc->instances[inst];
  const ptx_geometry* g = &sc->geometries[m->geometry_id];
  Ray tr = *ray;
  tr.origin = mat4_mul_xyz(m->inv_model, ray->origin, 1.0f);
  tr.direction = mat4_mul_xyz(m->inv_model, ray->direction, 0.0f); /* not normalised: t stays world-space */
  tr.invDirection = sdiv(1.0f, tr.direction);
  c->inst_entries++;
  intersect_blas(&tr, sc->blas_nodes + g->node_offset, sc->triangles + g->tri_offset, hit, count, c, overflow);
  ray->t = tr.t;
}

/* intersection.hpp:165-233; returns the instance index or UINT_MAX */
static uint32_t intersect_scene(const ptx_scene_desc* sc, Ray* ray, HitRecord* closestHit, uint32_t* count,
                                Counters* c, int* overflow) {
  const ptx_bvh_node* tlas = sc->tlas_nodes;
  const ptx_bvh_node* node = &tlas[0];
  uint32_t stack[ORACLE_STACK];
  uint32_t sp = 0;
  uint32_t meshIndex = UINT_MAX;
  HitRecord hitRecord = *closestHit;
  float closestDist = FLT_MAX;
  c->rays++;
  for (;;) {
    if (node->count > 0) {
      for (uint32_t i = 0; i < node->count; i++) {
        uint32_t idx = node->left_first + i;
        intersect_instance(sc, idx, ray, &hitRecord, count, c, overflow);
        if (ray->t < closestDist) {
          closestDist = ray->t;
          *closestHit = hitRecord;
          meshIndex = idx;
        }
      }
      if (sp == 0) break;
      node = &tlas[stack[--sp]];
      continue;
    }
    uint32_t idx1 = node->left_first, idx2 = idx1 + 1u;
    float d1 = slab(ray, tlas[idx1].aabb_min, tlas[idx1].aabb_max);
    float d2 = slab(ray, tlas[idx2].aabb_min, tlas[idx2].aabb_max);
    c->node_steps++;
    if (d1 > d2) {
      float td = d1; d1 = d2; d2 = td;
      uint32_t ti = idx1; idx1 = idx2; idx2 = ti;
    }
    if (d1 == FLT_MAX) {
      if (sp == 0) break;
      node = &tlas[stack[--sp]];
      continue;
    }
    node = &tlas[idx1];
    (*count)++;
    if (d2 != FLT_MAX) {
      (*count)++;
      if (sp >= ORACLE_STACK) { *overflow = 1; return meshIndex; }
      stack[sp++] = idx2;
    }
  }
  return meshIndex;
}

/* ---- camera: camera.hpp:8-20 ---- */
typedef struct {
  v3 c0, c1, c2; /* columns of mat3(xaxis, yaxis, -zaxis) */
  v3 position;
  float fov;
} CameraBasis;

static CameraBasis camera_basis(const ptx_camera* cam) {
  CameraBasis b;
  v3 pos = V3(cam->position[0], cam->position[1], cam->position[2]);
  v3 at = V3(cam->target[0], cam->target[1], cam->target[2]);
  v3 up = V3(cam->up[0], cam->up[1], cam->up[2]);
  v3 zaxis = normalize3(sub(at, pos));
  v3 xaxis = normalize3(cross3(zaxis, up));
  v3 yaxis = cross3(xaxis, zaxis);
  b.c0 = xaxis;
  b.c1 = yaxis;
  b.c2 = neg(zaxis);
  b.position = pos;
  b.fov = cam->fov_deg;
  return b;
}

static inline v3 ray_direction(float resx, float resy, float fov, float fx, float fy) {
  float x = fx - 0.5f * resx;
  float y = fy - 0.5f * resy;
  float z = (0.5f * resy) / tanf(0.5f * (fov * 0.01745329251994329576923690768489f));
  return normalize3(V3(x, y, -z));
}

static inline v3 mat3_mul(const CameraBasis* b, v3 v) { /* glm mat3 * vec3 */
  return V3(b->c0.x * v.x + b->c1.x * v.y + b->c2.x * v.z, b->c0.y * v.x + b->c1.y * v.y + b->c2.y * v.z,
            b->c0.z * v.x + b->c1.z * v.y + b->c2.z * v.z);
}

/* ---- environment: main.cpp:54-75 ---- */
#define INV2PI (1.0f / (2.0f * M_PI))
#define INVPI (1.0f / M_PI)

static inline v3 quat_rotate(v3 p, float qx, float qy, float qz, float qw) { /* main.cpp:54-56 */
  v3 q = V3(qx, qy, qz);
  v3 inner = add(muls(p, qw), cross3(q, p));
  return add(smul(2.0f, cross3(q, inner)), p);
}

static inline uint32_t environment_index(const ptx_scene_desc* sc, v3 direction) {
  float angle = (float)(-M_PI); /* rotateY(direction, -M_PI): float parameter */
  float qy = (float)sin(angle / 2.0), qw = (float)cos(angle / 2.0);
  v3 d = normalize3(quat_rotate(direction, 0.0f, qy, 0.0f, qw));
  uint32_t W = sc->env_width, H = sc->env_height;
  uint32_t u = (uint32_t)(W * (atan2f(d.z, d.x) * INV2PI + 0.5f));
  uint32_t v = (uint32_t)(H * acosf(d.y) * INVPI);
  return gminu(u + v * W, (W * H) - 1u);
}

static inline v3 environment_lookup(const ptx_scene_desc* sc, v3 direction) {
  uint32_t idx = environment_index(sc, direction);
  const float* e = sc->environment + 3ull * idx;
  return V3(e[0], e[1], e[2]);
}

/* ---- textures / materials: image.hpp:27-35, material.cpp:13-25 ---- */
static inline v3 texture_lookup(const ptx_scene_desc* sc, uint32_t tex, v2 uv) {
  const ptx_texture* t = &sc->textures[tex];
  float ux = uv.x - floorf(uv.x), uy = uv.y - floorf(uv.y); /* glm::fract */
  uint32_t cx = (uint32_t)(ux * (float)t->width);
  uint32_t cy = (uint32_t)(uy * (float)t->height);
  uint32_t idx = cy * t->width + cx;
  if (idx >= t->width * t->height) return V3(0, 0, 0);
  const float* p = sc->texels + 3ull * (t->texel_offset + idx);
  return V3(p[0], p[1], p[2]);
}

/* ---- attributes: geometry.cpp:69-87 ---- */
static inline v3 get_normal(const ptx_scene_desc* sc, const ptx_geometry* g, uint32_t idx, v2 b) {
  if (g->vtx_normal_offset == PTX_NONE) {
    const float* n = sc->face_normals + 3ull * (g->face_normal_offset + idx);
    return V3(n[0], n[1], n[2]);
  }
  const float* n = sc->vtx_normals + 3ull * (g->vtx_normal_offset + 3ull * idx);
  v3 n0 = V3(n[0], n[1], n[2]), n1 = V3(n[3], n[4], n[5]), n2 = V3(n[6], n[7], n[8]);
  return add(add(smul(b.x, n1), smul(b.y, n2)), smul(1.0f - (b.x + b.y), n0));
}
static inline v2 get_uv(const ptx_scene_desc* sc, const ptx_geometry* g, uint32_t idx, v2 b) {
  v2 r = {0, 0};
  if (g->vtx_uv_offset == PTX_NONE) return r;
  const float* t = sc->vtx_uvs + 2ull * (g->vtx_uv_offset + 3ull * idx);
  float w = 1.0f - (b.x + b.y);
  r.x = (b.x * t[2] + b.y * t[4]) + w * t[0];
  r.y = (b.x * t[3] + b.y * t[5]) + w * t[1];
  return r;
}

/* ---- BRDF: brdf.hpp:7-74 ---- */
static inline void pixar_onb(v3 n, v3* b1, v3* b2) {
  float sign_ = n.z >= 0.0f ? 1.0f : -1.0f;
  float a = -1.0f / (sign_ + n.z);
  float b = n.x * n.y * a;
  *b1 = V3(1.0f + sign_ * n.x * n.x * a, sign_ * b, -sign_ * n.x);
  *b2 = V3(b, sign_ + n.y * n.y * a, -n.y);
}
static inline v3 rotate_to_normal(v3 L, v3 N) {
  v3 t, b;
  pixar_onb(N, &t, &b);
  t = normalize3(t);
  b = normalize3(b);
  return normalize3(add(add(muls(t, L.x), muls(b, L.y)), muls(N, L.z)));
}
static inline v3 tangent_sample(float cosTheta, float xi_y) {
  float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
  float phi = (float)(xi_y * 2.0f * M_PI); /* float * float, then * double, narrowed into `float phi` */
  return normalize3(V3(cosf(phi) * sinTheta, sinf(phi) * sinTheta, cosTheta));
}
static inline v3 sample_cosine(v2 Xi, v3 N) {
  float cosTheta = sqrtf(1.0f - Xi.x);
  return rotate_to_normal(tangent_sample(cosTheta, Xi.y), N);
}
static inline v3 sample_ggx(v2 Xi, v3 N, float a) {
  float cosTheta = sqrtf((1.0f - Xi.x) / (1.0f + (a * a - 1.0f) * Xi.x));
  return rotate_to_normal(tangent_sample(cosTheta, Xi.y), N);
}
static inline float geometry_term(float cosTheta, float k) { return cosTheta / (cosTheta * (1.0f - k) + k); }
static inline float smiths(float NdotV, float NdotL, float roughness) {
  float k = roughness * 0.5f;
  return geometry_term(NdotV, k) * geometry_term(NdotL, k);
}
static inline v3 fresnel(float cosTheta, v3 F0) {
  float p = powf(1.0f - cosTheta, 5.0f);
  return add(F0, muls(sub(V3(1, 1, 1), F0), p));
}

/* ---- main.cpp:77-171 ---- */
static v3 get_illumination(const ptx_scene_desc* sc, Ray ray, uint32_t* rng, uint32_t bounceCount, uint32_t* testCount,
                           Counters* c, int* overflow) {
  v3 col = V3(1, 1, 1);
  for (uint32_t i = 0; i < bounceCount; i++) {
    HitRecord closestHit = {{0, 0}, UINT_MAX};
    uint32_t meshIdx = intersect_scene(sc, &ray, &closestHit, testCount, c, overflow);
    if (ray.t < FLT_MAX) {
      const ptx_instance* mesh = &sc->instances[meshIdx];
      const ptx_geometry* g = &sc->geometries[mesh->geometry_id];
      const ptx_material* mat = &sc->materials[mesh->material_id];
      c->hits++;

      v3 p = add(ray.origin, muls(ray.direction, ray.t));
      uint32_t local = closestHit.hitIndex; /* index into the geometry's own triangle array */
      v3 N = normalize3(mat4_mul_xyz(mesh->normal_matrix, get_normal(sc, g, local, closestHit.barycentric), 0.0f));
      if (dot3(ray.direction, N) > 0.0f) N = neg(N);
      v3 V = neg(ray.direction);

      const float metalness = mat->metalness;
      const float roughness = mat->roughness;
      v2 uv = get_uv(sc, g, local, closestHit.barycentric);
      v3 albedo = V3(mat->albedo[0], mat->albedo[1], mat->albedo[2]);
      if (mat->albedo_texture != PTX_NONE) albedo = mul(albedo, texture_lookup(sc, mat->albedo_texture, uv));
      v3 emissive = V3(mat->emissive[0], mat->emissive[1], mat->emissive[2]);
      if (mat->emissive_texture != PTX_NONE) emissive = mul(emissive, texture_lookup(sc, mat->emissive_texture, uv));

      /* glm::mix(F0, albedo, metalness) = F0*(1-a) + albedo*a */
      v3 F0m = V3(mat->f0[0], mat->f0[1], mat->f0[2]);
      v3 F0 = add(muls(F0m, 1.0f - metalness), muls(albedo, metalness));

      v3 localCol, sampleDir;
      if (metalness == 0.0f) {
        v2 Xi = random_vec2(rng);
        sampleDir = sample_cosine(Xi, N);
        localCol = albedo;
      } else {
        v2 Xi = random_vec2(rng);
        v3 H = sample_ggx(Xi, N, roughness);
        /* normalize(reflect(-V, H)); glm::reflect(I, N) = I - N * dot(N, I) * 2 */
        v3 I = neg(V);
        sampleDir = normalize3(sub(I, muls(muls(H, dot3(H, I)), 2.0f)));
        float NdotL = dot_c(N, sampleDir);
        float NdotV = dot_c(N, V);
        float NdotH = dot_c(N, H);
        float VdotH = dot_c(V, H);
        v3 F = fresnel(VdotH, F0);
        float G = smiths(NdotV, NdotL, roughness);
        localCol = divs(muls(muls(F, G), VdotH), NdotV * NdotH);
      }
      col = mul(col, add(localCol, emissive));
      ray.origin = add(p, smul(1e-4f, N));
      ray.direction = sampleDir;
      ray.invDirection = sdiv(1.0f, sampleDir);
      ray.t = FLT_MAX;
    } else {
      c->misses++;
      col = mul(col, environment_lookup(sc, ray.direction));
      break;
    }
  }
  return col;
}

/* ---- tonemap: main.cpp:225-235, tonemapping.hpp:6-8 ---- */
static inline float smoothstepf(float e0, float e1, float x) {
  float t = gclampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
  return t * t * (3.0f - 2.0f * t);
}
static inline float aces(float x) {
  return gclampf((x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f), 0.0f, 1.0f);
}
static inline float grade_channel(float c) {
  c *= smoothstepf(-0.75f, 1.45f, c);
  c = aces(0.275f * c);
  return powf(c, 1.0f / 2.2f);
}
static inline v3 resolve_pixel(v3 sum, float samples) {
  v3 col = divs(sum, samples);
  return V3(grade_channel(col.x), grade_channel(col.y), grade_channel(col.z));
}

/* ---- render: main.cpp:175-238 (CPU) with gpuRender.cu:118-170's seeding as an option ---- */
typedef struct {
  const ptx_scene_desc* sc;
  const ptx_camera* cam;
  const ptx_params* prm;
  int mode; /* 0 counts, 1 resolved radiance, 2 raw accumulation (4 floats/pixel) */
  uint32_t seed_mode;
  uint32_t* counts;
  float* out;
  atomic_uint* next;
  Counters totals;
  pthread_mutex_t* lock;
  int overflow;
} Job;

static inline uint32_t tile_of(uint32_t x, uint32_t y, uint32_t width) {
  uint32_t tiles_x = (width + 7u) / 8u;
  return (y / 8u) * tiles_x + (x / 8u);
}

static void* worker(void* arg) {
  Job* job = (Job*)arg;
  const ptx_params* P = job->prm;
  const uint32_t W = P->width, Hh = P->height, N = W * Hh;
  const CameraBasis basis = camera_basis(job->cam);
  Counters c;
  memset(&c, 0, sizeof(c));
  int overflow = 0;
  uint32_t threadRng = 1031u; /* main.cpp:186: one state per THREAD, carried across pixels */
  const float resx = (float)W, resy = (float)Hh;
  const uint32_t samples = job->mode == 0 ? 1u : P->spp;
  const uint32_t parts = P->part_count > 1u ? P->part_count : 1u;
  /* main.cpp:204 `samples > 1` speaks of the frame's sample count; a call that renders a share of a frame
   * (sample-slice split) says so in frame_spp (include/ptx_backend.h) */
  const int jitter = (P->frame_spp ? P->frame_spp : samples) > 1u;

  for (uint32_t idx = atomic_fetch_add_explicit(job->next, 1u, memory_order_relaxed); idx < N;
       idx = atomic_fetch_add_explicit(job->next, 1u, memory_order_relaxed)) {
    /* main.cpp:199 */
    float fx = (float)(idx % W), fy = floorf((float)idx / (float)W);
    uint32_t px = idx % W, py = idx / W;
    uint32_t rng;
    if (job->seed_mode == ORACLE_SEED_REFERENCE_CPU) rng = threadRng;
    else if (job->seed_mode == PTX_SEED_STREAM) rng = 1023u + idx + P->part_index * 0x9E3779B9u;
    else rng = 1023u + idx; /* gpuRender.cu:130 */

    if (job->mode == 0) {
      if (parts > 1u && (tile_of(px, py, W) % parts) != P->part_index) {
        job->counts[idx] = 0u;
        continue;
      }
      Ray ray;
      ray.origin = basis.position;
      ray.direction = normalize3(mat3_mul(&basis, ray_direction(resx, resy, basis.fov, fx, fy)));
      ray.invDirection = sdiv(1.0f, ray.direction);
      ray.t = FLT_MAX;
      uint32_t tests = 0;
      HitRecord h = {{0, 0}, UINT_MAX};
      intersect_scene(job->sc, &ray, &h, &tests, &c, &overflow);
      job->counts[idx] = tests;
      continue;
    }

    v3 col = V3(0, 0, 0);
    for (uint32_t s = 0; s < samples; s++) {
      float sx = fx, sy = fy;
      if (jitter) { /* main.cpp:204-207 */
        v2 xi = random_vec2(&rng);
        sx += 0.5f * (2.0f * xi.x - 1.0f);
        sy += 0.5f * (2.0f * xi.y - 1.0f);
      }
      Ray ray;
      ray.origin = basis.position;
      ray.direction = normalize3(mat3_mul(&basis, ray_direction(resx, resy, basis.fov, sx, sy)));
      ray.invDirection = sdiv(1.0f, ray.direction);
      ray.t = FLT_MAX;
      uint32_t tests = 0;
      col = add(col, get_illumination(job->sc, ray, &rng, P->max_bounces, &tests, &c, &overflow));
      c.box_hits += tests;
    }
    if (job->seed_mode == ORACLE_SEED_REFERENCE_CPU) threadRng = rng;
    if (job->mode == 1) {
      v3 o = resolve_pixel(col, (float)samples);
      job->out[3ull * idx] = o.x; job->out[3ull * idx + 1] = o.y; job->out[3ull * idx + 2] = o.z;
    } else {
      job->out[4ull * idx] = col.x; job->out[4ull * idx + 1] = col.y; job->out[4ull * idx + 2] = col.z;
      job->out[4ull * idx + 3] = (float)samples;
    }
  }
  pthread_mutex_lock(job->lock);
  job->totals.rays += c.rays; job->totals.node_steps += c.node_steps; job->totals.tri_tests += c.tri_tests;
  job->totals.inst_entries += c.inst_entries; job->totals.hits += c.hits; job->totals.misses += c.misses;
  job->totals.box_hits += c.box_hits;
  if (overflow) job->overflow = 1;
  pthread_mutex_unlock(job->lock);
  return NULL;
}

static int run(const ptx_scene_desc* sc, const ptx_camera* cam, const ptx_params* prm, int mode, uint32_t seed_mode,
               int threads, uint32_t* counts, float* out, ptx_stats* stats) {
  if (!sc || !cam || !prm || prm->width == 0 || prm->height == 0) return -1;
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  atomic_uint next;
  atomic_init(&next, 0u);
  pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
  Job job;
  memset(&job, 0, sizeof(job));
  job.sc = sc; job.cam = cam; job.prm = prm; job.mode = mode; job.seed_mode = seed_mode;
  job.counts = counts; job.out = out; job.next = &next; job.lock = &lock;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  pthread_t tid[256];
  for (int i = 1; i < threads; i++) pthread_create(&tid[i], NULL, worker, &job);
  worker(&job);
  for (int i = 1; i < threads; i++) pthread_join(tid[i], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (mode == 0) { /* counts: box_hits = sum of counts */
    uint64_t s = 0;
    for (uint64_t i = 0; i < (uint64_t)prm->width * prm->height; i++) s += counts[i];
    job.totals.box_hits = s;
  }
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->render_s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    stats->rays = job.totals.rays; stats->node_steps = job.totals.node_steps; stats->tri_tests = job.totals.tri_tests;
    stats->inst_entries = job.totals.inst_entries; stats->hits = job.totals.hits; stats->misses = job.totals.misses;
    stats->box_hits = job.totals.box_hits;
  }
  return job.overflow ? -3 : 0;
}

int oracle_render_counts(const ptx_scene_desc* sc, const ptx_camera* cam, const ptx_params* prm, int threads,
                         uint32_t* out_counts, ptx_stats* stats) {
  return run(sc, cam, prm, 0, 0u, threads, out_counts, NULL, stats);
}
int oracle_render(const ptx_scene_desc* sc, const ptx_camera* cam, const ptx_params* prm, uint32_t seed_mode,
                  int threads, float* out_rgb, ptx_stats* stats) {
  return run(sc, cam, prm, 1, seed_mode, threads, NULL, out_rgb, stats);
}
int oracle_accumulate(const ptx_scene_desc* sc, const ptx_camera* cam, const ptx_params* prm, uint32_t seed_mode,
                      int threads, float* out_accum4, ptx_stats* stats) {
  return run(sc, cam, prm, 2, seed_mode, threads, NULL, out_accum4, stats);
}
void oracle_resolve(const float* accum4, uint32_t n_pixels, float* out_rgb) {
  for (uint32_t i = 0; i < n_pixels; i++) {
    v3 o = resolve_pixel(V3(accum4[4ull * i], accum4[4ull * i + 1], accum4[4ull * i + 2]), accum4[4ull * i + 3]);
    out_rgb[3ull * i] = o.x; out_rgb[3ull * i + 1] = o.y; out_rgb[3ull * i + 2] = o.z;
  }
}

/* ---- probes of single functions, for unit tests and device-vs-host function parity ---- */
static Ray make_ray(const float* o, const float* d, float t) {
  Ray r;
  r.origin = V3(o[0], o[1], o[2]);
  r.direction = V3(d[0], d[1], d[2]);
  r.invDirection = sdiv(1.0f, r.direction);
  r.t = t;
  return r;
}
float oracle_slab(const float* origin, const float* direction, float t, const float* bmin, const float* bmax) {
  Ray r = make_ray(origin, direction, t);
  return slab(&r, bmin, bmax);
}
float oracle_triangle(const float* origin, const float* direction, const float* tri9, float* bary2) {
  Ray r = make_ray(origin, direction, FLT_MAX);
  ptx_triangle t;
  memcpy(&t, tri9, sizeof(t));
  v2 b = {0, 0};
  float d = moller_trumbore(&r, &t, &b);
  bary2[0] = b.x; bary2[1] = b.y;
  return d;
}
uint32_t oracle_pcg_hash(uint32_t* state) { return pcg_hash(state); }
float oracle_random_float(uint32_t* state) { return random_float(state); }
void oracle_random_vec2(uint32_t* state, float* out2) { v2 r = random_vec2(state); out2[0] = r.x; out2[1] = r.y; }
void oracle_primary_ray(const ptx_camera* cam, uint32_t width, uint32_t height, float fx, float fy, float* dir3) {
  CameraBasis b = camera_basis(cam);
  v3 d = normalize3(mat3_mul(&b, ray_direction((float)width, (float)height, b.fov, fx, fy)));
  dir3[0] = d.x; dir3[1] = d.y; dir3[2] = d.z;
}
void oracle_sample_cosine(const float* xi2, const float* n3, float* out3) {
  v2 xi = {xi2[0], xi2[1]};
  v3 r = sample_cosine(xi, V3(n3[0], n3[1], n3[2]));
  out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
void oracle_sample_ggx(const float* xi2, const float* n3, float a, float* out3) {
  v2 xi = {xi2[0], xi2[1]};
  v3 r = sample_ggx(xi, V3(n3[0], n3[1], n3[2]), a);
  out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
uint32_t oracle_environment_index(const ptx_scene_desc* sc, const float* dir3) {
  return environment_index(sc, V3(dir3[0], dir3[1], dir3[2]));
}
void oracle_tonemap(const float* sum3, float samples, float* out3) {
  v3 o = resolve_pixel(V3(sum3[0], sum3[1], sum3[2]), samples);
  out3[0] = o.x; out3[1] = o.y; out3[2] = o.z;
}
uint32_t oracle_trace(const ptx_scene_desc* sc, const float* origin, const float* direction, float* t_out,
                      uint32_t* tri_out, float* bary2, uint32_t* count_out) {
  Ray r = make_ray(origin, direction, FLT_MAX);
  HitRecord h = {{0, 0}, UINT_MAX};
  Counters c;
  memset(&c, 0, sizeof(c));
  int overflow = 0;
  uint32_t count = 0;
  uint32_t inst = intersect_scene(sc, &r, &h, &count, &c, &overflow);
  *t_out = r.t; *tri_out = h.hitIndex; bary2[0] = h.barycentric.x; bary2[1] = h.barycentric.y; *count_out = count;
  return inst;
}
