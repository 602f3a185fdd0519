// C entry points of the host scene pipeline (include/pt_host.h).
#include "../../include/pt_host.h"

#include <cstring>
#include <string>

#include "colors.hpp"
#include "flatten.hpp"
#include "loaders.hpp"
#include "output.hpp"
#include "rng.hpp"
#include "sample_scenes.hpp"

using namespace pt;

struct pth_scene {
  Scene scene;
  std::vector<std::shared_ptr<Geometry>> geometryPool;
  std::vector<std::shared_ptr<Material>> materialPool;
  Camera camera;
  Image environment;
  SceneInfo info;
  FlatScene flat;
};

static thread_local std::string g_err;

extern "C" {

int pth_scene_create(int preset, const char* models_dir, const char* env_path, uint32_t scatter_instances,
                     uint32_t sky_w, uint32_t sky_h, pth_scene** out) {
  if (!out || preset < 0 || preset > 2) {
    g_err = "pth_scene_create: bad arguments";
    return -1;
  }
  g_quiet = true;
  SceneAssets assets;
  if (models_dir && *models_dir) assets.modelsDir = models_dir;
  assets.environmentPath = env_path ? env_path : "";
  assets.scatterInstances = scatter_instances;
  assets.skyWidth = sky_w;
  assets.skyHeight = sky_h;
  if (preset == 2 && !fileExists(assets.modelsDir + "/viking-room/viking_room.obj")) {
    g_err = "pth_scene_create: " + assets.modelsDir + "/viking-room/viking_room.obj not found";
    return -1;
  }
  auto* s = new pth_scene();
  getScene((SampleScene)preset, assets, s->scene, s->geometryPool, s->materialPool, s->camera, s->info);
  s->environment = getEnvironment(assets, s->info);
  flattenScene(s->scene, s->geometryPool, s->materialPool, s->environment, s->flat);
  *out = s;
  return 0;
}

const ptx_scene_desc* pth_scene_desc(const pth_scene* s) { return &s->flat.desc; }

void pth_scene_camera(const pth_scene* s, ptx_camera* out) { *out = toPtxCamera(s->camera); }

void pth_scene_info(const pth_scene* s, pth_info* o) {
  memset(o, 0, sizeof(*o));
  const Geometry& g = *s->geometryPool[0];
  o->triangles_per_mesh = (uint32_t)g.primitives.size();
  o->blas_nodes = (uint32_t)g.bvh.size();
  o->tlas_nodes = (uint32_t)s->scene.tlas.size();
  o->instances = (uint32_t)s->scene.meshes.size();
  o->blas_depth = g.bvhDepth;
  o->tlas_depth = s->scene.tlasDepth;
  o->blas_max_leaf = g.maxLeafSize;
  o->tlas_max_leaf = s->scene.maxLeafSize;
  o->stand_in_geometry = s->info.standInGeometry;
  o->stand_in_environment = s->info.standInEnvironment;
  o->blas_build_s = g.buildSeconds;
  o->tlas_build_s = s->scene.buildSeconds;
}

void pth_scene_destroy(pth_scene* s) { delete s; }

uint32_t pth_heatmap_colour(const uint32_t* counts, uint32_t n, float* rgb) {
  return heatmapColour(counts, n, rgb);
}

int pth_write_bmp(const char* path, const float* rgb, uint32_t w, uint32_t h) {
  return writeBmp(path, rgb, w, h) ? 0 : -1;
}

uint32_t pth_bmp_row_bytes(uint32_t w) { return bmpRowBytes(w); }
void pth_quantise_rows(const float* rgb, uint32_t w, uint32_t h, unsigned char* rows) { quantiseRows(rgb, w, h, rows); }
int pth_write_bmp_rows(const char* path, const unsigned char* rows, uint32_t w, uint32_t h) {
  return writeBmpRows(path, rows, w, h) ? 0 : -1;
}

uint32_t pth_pcg_hash(uint32_t* state) { return pcg_hash(*state); }
float pth_random_float(uint32_t* state) { return getRandomFloat(*state); }
int pth_parse_decimal(const char* text, double* out) { return parseDecimal(text, text + strlen(text), out) ? 0 : -1; }

static mat4 loadMat(const float m[16]) {
  mat4 r;
  for (int c = 0; c < 4; c++) r.c[c] = vec4(m[4 * c], m[4 * c + 1], m[4 * c + 2], m[4 * c + 3]);
  return r;
}
static void storeMat(const mat4& m, float o[16]) {
  for (int c = 0; c < 4; c++) { o[4 * c] = m.c[c].x; o[4 * c + 1] = m.c[c].y; o[4 * c + 2] = m.c[c].z; o[4 * c + 3] = m.c[c].w; }
}
void pth_mat4_inverse(const float m[16], float out[16]) { storeMat(inverse(loadMat(m)), out); }

void pth_mesh_matrices(const float centroid[3], const float rotation[3], float scale, const float translation[3],
                       float model[16], float inv_model[16]) {
  // A Mesh over a one-triangle geometry whose centroid is forced to the requested value.
  std::vector<Triangle> t(1);
  t[0].v0 = vec3(0, 0, 0); t[0].v1 = vec3(1, 0, 0); t[0].v2 = vec3(0, 1, 0);
  t[0].centroid = (t[0].v0 + t[0].v1 + t[0].v2) / 3.0f;
  auto g = std::make_shared<Geometry>(t);
  g->centroid = vec3(centroid[0], centroid[1], centroid[2]);
  Mesh m{g, std::make_shared<Material>()};
  m.rotation = vec3(rotation[0], rotation[1], rotation[2]);
  m.scale = scale;
  m.translation = vec3(translation[0], translation[1], translation[2]);
  m.update();
  storeMat(m.modelMatrix, model);
  storeMat(m.invModelMatrix, inv_model);
}

const char* pth_last_error(void) { return g_err.c_str(); }

}  // extern "C"
