// C entry points of the host scene pipeline (include/pt_host.h).
#include "../../include/pt_host.h"

#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
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
  // a scene that came out of the cache has only the flattened arrays
  bool fromCache = false;
  ptx_camera cachedCamera{};
  pth_info cachedInfo{};
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

void pth_scene_camera(const pth_scene* s, ptx_camera* out) { *out = s->fromCache ? s->cachedCamera : toPtxCamera(s->camera); }

void pth_scene_info(const pth_scene* s, pth_info* o) {
  if (s->fromCache) {
    *o = s->cachedInfo;
    return;
  }
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

}  // extern "C"

// ---- flattened-scene cache -------------------------------------------------------------------
namespace {
constexpr char kCacheMagic[8] = {'P', 'T', 'X', 'F', 'L', 'A', 'T', '3'};  // bump the digit with the format

uint64_t fnv(uint64_t h, const void* data, size_t n) {
  const unsigned char* p = (const unsigned char*)data;
  for (size_t i = 0; i < n; i++) {
    h ^= p[i];
    h *= 1099511628211ull;
  }
  return h;
}
uint64_t fnvFile(uint64_t h, const std::string& path, const char* absentTag) {
  h = fnv(h, path.data() + (path.find_last_of('/') == std::string::npos ? 0 : path.find_last_of('/') + 1),
          path.size() - (path.find_last_of('/') == std::string::npos ? 0 : path.find_last_of('/') + 1));
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return fnv(h, absentTag, strlen(absentTag));
  std::vector<unsigned char> buf(1 << 20);
  size_t got;
  uint64_t total = 0;
  while ((got = fread(buf.data(), 1, buf.size(), f)) > 0) {
    h = fnv(h, buf.data(), got);
    total += got;
  }
  fclose(f);
  return fnv(h, &total, sizeof(total));
}

template <typename T>
bool putVec(FILE* f, const std::vector<T>& v) {
  uint64_t n = v.size();
  return fwrite(&n, sizeof(n), 1, f) == 1 && (n == 0 || fwrite(v.data(), sizeof(T), n, f) == n);
}
template <typename T>
bool getVec(FILE* f, std::vector<T>& v, uint64_t limitBytes) {
  uint64_t n = 0;
  if (fread(&n, sizeof(n), 1, f) != 1 || n > limitBytes / sizeof(T)) return false;  // no wrap-around for a huge n
  v.resize(n);
  return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

template <typename F>
bool eachArray(FlatScene& fl, F&& io) {
  return io(fl.tlas) && io(fl.instances) && io(fl.geometries) && io(fl.blasNodes) && io(fl.triangles) && io(fl.faceNormals) &&
         io(fl.vtxNormals) && io(fl.vtxUvs) && io(fl.materials) && io(fl.textures) && io(fl.texels) && io(fl.environment);
}

struct CacheHeader {
  char magic[8];
  uint64_t key;
  ptx_camera camera;
  pth_info info;
  uint32_t envWidth, envHeight;
};

bool loadCache(const std::string& path, uint64_t key, pth_scene* s) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  struct stat st {};
  if (fstat(fileno(f), &st) != 0) {
    fclose(f);
    return false;
  }
  CacheHeader h{};
  bool ok = fread(&h, sizeof(h), 1, f) == 1 && memcmp(h.magic, kCacheMagic, 8) == 0 && h.key == key;
  const uint64_t limit = (uint64_t)st.st_size;
  ok = ok && eachArray(s->flat, [&](auto& v) { return getVec(f, v, limit); });
  uint64_t tail = 0;
  ok = ok && fread(&tail, sizeof(tail), 1, f) == 1 && tail == (key ^ 0x5043414348454f4bull);  // written last: a torn file has no tail
  fclose(f);
  // the header must agree with the arrays it describes: the backend reads env_width * env_height texels
  ok = ok && h.envWidth != 0 && h.envHeight != 0 && (uint64_t)h.envWidth * h.envHeight * 3u == s->flat.environment.size();
  if (!ok) return false;
  s->flat.envWidth = h.envWidth;
  s->flat.envHeight = h.envHeight;
  s->flat.bind();
  s->fromCache = true;
  s->cachedCamera = h.camera;
  s->cachedInfo = h.info;
  return true;
}

bool storeCache(const std::string& path, uint64_t key, pth_scene* s) {
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  CacheHeader h{};
  memcpy(h.magic, kCacheMagic, 8);
  h.key = key;
  pth_scene_camera(s, &h.camera);
  pth_scene_info(s, &h.info);
  h.envWidth = s->flat.envWidth;
  h.envHeight = s->flat.envHeight;
  bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
  ok = ok && eachArray(s->flat, [&](auto& v) { return putVec(f, v); });
  const uint64_t tail = key ^ 0x5043414348454f4bull;
  ok = ok && fwrite(&tail, sizeof(tail), 1, f) == 1;
  ok = (fclose(f) == 0) && ok;
  if (ok) ok = rename(tmp.c_str(), path.c_str()) == 0;  // readers see the old file or the complete new one
  if (!ok) unlink(tmp.c_str());
  return ok;
}
}  // namespace

extern "C" {

int pth_scene_create_cached(int preset, const char* models_dir, const char* env_path, uint32_t scatter_instances,
                            uint32_t sky_w, uint32_t sky_h, const char* cache_dir, int* cache_hit, pth_scene** out) {
  if (cache_hit) *cache_hit = 0;
  if (!cache_dir || !*cache_dir) return pth_scene_create(preset, models_dir, env_path, scatter_instances, sky_w, sky_h, out);
  if (!out || preset < 0 || preset > 2) {
    g_err = "pth_scene_create_cached: bad arguments";
    return -1;
  }
  SceneAssets defaults;
  const std::string models = (models_dir && *models_dir) ? models_dir : defaults.modelsDir;
  const std::string env = env_path ? env_path : "";
  uint64_t key = 14695981039346656037ull;
  key = fnv(key, kCacheMagic, 8);
  // the file format is the flattened arrays themselves: every record size and a revision of the code that fills
  // them (flatten.cpp, sah_builder.hpp, sample_scenes.cpp) go into the key, so a changed builder cannot reuse old arrays
  constexpr uint32_t kFlattenRevision = 2;
  const uint32_t params[] = {(uint32_t)preset, scatter_instances, sky_w, sky_h, kFlattenRevision,
                             (uint32_t)sizeof(ptx_instance), (uint32_t)sizeof(ptx_bvh_node), (uint32_t)sizeof(ptx_triangle),
                             (uint32_t)sizeof(ptx_geometry), (uint32_t)sizeof(ptx_material), (uint32_t)sizeof(ptx_texture),
                             (uint32_t)sizeof(CacheHeader)};
  key = fnv(key, params, sizeof(params));
  if (preset == 2) {
    key = fnvFile(key, models + "/viking-room/viking_room.obj", "absent");
    key = fnvFile(key, models + "/viking-room/albedo.png", "absent");
    key = fnvFile(key, models + "/viking-room/emissive.png", "absent");
  } else {
    key = fnvFile(key, models + "/bust-of-menelaus.stl", "stand-in bust v1");
  }
  key = fnvFile(key, env, "stand-in sky v1");
  mkdir(cache_dir, 0700);  // private to the user: the arrays are trusted as far as the checks in loadCache go
  char name[64];
  snprintf(name, sizeof(name), "/scene%d_%016llx.ptxflat", preset, (unsigned long long)key);
  const std::string path = std::string(cache_dir) + name;
  auto* s = new pth_scene();
  if (loadCache(path, key, s)) {
    if (cache_hit) *cache_hit = 1;
    *out = s;
    return 0;
  }
  delete s;
  int rc = pth_scene_create(preset, models_dir, env_path, scatter_instances, sky_w, sky_h, out);
  if (rc == 0) (void)storeCache(path, key, *out);  // a cache that cannot be written is not an error
  return rc;
}

uint32_t pth_heatmap_colour(const uint32_t* counts, uint32_t n, float* rgb) {
  return heatmapColour(counts, n, rgb);
}

int pth_write_bmp(const char* path, const float* rgb, uint32_t w, uint32_t h) {
  return writeBmp(path, rgb, w, h) ? 0 : -1;
}

long pth_write_standin_stl(const char* path, uint32_t degenerate_every) {
  if (!path) return -1;
  const bool quiet = g_quiet;
  g_quiet = true;
  std::vector<Triangle> tris = standInBust();
  g_quiet = quiet;
  if (degenerate_every) {
    std::vector<Triangle> with;
    with.reserve(tris.size() + tris.size() / degenerate_every + 1);
    for (size_t i = 0; i < tris.size(); i++) {
      with.push_back(tris[i]);
      if ((i + 1) % degenerate_every == 0) {
        Triangle d = tris[i];
        d.v2 = d.v0;  // names one vertex twice: stl_reader drops it
        with.push_back(d);
      }
    }
    tris.swap(with);
  }
  return writeBinarySTL(path, tris) ? (long)tris.size() : -1;
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
