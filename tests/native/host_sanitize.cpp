// The host pipeline (libpt_host's sources) and the CPU oracle, linked into one program and run under
// AddressSanitizer + UndefinedBehaviourSanitizer (host code only; tests/test_host_sanitizers.py builds and runs it).
// Walks what the GPU-less part of the product does: the three presets, the flattened-scene cache (store, hit,
// a truncated file), the STL ingest on the stand-in bust with degenerate facets, the output stage, and the
// oracle's two render modes on what came out.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include "../../include/pt_host.h"
#include "../../oracle/pt_oracle.h"

#define CHECK(cond)                                                                  \
  do {                                                                               \
    if (!(cond)) {                                                                   \
      std::fprintf(stderr, "host_sanitize: %s:%d: %s failed (%s)\n", __FILE__, __LINE__, #cond, pth_last_error()); \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

static ptx_params params(uint32_t w, uint32_t h, uint32_t spp, uint32_t bounces) {
  ptx_params p;
  std::memset(&p, 0, sizeof p);
  p.width = w;
  p.height = h;
  p.spp = spp;
  p.max_bounces = bounces;
  p.part_count = 1;
  return p;
}

static int renderBoth(const pth_scene* s, uint32_t w, uint32_t h) {
  ptx_camera cam;
  pth_scene_camera(s, &cam);
  const ptx_scene_desc* d = pth_scene_desc(s);
  std::vector<uint32_t> counts((size_t)w * h);
  ptx_stats st;
  ptx_params pc = params(w, h, 1, 1);
  CHECK(oracle_render_counts(d, &cam, &pc, 2, counts.data(), &st) == 0);
  std::vector<float> rgb((size_t)w * h * 3);
  const uint32_t mx = pth_heatmap_colour(counts.data(), w * h, rgb.data());
  CHECK(mx > 0);
  ptx_params pr = params(w, h, 2, 4);
  CHECK(oracle_render(d, &cam, &pr, PTX_SEED_REFERENCE_GPU, 2, rgb.data(), &st) == 0);
  for (float c : rgb) CHECK(std::isfinite(c) && c >= 0.0f && c <= 1.0f);
  std::vector<unsigned char> rows((size_t)pth_bmp_row_bytes(w) * h);
  pth_quantise_rows(rgb.data(), w, h, rows.data());
  return 0;
}

int main(int argc, char** argv) {
  CHECK(argc == 3);  // scratch directory, models directory (viking-room/ for preset 2)
  const std::string dir = argv[1];
  const char* goldenModels = argv[2];
  mkdir(dir.c_str(), 0700);
  const std::string cache = dir + "/cache";

  // preset 2 (the viking room: OBJ + texture), odd sizes
  pth_scene* s = nullptr;
  CHECK(pth_scene_create(2, goldenModels, "/nonexistent.hdr", 0, 64, 32, &s) == 0);
  if (renderBoth(s, 37, 29)) return 1;
  const std::string bmp = dir + "/x.bmp";
  std::vector<float> img(37 * 29 * 3, 0.5f);
  CHECK(pth_write_bmp(bmp.c_str(), img.data(), 37, 29) == 0);
  pth_scene_destroy(s);

  // preset 1 with few instances; through the cache: miss, hit, and a file cut short
  int hit = -1;
  CHECK(pth_scene_create_cached(1, "/nonexistent", "/nonexistent.hdr", 40, 64, 32, cache.c_str(), &hit, &s) == 0 && hit == 0);
  pth_info info;
  pth_scene_info(s, &info);
  CHECK(info.instances == 40 && info.stand_in_geometry == 1);
  if (renderBoth(s, 48, 20)) return 1;
  pth_scene_destroy(s);
  CHECK(pth_scene_create_cached(1, "/nonexistent", "/nonexistent.hdr", 40, 64, 32, cache.c_str(), &hit, &s) == 0 && hit == 1);
  if (renderBoth(s, 48, 20)) return 1;
  pth_scene_destroy(s);
  {
    // truncate every file in the cache directory to two thirds: must be ignored and rewritten
    std::string cmd = "for f in " + cache + "/*; do s=$(stat -c %s \"$f\"); truncate -s $((s*2/3)) \"$f\"; done";
    CHECK(std::system(cmd.c_str()) == 0);
  }
  CHECK(pth_scene_create_cached(1, "/nonexistent", "/nonexistent.hdr", 40, 64, 32, cache.c_str(), &hit, &s) == 0 && hit == 0);
  pth_scene_destroy(s);

  // the STL ingest: the stand-in bust as a file with degenerate facets, read back as preset 0's model
  const std::string models = dir + "/models";
  mkdir(models.c_str(), 0700);
  CHECK(pth_write_standin_stl((models + "/bust-of-menelaus.stl").c_str(), 1000) > 0);
  CHECK(pth_scene_create(0, models.c_str(), "/nonexistent.hdr", 0, 64, 32, &s) == 0);
  pth_scene_info(s, &info);
  CHECK(info.stand_in_geometry == 0 && info.triangles_per_mesh > 100000);
  if (renderBoth(s, 40, 24)) return 1;
  pth_scene_destroy(s);

  // small probes
  uint32_t st = 1031;
  (void)pth_pcg_hash(&st);
  (void)pth_random_float(&st);
  double v = 0;
  CHECK(pth_parse_decimal("-1.25e2", &v) == 1 || v == -125.0);
  std::puts("host_sanitize: ok");
  return 0;
}
