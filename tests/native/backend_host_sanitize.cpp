// The HOST side of the HIP backend (validation of the description, the four-wide re-layout, the tight TLAS refit)
// under AddressSanitizer + UndefinedBehaviourSanitizer: ptx_backend.hip is compiled with the sanitizers on its host
// half only (-Xarch_host) and linked with the host pipeline; no GPU is touched (ptx_check_wide_layout needs none,
// ptx_check_tight_tlas is called with device -1).  tests/test_host_sanitizers.py builds and runs it.
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/pt_host.h"

#define CHECK(cond)                                                                                        \
  do {                                                                                                     \
    if (!(cond)) {                                                                                         \
      std::fprintf(stderr, "backend_host_sanitize: %s:%d: %s failed (%s / %s)\n", __FILE__, __LINE__, #cond, \
                   pth_last_error(), ptx_last_error());                                                    \
      return 1;                                                                                            \
    }                                                                                                      \
  } while (0)

int main(int argc, char** argv) {
  CHECK(argc == 2);
  for (int preset = 0; preset < 3; preset++) {
    pth_scene* s = nullptr;
    CHECK(pth_scene_create(preset, argv[1], "/nonexistent.hdr", preset == 1 ? 60 : 0, 32, 16, &s) == 0);
    const ptx_scene_desc* d = pth_scene_desc(s);
    uint64_t r[6];
    CHECK(ptx_check_wide_layout(d, r) == 0);
    CHECK(r[0] > 0 && r[2] == 0 && r[3] == 0);
    std::printf("preset %d: %llu wide nodes, %llu slots, depth %llu\n", preset, (unsigned long long)r[0], (unsigned long long)r[1],
                (unsigned long long)r[4]);
    CHECK(ptx_check_tight_tlas(d, -1, r) == 0);
    CHECK(r[2] == 0 && r[3] == 0 && r[4] == 0);
    // a description with a link out of range must be refused, not followed
    ptx_scene_desc bad = *d;
    std::vector<ptx_bvh_node> nodes(d->tlas_nodes, d->tlas_nodes + d->tlas_node_count);
    if (nodes[0].count == 0) {
      nodes[0].left_first = d->tlas_node_count + 7u;
      bad.tlas_nodes = nodes.data();
      CHECK(ptx_check_wide_layout(&bad, r) != 0);
    }
    pth_scene_destroy(s);
  }
  std::puts("backend_host_sanitize: ok");
  return 0;
}
