#include "gpu_render.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../include/ptx_backend.h"
#include "flatten.hpp"
#include "output.hpp"

namespace pt {

namespace {
// error.hpp:10-26: any device error prints and exits.
void check(int rc, const char* what) {
  if (rc != PTX_OK) {
    fprintf(stderr, "GPU backend error in %s: %s\n", what, ptx_last_error());
    std::exit(EXIT_FAILURE);
  }
}
double floor4(double s) { return std::floor(s * 1e4) / 1e4; }
GpuOptions g_options;

ptx_params makeParams(uint32_t width, uint32_t height, uint32_t samples, int maxBounces) {
  ptx_params prm{};
  prm.width = width;
  prm.height = height;
  prm.spp = samples;
  prm.max_bounces = maxBounces < 0 ? 0u : (uint32_t)maxBounces;
  const bool streams = g_options.numGpus > 1 || g_options.subStreams > 1;
  prm.seed_mode = streams ? PTX_SEED_STREAM : PTX_SEED_REFERENCE_GPU;
  prm.part_index = 0;
  prm.part_count = 1;
  prm.sub_streams = g_options.subStreams;
  return prm;
}
}  // namespace

void setGpuOptions(const GpuOptions& options) { g_options = options; }

// gpuRender.cu:172-311.  Upload -> "GPU data transfer time", kernel -> "Render time" (kernel only,
// measured with HIP events inside the backend), copy back, free.  With renderBVH the image receives
// float(count) in all three channels (gpuRender.cu:150) for main()'s heat-map colouring.
void renderGPU(const Scene& scene, const std::vector<std::shared_ptr<Geometry>>& geometryPool,
               const std::vector<std::shared_ptr<Material>>& materialPool, const Camera& camera, Image& image,
               const Image& environment, const uint samples, const int maxBounces, const bool renderBVH) {
  FlatScene flat;
  flattenScene(scene, geometryPool, materialPool, environment, flat);

  // the scene is replicated: one handle per device (-g), all driven from this thread
  const uint32_t nGpus = g_options.numGpus ? g_options.numGpus : 1u;
  std::vector<ptx_scene*> handles(nGpus, nullptr);
  double upload = 0.0;
  for (uint32_t g = 0; g < nGpus; g++) {
    check(ptx_create(&flat.desc, g_options.sameDevice ? 0 : (int)g, &handles[g]), "ptx_create");
  }

  ptx_camera cam = toPtxCamera(camera);
  ptx_params prm = makeParams(image.width, image.height, samples, maxBounces);

  ptx_stats stats{};
  const size_t n = (size_t)image.width * image.height;
  printf("Rendering...\n");
  if (renderBVH) {
    std::vector<uint32_t> counts(n);
    if (nGpus > 1) check(ptx_render_counts_multi(handles.data(), nGpus, &cam, &prm, counts.data(), &stats), "ptx_render_counts_multi");
    else check(ptx_render_counts(handles[0], &cam, &prm, counts.data(), &stats), "ptx_render_counts");
    for (size_t i = 0; i < n; i++) image.data[i] = vec3((float)counts[i]);
  } else {
    float* out = reinterpret_cast<float*>(image.data.data());
    if (nGpus > 1) check(ptx_render_multi(handles.data(), nGpus, &cam, &prm, out, &stats), "ptx_render_multi");
    else check(ptx_render(handles[0], &cam, &prm, out, &stats), "ptx_render");
  }
  upload = stats.upload_s;
  printf("\nGPU data transfer time: %g s\n", floor4(upload));
  printf("\nRender time: %g s\n", floor4(stats.render_s));
  for (ptx_scene* h : handles) ptx_destroy(h);
}

void renderGPURows(const Scene& scene, const std::vector<std::shared_ptr<Geometry>>& geometryPool,
                   const std::vector<std::shared_ptr<Material>>& materialPool, const Camera& camera, uint32_t width,
                   uint32_t height, const Image& environment, const uint samples, const int maxBounces, const bool renderBVH,
                   std::vector<unsigned char>& rows, uint32_t& maxTests) {
  FlatScene flat;
  flattenScene(scene, geometryPool, materialPool, environment, flat);
  ptx_scene* handle = nullptr;
  check(ptx_create(&flat.desc, 0, &handle), "ptx_create");
  ptx_camera cam = toPtxCamera(camera);
  ptx_params prm = makeParams(width, height, samples, maxBounces);
  ptx_stats stats{};
  const uint32_t rowBytes = bmpRowBytes(width);
  rows.assign((size_t)rowBytes * height, 0);
  maxTests = 0;
  printf("Rendering...\n");
  if (renderBVH) check(ptx_render_counts_bgr8(handle, &cam, &prm, rows.data(), rowBytes, &maxTests, &stats), "ptx_render_counts_bgr8");
  else check(ptx_render_bgr8(handle, &cam, &prm, rows.data(), rowBytes, &stats), "ptx_render_bgr8");
  printf("\nGPU data transfer time: %g s\n", floor4(stats.upload_s));
  printf("\nRender time: %g s\n", floor4(stats.render_s));
  ptx_destroy(handle);
}

}  // namespace pt
