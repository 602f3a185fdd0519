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
}  // namespace

// gpuRender.cu:172-311.  Upload -> "GPU data transfer time", kernel -> "Render time" (kernel only,
// measured with HIP events inside the backend), copy back, free.  With renderBVH the image receives
// float(count) in all three channels (gpuRender.cu:150) for main()'s heat-map colouring.
void renderGPU(const Scene& scene, const std::vector<std::shared_ptr<Geometry>>& geometryPool,
               const std::vector<std::shared_ptr<Material>>& materialPool, const Camera& camera, Image& image,
               const Image& environment, const uint samples, const int maxBounces, const bool renderBVH) {
  FlatScene flat;
  flattenScene(scene, geometryPool, materialPool, environment, flat);

  ptx_scene* handle = nullptr;
  check(ptx_create(&flat.desc, 0, &handle), "ptx_create");

  ptx_camera cam = toPtxCamera(camera);
  ptx_params prm{};
  prm.width = image.width;
  prm.height = image.height;
  prm.spp = samples;
  prm.max_bounces = maxBounces < 0 ? 0u : (uint32_t)maxBounces;
  prm.seed_mode = PTX_SEED_REFERENCE_GPU;
  prm.part_index = 0;
  prm.part_count = 1;

  ptx_stats stats{};
  const size_t n = (size_t)image.width * image.height;
  if (renderBVH) {
    std::vector<uint32_t> counts(n);
    printf("Rendering...\n");
    check(ptx_render_counts(handle, &cam, &prm, counts.data(), &stats), "ptx_render_counts");
    for (size_t i = 0; i < n; i++) image.data[i] = vec3((float)counts[i]);
  } else {
    printf("Rendering...\n");
    check(ptx_render(handle, &cam, &prm, reinterpret_cast<float*>(image.data.data()), &stats), "ptx_render");
  }
  printf("\nGPU data transfer time: %g s\n", floor4(stats.upload_s));
  printf("\nRender time: %g s\n", floor4(stats.render_s));
  ptx_destroy(handle);
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
  ptx_params prm{};
  prm.width = width;
  prm.height = height;
  prm.spp = samples;
  prm.max_bounces = maxBounces < 0 ? 0u : (uint32_t)maxBounces;
  prm.seed_mode = PTX_SEED_REFERENCE_GPU;
  prm.part_index = 0;
  prm.part_count = 1;
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
