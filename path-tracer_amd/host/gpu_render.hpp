// Host side of the drop-in boundary: the reference's renderGPU (gpuRender.hpp:6-10) with the same
// name, argument list and console output, implemented on the C ABI of include/ptx_backend.h.
#pragma once

#include <memory>
#include <vector>

#include "scene.hpp"

namespace pt {

// What the reference's argument list has no place for (set once before the render calls; defaults =
// the reference's behaviour: one GPU, the reference's seeding).
struct GpuOptions {
  uint32_t numGpus = 1;     // -g: devices 0 .. numGpus-1 driven from this process (ptx_render_multi); > 1 implies
                            //     PTX_SEED_STREAM (one RNG stream set per GPU; part 0 is the reference's seeding)
  uint32_t subStreams = 0;  // -r: RNG streams per pixel and GPU (ptx_params.sub_streams); > 1 implies PTX_SEED_STREAM
  bool sameDevice = false;  // PT_MULTI_SAME_DEVICE=1: every part on device 0 (rehearsal of -g on a one-GPU machine)
};
void setGpuOptions(const GpuOptions& options);

void renderGPU(const Scene& scene, const std::vector<std::shared_ptr<Geometry>>& geometryPool,
               const std::vector<std::shared_ptr<Material>>& materialPool, const Camera& camera, Image& image,
               const Image& environment, const uint samples, const int maxBounces, const bool renderBVH);

// The same call with the device-side output stage: the backend also does main()'s heat-map
// colouring (main.cpp:363-374) and output.cpp's clamp / 8-bit quantise / flip, and hands back the BMP pixel
// array (bmpRowBytes(width) * height bytes) -- 3 bytes per pixel cross PCIe instead of 12.  maxTests
// receives "Maximum BVH tests" when renderBVH is set.  Same console output as renderGPU.
void renderGPURows(const Scene& scene, const std::vector<std::shared_ptr<Geometry>>& geometryPool,
                   const std::vector<std::shared_ptr<Material>>& materialPool, const Camera& camera, uint32_t width,
                   uint32_t height, const Image& environment, const uint samples, const int maxBounces, const bool renderBVH,
                   std::vector<unsigned char>& rows, uint32_t& maxTests);

}  // namespace pt
