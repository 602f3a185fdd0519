// Host side of the drop-in boundary: the reference's renderGPU (gpuRender.hpp:6-10) with the same
// name, argument list and console output, implemented on the C ABI of include/ptx_backend.h.
#pragma once

#include <memory>
#include <vector>

#include "scene.hpp"

namespace pt {

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
