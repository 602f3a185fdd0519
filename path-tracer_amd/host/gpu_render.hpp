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

}  // namespace pt
