// Scene graph -> POD arrays of ptx_scene_desc (include/ptx_backend.h).  This replaces the reference's
// pointer-chasing device mirror (GPUImage/GPUGeometry/GPUMaterial/GPUMesh/GPUScene constructors,
// gpuRender.cu:188-262): the C++ objects are flattened once on the host into pooled arrays in the
// reference's own element layouts, and the backend re-lays those out for the GPU.
#pragma once

#include <memory>
#include <vector>

#include "../../include/ptx_backend.h"
#include "scene.hpp"

namespace pt {

struct FlatScene {
  std::vector<ptx_bvh_node> tlas;
  std::vector<ptx_instance> instances;
  std::vector<ptx_geometry> geometries;
  std::vector<ptx_bvh_node> blasNodes;
  std::vector<ptx_triangle> triangles;
  std::vector<float> faceNormals, vtxNormals, vtxUvs;
  std::vector<ptx_material> materials;
  std::vector<ptx_texture> textures;
  std::vector<float> texels;
  std::vector<float> environment;
  uint envWidth = 0, envHeight = 0;
  ptx_scene_desc desc{};  // points into the vectors above; valid while this object lives unchanged
  void bind();
};

// Mesh -> pool index lookup follows gpuRender.cu:251-256 (pointer identity in the two pools).
void flattenScene(const Scene& scene, const std::vector<std::shared_ptr<Geometry>>& geometryPool,
                  const std::vector<std::shared_ptr<Material>>& materialPool, const Image& environment,
                  FlatScene& out);

ptx_camera toPtxCamera(const Camera& camera);

}  // namespace pt
