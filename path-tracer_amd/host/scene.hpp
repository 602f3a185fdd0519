// Host scene graph: Image, Material, Geometry (BLAS owner), Mesh (instance), Scene (TLAS owner).
// Same class and member names, argument meaning and setter behaviour as the reference's
// image.hpp:9-36, material.hpp:6-33, geometry.hpp:9-36, mesh.hpp:12-57 and scene.hpp:9-27, so that
// sample_scenes.cpp reads like the reference's presets.  Nothing here traverses a BVH: on this
// repository's product path that only happens in the HIP kernels (path-tracer_amd/csrc).
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "scene_types.hpp"

namespace pt {

class Image {  // image.hpp:9-36, row 0 = bottom scan line for render targets
 public:
  uint width{0};
  uint height{0};
  std::vector<vec3> data{};
  Image() = default;
  Image(uint width, uint height) : width{width}, height{height}, data(size_t(width) * height) {}
  Image(uint width, uint height, std::vector<vec3> data) : width{width}, height{height}, data{std::move(data)} {}
  vec3 operator[](uint i) const { return data[i]; }
  vec3& operator[](uint i) { return data[i]; }
};

class Material {  // material.hpp:6-33
 public:
  vec3 albedo{1};
  float metalness{0};
  float roughness{0.01f};
  vec3 emissive{0};
  Image albedoTexture{};
  Image emissiveTexture{};
  // material.hpp:23-25: IOR 1.5 -> F0 = (IOR-1)^2 / (IOR+1)^2 (powf), 0.04
  float IOR{1.5f};
  vec3 F0 = vec3(powf(IOR - 1.0f, 2.0f) / powf(IOR + 1.0f, 2.0f));
};

class Geometry {  // geometry.hpp:9-36
 public:
  std::vector<Triangle> primitives{};
  std::vector<vec3> faceNormals{};
  VertexAttributes attributes{};
  std::vector<BVHNode> bvh{};
  vec3 aabbMin{FLT_MAX};
  vec3 aabbMax{-FLT_MAX};
  vec3 corners[8];
  vec3 centroid{0};
  uint bvhDepth{0};
  uint maxLeafSize{0};
  double buildSeconds{0};

  Geometry(std::vector<Triangle> primitives, VertexAttributes attributes = VertexAttributes{});
};

class Mesh {  // mesh.hpp:12-57
 public:
  std::shared_ptr<Geometry> geometry;
  std::shared_ptr<Material> material;
  vec3 translation{0};
  vec3 rotation{0};
  float scale{1};
  mat4 modelMatrix = mat4::identity();
  mat4 invModelMatrix = mat4::identity();
  mat4 normalMatrix = mat4::identity();
  vec3 aabbMin{FLT_MAX};
  vec3 aabbMax{-FLT_MAX};
  vec3 centroid{0};

  Mesh() = default;
  Mesh(std::shared_ptr<Geometry> geometry, std::shared_ptr<Material> material);
  void setPosition(vec3 translation);
  void setRotationX(float angle);
  void setRotationY(float angle);
  void setRotationZ(float angle);
  void setScale(float scale);
  void centerGeometry();
  void update();
  vec3 getMin() const { return aabbMin; }
  vec3 getMax() const { return aabbMax; }
};

class Scene {  // scene.hpp:9-27
 public:
  std::vector<Mesh> meshes;
  std::vector<BVHNode> tlas;
  uint tlasDepth{0};
  uint maxLeafSize{0};
  double buildSeconds{0};
  bool quiet{false};
  Scene() = default;
  Scene(const Scene&) = delete;
  void completeScene();  // scene.cpp:16-41: build the TLAS, reorder meshes into leaf order
};

}  // namespace pt
