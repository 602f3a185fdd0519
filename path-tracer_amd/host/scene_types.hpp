// Plain data types of the host scene pipeline.  They mirror the reference's dataTypes.hpp:26-108
// (same names and meaning, packed 12-byte vec3 so a BVHNode is 32 bytes) because their byte layout
// is what ptx_scene_desc (include/ptx_backend.h) carries across the C ABI.
#pragma once

#include <climits>
#include <vector>

#include "ptmath.hpp"

namespace pt {

struct AABB {  // dataTypes.hpp:26-47
  vec3 min{FLT_MAX};
  vec3 max{-FLT_MAX};
  void grow(vec3 p) {
    min = gmin(min, p);
    max = gmax(max, p);
  }
  void grow(const AABB& b) {
    min = gmin(min, b.min);
    max = gmax(max, b.max);
  }
  float area() const {
    vec3 dim = max - min;
    return 2.0f * (dim.x * dim.y + dim.y * dim.z + dim.z * dim.x);
  }
};

struct Triangle {  // dataTypes.hpp:49-57
  vec3 v0, v1, v2;
  vec3 centroid{FLT_MAX};
  vec3 getMin() const { return gmin(gmin(v0, v1), v2); }
  vec3 getMax() const { return gmax(gmax(v0, v1), v2); }
};

struct VertexAttributes {  // dataTypes.hpp:66-69, three entries per triangle
  std::vector<vec3> normals;
  std::vector<vec2> texCoords;
};

struct BVHNode {  // dataTypes.hpp:72-78; same bytes as ptx_bvh_node
  AABB aabb;
  uint leftFirst;
  uint count;
};
static_assert(sizeof(vec3) == 12, "vec3 must be packed");
static_assert(sizeof(BVHNode) == 32, "BVHNode must match ptx_bvh_node");

struct Camera {  // dataTypes.hpp:101-108
  vec3 position;
  vec3 target;
  vec3 up;
  float fieldOfView = 0.0f;
};

}  // namespace pt
