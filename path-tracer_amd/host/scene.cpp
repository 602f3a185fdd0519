#include "scene.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "sah_builder.hpp"

namespace pt {

namespace {
template <typename T>
void permute(std::vector<T>& values, const std::vector<uint>& order, uint group = 1) {  // dataTypes.hpp:133-142
  std::vector<T> out(values.size());
  for (size_t i = 0; i < order.size(); i++)
    for (uint j = 0; j < group; j++) out[size_t(group) * i + j] = values[size_t(group) * order[i] + j];
  values.swap(out);
}
// Threads of the BVH builder: PT_BUILD_THREADS, else the machine's cores, at most 16 (the tree does not
// depend on it: sah_builder.hpp).
uint buildThreads() {
  if (const char* e = getenv("PT_BUILD_THREADS")) {
    int v = atoi(e);
    if (v >= 1) return (uint)(v > 64 ? 64 : v);
  }
  uint hw = std::thread::hardware_concurrency();
  return hw == 0 ? 1u : (hw > 16u ? 16u : hw);
}
double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// geometry.cpp:11-34 + generateBVH (:36-67) + generateNormals (:89-104)
Geometry::Geometry(std::vector<Triangle> prims, VertexAttributes attrs)
    : primitives{std::move(prims)}, attributes{std::move(attrs)} {
  const uint n = (uint)primitives.size();
  if (n < 1) {
    fprintf(stderr, "Can't generate BVH for empty primitive container\n");
    return;
  }
  double t0 = now_s();
  std::vector<vec3> lo(n), hi(n), cen(n);
  for (uint i = 0; i < n; i++) {
    lo[i] = primitives[i].getMin();
    hi[i] = primitives[i].getMax();
    cen[i] = primitives[i].centroid;
  }
  BuildResult built = buildBVHParallel(BuildInput{lo.data(), hi.data(), cen.data(), n}, buildThreads());
  bvh = std::move(built.nodes);
  bvhDepth = built.maxDepth;
  maxLeafSize = built.maxLeafSize;

  // Triangles and per-vertex attributes are physically moved into leaf order (geometry.cpp:54-62).
  permute(primitives, built.order);
  if (!attributes.normals.empty()) permute(attributes.normals, built.order, 3u);
  if (!attributes.texCoords.empty()) permute(attributes.texCoords, built.order, 3u);
  buildSeconds = now_s() - t0;

  faceNormals.resize(n);
  for (uint i = 0; i < n; i++) {  // geometry.cpp:89-91
    const Triangle& t = primitives[i];
    faceNormals[i] = normalize(cross(t.v0 - t.v1, t.v0 - t.v2));
  }

  aabbMin = bvh[0].aabb.min;
  aabbMax = bvh[0].aabb.max;
  corners[0] = vec3{aabbMin.x, aabbMin.y, aabbMin.z};
  corners[1] = vec3{aabbMin.x, aabbMax.y, aabbMin.z};
  corners[2] = vec3{aabbMin.x, aabbMin.y, aabbMax.z};
  corners[3] = vec3{aabbMin.x, aabbMax.y, aabbMax.z};
  corners[4] = vec3{aabbMax.x, aabbMin.y, aabbMin.z};
  corners[5] = vec3{aabbMax.x, aabbMax.y, aabbMin.z};
  corners[6] = vec3{aabbMax.x, aabbMin.y, aabbMax.z};
  corners[7] = vec3{aabbMax.x, aabbMax.y, aabbMax.z};
  centroid = bvh[0].aabb.min + 0.5f * (bvh[0].aabb.max - bvh[0].aabb.min);
}

Mesh::Mesh(std::shared_ptr<Geometry> g, std::shared_ptr<Material> m) : geometry{std::move(g)}, material{std::move(m)} {
  update();
}

// mesh.cpp:18-41.  M = I*S*Rx*Ry*Rz, then a world-space translation expressed in model space, then
// the geometry is centred on its own AABB centre.
void Mesh::update() {
  modelMatrix = mat4::identity();
  modelMatrix = pt::scale(modelMatrix, vec3(scale));
  modelMatrix = pt::rotate(modelMatrix, rotation.x, vec3(1, 0, 0));
  modelMatrix = pt::rotate(modelMatrix, rotation.y, vec3(0, 1, 0));
  modelMatrix = pt::rotate(modelMatrix, rotation.z, vec3(0, 0, 1));
  modelMatrix = pt::translate(modelMatrix, xyz(inverse(modelMatrix) * vec4(translation, 1.0f)));
  modelMatrix = pt::translate(modelMatrix, -geometry->centroid);
  invModelMatrix = inverse(modelMatrix);
  normalMatrix = transpose(invModelMatrix);

  aabbMin = vec3{FLT_MAX};
  aabbMax = vec3{-FLT_MAX};
  for (const vec3& corner : geometry->corners) {
    vec3 w = xyz(modelMatrix * vec4(corner, 1.0f));
    aabbMin = gmin(aabbMin, w);
    aabbMax = gmax(aabbMax, w);
  }
  centroid = aabbMin + 0.5f * (aabbMax - aabbMin);
}

void Mesh::setPosition(vec3 t) { translation = t; update(); }
void Mesh::setRotationX(float a) { rotation.x = a; update(); }
void Mesh::setRotationY(float a) { rotation.y = a; update(); }
void Mesh::setRotationZ(float a) { rotation.z = a; update(); }
void Mesh::setScale(float s) { scale = s; update(); }
void Mesh::centerGeometry() { translation = -geometry->centroid; update(); }

void Scene::completeScene() {
  const uint n = (uint)meshes.size();
  if (n == 0) return;
  double t0 = now_s();
  std::vector<vec3> lo(n), hi(n), cen(n);
  for (uint i = 0; i < n; i++) {
    lo[i] = meshes[i].getMin();
    hi[i] = meshes[i].getMax();
    cen[i] = meshes[i].centroid;
  }
  BuildResult built = buildBVHParallel(BuildInput{lo.data(), hi.data(), cen.data(), n}, buildThreads());
  tlas = std::move(built.nodes);
  tlasDepth = built.maxDepth;
  maxLeafSize = built.maxLeafSize;
  permute(meshes, built.order);  // scene.cpp:36
  buildSeconds = now_s() - t0;
  if (!quiet) {
    printf("\nTLAS build time: %g s\n", std::floor(buildSeconds * 1e4) / 1e4);
    printf("TLAS nodes: %zu\n", tlas.size());
  }
}

}  // namespace pt
