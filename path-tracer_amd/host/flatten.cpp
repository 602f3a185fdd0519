#include "flatten.hpp"

#include <algorithm>
#include <cstring>

namespace pt {

static_assert(sizeof(ptx_bvh_node) == sizeof(BVHNode), "node layout");
static_assert(sizeof(ptx_triangle) == 36, "triangle layout");

void FlatScene::bind() {
  desc = ptx_scene_desc{};
  desc.tlas_nodes = tlas.data();            desc.tlas_node_count = (uint32_t)tlas.size();
  desc.instances = instances.data();        desc.instance_count = (uint32_t)instances.size();
  desc.geometries = geometries.data();      desc.geometry_count = (uint32_t)geometries.size();
  desc.blas_nodes = blasNodes.data();       desc.blas_node_total = (uint32_t)blasNodes.size();
  desc.triangles = triangles.data();        desc.triangle_total = (uint32_t)triangles.size();
  desc.face_normals = faceNormals.data();   desc.face_normal_total = (uint32_t)(faceNormals.size() / 3);
  desc.vtx_normals = vtxNormals.empty() ? nullptr : vtxNormals.data();
  desc.vtx_normal_total = (uint32_t)(vtxNormals.size() / 3);
  desc.vtx_uvs = vtxUvs.empty() ? nullptr : vtxUvs.data();
  desc.vtx_uv_total = (uint32_t)(vtxUvs.size() / 2);
  desc.materials = materials.data();        desc.material_count = (uint32_t)materials.size();
  desc.textures = textures.empty() ? nullptr : textures.data();
  desc.texture_count = (uint32_t)textures.size();
  desc.texels = texels.empty() ? nullptr : texels.data();
  desc.texel_total = texels.size() / 3;
  desc.environment = environment.data();
  desc.env_width = envWidth;
  desc.env_height = envHeight;
}

namespace {
void copyMat(float dst[16], const mat4& m) {
  for (int c = 0; c < 4; c++) {
    dst[4 * c + 0] = m.c[c].x;
    dst[4 * c + 1] = m.c[c].y;
    dst[4 * c + 2] = m.c[c].z;
    dst[4 * c + 3] = m.c[c].w;
  }
}
uint32_t addTexture(FlatScene& out, const Image& img) {
  if (img.width == 0) return PTX_NONE;  // material.cpp:14,21: texture used only when width > 0
  ptx_texture t{};
  t.width = img.width;
  t.height = img.height;
  t.texel_offset = out.texels.size() / 3;
  const float* src = reinterpret_cast<const float*>(img.data.data());
  out.texels.insert(out.texels.end(), src, src + img.data.size() * 3);
  out.textures.push_back(t);
  return (uint32_t)out.textures.size() - 1;
}
}  // namespace

void flattenScene(const Scene& scene, const std::vector<std::shared_ptr<Geometry>>& geometryPool,
                  const std::vector<std::shared_ptr<Material>>& materialPool, const Image& environment,
                  FlatScene& out) {
  out = FlatScene{};
  out.tlas.resize(scene.tlas.size());
  memcpy(out.tlas.data(), scene.tlas.data(), scene.tlas.size() * sizeof(BVHNode));

  for (const auto& gp : geometryPool) {
    const Geometry& g = *gp;
    ptx_geometry d{};
    d.node_offset = (uint32_t)out.blasNodes.size();
    d.node_count = (uint32_t)g.bvh.size();
    d.tri_offset = (uint32_t)out.triangles.size();
    d.tri_count = (uint32_t)g.primitives.size();
    d.face_normal_offset = (uint32_t)(out.faceNormals.size() / 3);
    d.vtx_normal_offset = g.attributes.normals.empty() ? PTX_NONE : (uint32_t)(out.vtxNormals.size() / 3);
    // geometry.cpp:80 / gpuFunctions.hpp:6: uvs are used when the geometry has any
    d.vtx_uv_offset = g.attributes.texCoords.empty() ? PTX_NONE : (uint32_t)(out.vtxUvs.size() / 2);
    size_t nb = out.blasNodes.size();
    out.blasNodes.resize(nb + g.bvh.size());
    memcpy(out.blasNodes.data() + nb, g.bvh.data(), g.bvh.size() * sizeof(BVHNode));
    for (const Triangle& t : g.primitives) {  // geometry.cpp:111-115
      ptx_triangle p{{t.v0.x, t.v0.y, t.v0.z}, {t.v1.x, t.v1.y, t.v1.z}, {t.v2.x, t.v2.y, t.v2.z}};
      out.triangles.push_back(p);
    }
    for (const vec3& n : g.faceNormals) { out.faceNormals.push_back(n.x); out.faceNormals.push_back(n.y); out.faceNormals.push_back(n.z); }
    for (const vec3& n : g.attributes.normals) { out.vtxNormals.push_back(n.x); out.vtxNormals.push_back(n.y); out.vtxNormals.push_back(n.z); }
    for (const vec2& t : g.attributes.texCoords) { out.vtxUvs.push_back(t.x); out.vtxUvs.push_back(t.y); }
    out.geometries.push_back(d);
  }

  for (const auto& mp : materialPool) {
    const Material& m = *mp;
    ptx_material d{};
    d.albedo[0] = m.albedo.x; d.albedo[1] = m.albedo.y; d.albedo[2] = m.albedo.z;
    d.metalness = m.metalness;
    d.roughness = m.roughness;
    d.emissive[0] = m.emissive.x; d.emissive[1] = m.emissive.y; d.emissive[2] = m.emissive.z;
    d.f0[0] = m.F0.x; d.f0[1] = m.F0.y; d.f0[2] = m.F0.z;
    d.albedo_texture = addTexture(out, m.albedoTexture);
    d.emissive_texture = addTexture(out, m.emissiveTexture);
    out.materials.push_back(d);
  }

  out.instances.reserve(scene.meshes.size());
  for (const Mesh& mesh : scene.meshes) {
    ptx_instance d{};
    copyMat(d.inv_model, mesh.invModelMatrix);
    copyMat(d.normal_matrix, mesh.normalMatrix);
    d.geometry_id = (uint32_t)std::distance(geometryPool.begin(), std::find(geometryPool.begin(), geometryPool.end(), mesh.geometry));
    d.material_id = (uint32_t)std::distance(materialPool.begin(), std::find(materialPool.begin(), materialPool.end(), mesh.material));
    out.instances.push_back(d);
  }

  out.envWidth = environment.width;
  out.envHeight = environment.height;
  const float* e = reinterpret_cast<const float*>(environment.data.data());
  out.environment.assign(e, e + environment.data.size() * 3);
  out.bind();
}

ptx_camera toPtxCamera(const Camera& c) {
  ptx_camera p{};
  p.position[0] = c.position.x; p.position[1] = c.position.y; p.position[2] = c.position.z;
  p.target[0] = c.target.x; p.target[1] = c.target.y; p.target[2] = c.target.z;
  p.up[0] = c.up.x; p.up[1] = c.up.y; p.up[2] = c.up.z;
  p.fov_deg = c.fieldOfView;
  return p;
}

}  // namespace pt
