#include "sample_scenes.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "colors.hpp"
#include "loaders.hpp"
#include "rng.hpp"

namespace pt {

namespace {

std::shared_ptr<Geometry> bustGeometry(const SceneAssets& assets, SceneInfo& info) {
  std::vector<Triangle> triangles;
  std::string stl = assets.modelsDir + "/bust-of-menelaus.stl";
  if (fileExists(stl)) triangles = loadSTL(stl);  // sampleScenes.cpp:25,71
  if (triangles.empty()) {
    triangles = standInBust();
    info.standInGeometry = true;
  }
  auto g = std::make_shared<Geometry>(std::move(triangles));
  if (!g_quiet) {  // geometry.cpp:65-66
    printf("BVH build time: %g s\n", std::floor(g->buildSeconds * 1e4) / 1e4);
    printf("Nodes: %zu\n", g->bvh.size());
  }
  return g;
}

}  // namespace

// sampleScenes.cpp:9-143.  Camera, materials, transforms and the order of RNG draws follow the
// reference presets; angles are double expressions narrowed to float at the setter call, as there.
void getScene(SampleScene sampleScene, const SceneAssets& assets, Scene& scene,
              std::vector<std::shared_ptr<Geometry>>& geometryPool,
              std::vector<std::shared_ptr<Material>>& materialPool, Camera& camera, SceneInfo& info) {
  auto start = std::chrono::steady_clock::now();
  const double PI = M_PI;

  switch (sampleScene) {
    case THREE_STL: {
      camera.position = 200.0f * vec3{-1.0f, 0.2f, 0.05f};
      camera.target = vec3{0};
      camera.up = normalize(vec3{0, 1, 0});
      camera.fieldOfView = 45.0f;

      geometryPool.emplace_back(bustGeometry(assets, info));

      materialPool.emplace_back(std::make_shared<Material>());
      materialPool[0]->albedo = vec3{1.0f, 0.8f, 0.6f};
      materialPool[0]->metalness = 1.0f;
      materialPool[0]->roughness = 0.05f;
      materialPool.emplace_back(std::make_shared<Material>());
      materialPool[1]->albedo = vec3{1.0f};
      materialPool.emplace_back(std::make_shared<Material>());
      materialPool[2]->albedo = vec3{0.1f, 0.7f, 0.4f};
      materialPool[2]->metalness = 1.0f;
      materialPool[2]->roughness = 0.2f;

      scene.meshes.emplace_back(Mesh{geometryPool[0], materialPool[0]});
      scene.meshes[0].setRotationX((float)(-0.5f * PI));
      scene.meshes[0].setRotationZ((float)(0.0f * PI));
      scene.meshes[0].setPosition(vec3{0.0f, 0.0f, -90.0f});
      scene.meshes[0].setScale(0.85f);

      scene.meshes.emplace_back(Mesh{geometryPool[0], materialPool[1]});
      scene.meshes[1].setRotationX((float)(-0.5f * PI));
      scene.meshes[1].setRotationZ((float)(0.0f * PI));

      scene.meshes.emplace_back(Mesh{geometryPool[0], materialPool[2]});
      scene.meshes[2].setRotationX((float)(-0.5f * PI));
      scene.meshes[2].setRotationZ((float)(-0.3f * PI));
      scene.meshes[2].setPosition(vec3{0.0f, 0.0f, 90.0f});
      scene.meshes[2].setScale((float)0.8);
    } break;

    case SCATTER_STL: {
      camera.position = 500.0f * vec3{1.0f, 0.2f, -0.15f};
      camera.target = vec3{0};
      camera.up = normalize(vec3{0, 1, 0});
      camera.fieldOfView = 45.0f;

      geometryPool.emplace_back(bustGeometry(assets, info));

      materialPool.emplace_back(std::make_shared<Material>());
      materialPool[0]->albedo = vec3{1.0f, 0.8f, 0.6f};
      materialPool[0]->metalness = 1.0f;
      materialPool[0]->roughness = 0.05f;
      materialPool.emplace_back(std::make_shared<Material>());
      materialPool[1]->albedo = vec3{1.0f};

      // The loop appends ten materials but recolours entries 0..9 (sampleScenes.cpp:85-91): the pool
      // ends with 12 entries, entry 0 keeps roughness 0.05, entries 10 and 11 stay default white.
      uint rngState{7142u};
      uint materialCount = 10u;
      for (uint i = 0; i < materialCount; i++) {
        materialPool.emplace_back(std::make_shared<Material>());
        materialPool[i]->albedo = hsv(static_cast<float>(i) / static_cast<float>(materialCount));
        materialPool[i]->metalness = getRandomFloat(rngState) > 0.7f;
      }

      const uint instances = assets.scatterInstances ? assets.scatterInstances : 10000u;
      scene.meshes.reserve(instances);
      for (uint i = 0u; i < instances; i++) {
        // floor(float * size_t) in float; a draw of exactly 1.0f would index past the pool in the
        // reference (undefined behaviour, p = 3e-8 per draw) -- clamped here.
        uint materialIdx = (uint)floorf(getRandomFloat(rngState) * (float)materialPool.size());
        if (materialIdx >= materialPool.size()) materialIdx = (uint)materialPool.size() - 1u;
        Mesh m;
        m.geometry = geometryPool[0];
        m.material = materialPool[materialIdx];
        // Every setter re-runs update() from scratch (mesh.cpp:43-66), so only the final state
        // matters; the draws are taken in the reference's order and update() runs once.
        m.rotation.x = (float)(-0.5f * PI);
        m.rotation.y = (float)(2.0f * PI * getRandomFloat(rngState));
        m.rotation.z = (float)(PI * getRandomFloat(rngState));
        m.scale = 0.2f;
        m.translation = 700.0f * (2.0f * getRandomVec3(rngState) - 1.0f);
        m.update();
        scene.meshes.emplace_back(std::move(m));
      }
    } break;

    case OBJ: {
      camera.position = 1.0f * vec3{0.5f, 0.35f, -1.0f};
      camera.target = vec3{0};
      camera.up = normalize(vec3{0, 1, 0});
      camera.fieldOfView = 45.0f;

      std::vector<vec3> normals{};
      std::vector<vec2> texCoords{};
      std::vector<Triangle> triangles = loadObj(assets.modelsDir + "/viking-room/viking_room.obj", normals, texCoords);
      VertexAttributes attributes{normals, texCoords};
      auto g = std::make_shared<Geometry>(std::move(triangles), std::move(attributes));
      if (!g_quiet) {
        printf("BVH build time: %g s\n", std::floor(g->buildSeconds * 1e4) / 1e4);
        printf("Nodes: %zu\n", g->bvh.size());
      }
      geometryPool.emplace_back(std::move(g));

      materialPool.emplace_back(std::make_shared<Material>());
      materialPool[0]->albedoTexture = loadImage(assets.modelsDir + "/viking-room/albedo.png");
      materialPool[0]->emissiveTexture = loadImage(assets.modelsDir + "/viking-room/emissive.png");
      materialPool[0]->emissive = vec3{1};

      scene.meshes.emplace_back(Mesh{geometryPool[0], materialPool[0]});
      scene.meshes[0].setRotationX((float)(-0.5f * PI));
      scene.meshes[0].setRotationZ((float)(0.15f * PI));
    } break;

    default:
      printf("Scene %d is not defined\n", (int)sampleScene);
      exit(1);
  }
  std::chrono::duration<double> duration = std::chrono::steady_clock::now() - start;
  if (!g_quiet) printf("\nObject processing time: %g s\n", std::floor(duration.count() * 1e4) / 1e4);

  scene.quiet = g_quiet;
  scene.completeScene();
}

// main.cpp:33 loads the HDR in a static initialiser and crashes when it is missing; here a missing
// file selects the analytic stand-in and says so.
Image getEnvironment(const SceneAssets& assets, SceneInfo& info) {
  if (!assets.environmentPath.empty() && fileExists(assets.environmentPath)) {
    Image env = loadEnvironmentImage(assets.environmentPath);
    if (env.width > 0) return env;
  }
  info.standInEnvironment = true;
  if (!g_quiet) printf("Environment map %s not found: using the analytic stand-in sky\n", assets.environmentPath.c_str());
  return standInSky(assets.skyWidth ? assets.skyWidth : 2048u, assets.skyHeight ? assets.skyHeight : 1024u);
}

}  // namespace pt
