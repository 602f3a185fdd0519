// Preset scenes 0/1/2 of the reference (sampleScenes.hpp:6-14, sampleScenes.cpp:9-143).
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "scene.hpp"

namespace pt {

enum SampleScene { THREE_STL, SCATTER_STL, OBJ };

struct SceneAssets {
  std::string modelsDir = "models";  // holds bust-of-menelaus.stl and viking-room/
  std::string environmentPath = "environment/evening_road_01_puresky_2k.hdr";
  uint scatterInstances = 0;  // 0 = the reference's 10,000 (tests use fewer)
  uint skyWidth = 0, skyHeight = 0;  // stand-in sky size, 0 = 2048 x 1024
};

struct SceneInfo {
  bool standInGeometry = false;     // procedural bust instead of the Menelaus STL
  bool standInEnvironment = false;  // analytic sky instead of the HDR
};

void getScene(SampleScene sampleScene, const SceneAssets& assets, Scene& scene,
              std::vector<std::shared_ptr<Geometry>>& geometryPool,
              std::vector<std::shared_ptr<Material>>& materialPool, Camera& camera, SceneInfo& info);

Image getEnvironment(const SceneAssets& assets, SceneInfo& info);

}  // namespace pt
