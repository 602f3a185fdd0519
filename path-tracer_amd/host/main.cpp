// PathTracer command line: the reference's main.cpp:240-379 on this repository's host pipeline and
// HIP backend.  Same options (-d -w -h -s -b -t -p -a), defaults, console lines and output.bmp.
//
// -d 0 (default) renders on the GPU through renderGPU -> C ABI -> hand-written HIP kernels.
// -d 1 (the reference's CPU threads path) is NOT part of this program: the CPU restatement of the
// render path is test infrastructure (oracle/), kept out of the product so that no result of this
// binary can come from a CPU fallback.  tests/ and bench.py run it for parity and as the CPU baseline.
#include <getopt.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "colors.hpp"
#include "gpu_render.hpp"
#include "loaders.hpp"
#include "output.hpp"
#include "sample_scenes.hpp"

using namespace pt;

int main(int argc, char** argv) {
  uint width{750};
  uint height{400};
  uint samples{32};
  uint bounces{6};
  uint numThreads{10};
  bool renderBVH{false};
  bool gpuRender{true};
  SampleScene sampleScene{THREE_STL};

  int opt;
  GpuOptions gpuOptions;  // -g / -r: not in the reference
  if (const char* e = getenv("PT_MULTI_SAME_DEVICE")) gpuOptions.sameDevice = e[0] == '1';
  while ((opt = getopt(argc, argv, " w:h:s:b:t:p:ad:g:r:")) != -1) {
    switch (opt) {
      case 'w': width = atoi(optarg); continue;
      case 'h': height = atoi(optarg); continue;
      case 's': samples = atoi(optarg); continue;
      case 'b': bounces = atoi(optarg); continue;
      case 't': numThreads = atoi(optarg); continue;
      case 'p': sampleScene = static_cast<SampleScene>(atoi(optarg)); continue;
      case 'a': renderBVH = true; continue;
      case 'd': gpuRender = atoi(optarg) == 0; continue;
      case 'g': gpuOptions.numGpus = (uint32_t)std::max(1, atoi(optarg)); continue;
      case 'r': gpuOptions.subStreams = (uint32_t)std::max(0, atoi(optarg)); continue;
      default:
        printf("Options\n-d\t0: GPU (default), 1: CPU (not in this binary: see oracle/)\n-w\t<width>\n-h\t<height>\n-s\t<samples>\n"
               "-t\t<threads>\n-b\t<bounces>\n-p\t<0|1|2>\tpreset scene\n-a\trender BVH heat map (only main ray, single sample, no jitter)\n"
               "-g\t<gpus>\tGPUs to split the frame over (samples for radiance, 8x8 tiles for -a); default 1\n"
               "-r\t<streams>\tRNG streams per pixel and GPU (faster on large GPUs; default: the reference's one stream per pixel)\n");
        return EXIT_SUCCESS;
    }
    break;
  }

  if (renderBVH) {  // main.cpp:296-299
    samples = 1u;
    bounces = 1u;
  }
  if (!gpuRender) {
    fprintf(stderr, "-d 1: the CPU render path exists only as the test checker of this repository;\n"
                    "this binary renders on the GPU only and has no CPU fallback.\n");
    return EXIT_FAILURE;
  }
  numThreads = 1u;
  setGpuOptions(gpuOptions);
  if (gpuOptions.numGpus > 1) printf("Rendering using HIP (gfx950) on %u GPUs\n", gpuOptions.numGpus);
  else printf("Rendering using HIP (gfx950)\n");
  printf("\nDimensions: [%u, %u]\tSamples: %u\tBounces: %u\tThreads: %u\n\n", width, height, samples, bounces, numThreads);

  Image image{width, height};
  for (auto& v : image.data) v = vec3{0};

  Camera camera{};
  camera.position = 1.0f * vec3{0.5f, 0.25f, -0.8f};
  camera.target = vec3{0};
  camera.up = normalize(vec3{0, 1, 0});
  camera.fieldOfView = 45.0f;

  // Asset locations: the reference uses paths relative to the working directory (main.cpp:33,
  // sampleScenes.cpp:25,116); PT_MODELS_DIR / PT_ENV_PATH override them.
  SceneAssets assets;
  if (const char* m = getenv("PT_MODELS_DIR")) assets.modelsDir = m;
  if (const char* e = getenv("PT_ENV_PATH")) assets.environmentPath = e;
  if (sampleScene == OBJ && !fileExists(assets.modelsDir + "/viking-room/viking_room.obj") &&
      fileExists("tests/golden/models/viking-room/viking_room.obj"))
    assets.modelsDir = "tests/golden/models";

  std::vector<std::shared_ptr<Geometry>> geometryPool{};
  std::vector<std::shared_ptr<Material>> materialPool{};
  Scene scene{};
  SceneInfo info;
  getScene(sampleScene, assets, scene, geometryPool, materialPool, camera, info);
  Image environment = getEnvironment(assets, info);

  // The output stage (heat-map colouring, clamp, 8-bit quantise, flip: main.cpp:363-374, output.cpp:13-23)
  // runs on the device by default; PT_HOST_OUTPUT=1 takes the reference-shaped route through renderGPU
  // and the host functions (same bytes in output.bmp, 4x the PCIe traffic).
  const char* hostOutput = getenv("PT_HOST_OUTPUT");
  if ((hostOutput && hostOutput[0] == '1') || gpuOptions.numGpus > 1) {  // the multi-GPU entry points return floats / counts
    renderGPU(scene, geometryPool, materialPool, camera, image, environment, samples, (int)bounces, renderBVH);
    if (renderBVH) {  // main.cpp:363-374
      std::vector<uint32_t> counts(image.data.size());
      for (size_t i = 0; i < counts.size(); i++) counts[i] = (uint32_t)image.data[i].x;
      uint32_t mx = heatmapColour(counts.data(), (uint32_t)counts.size(), reinterpret_cast<float*>(image.data.data()));
      printf("Maximum BVH tests: %u\n", mx);
    }
    if (!writeBmp("output.bmp", reinterpret_cast<const float*>(image.data.data()), image.width, image.height)) {
      fprintf(stderr, "cannot write output.bmp\n");
      return EXIT_FAILURE;
    }
  } else {
    std::vector<unsigned char> rows;
    uint32_t maxTests = 0;
    renderGPURows(scene, geometryPool, materialPool, camera, width, height, environment, samples, (int)bounces, renderBVH, rows,
                  maxTests);
    if (renderBVH) printf("Maximum BVH tests: %u\n", maxTests);
    if (!writeBmpRows("output.bmp", rows.data(), width, height)) {
      fprintf(stderr, "cannot write output.bmp\n");
      return EXIT_FAILURE;
    }
  }
  printf("Output result to output.bmp\n");
  return EXIT_SUCCESS;
}
