// Asset ingest for the host pipeline: the four loaders of the reference's input.hpp:9-16 with the
// same names and results (triangles with centroids; per-corner normals / flipped uvs; 8-bit images
// scaled by 1/255; Radiance HDR as float RGB) plus deterministic procedural stand-ins for the two
// assets that are absent from the reference checkout (.MISSING_LARGE_BLOBS).
//
// The reference delegates parsing to vendored third-party headers (stb_image 2.28, stl_reader,
// tinyobjloader 2.0).  Those are not copied here; the parts of their behaviour that change results
// are restated: tinyobjloader's own decimal parser (tiny_obj_loader.h:890-1027), stl_reader's exact
// vertex de-duplication followed by degenerate-triangle removal (stl_reader.h:498-521), stb's
// RGBE -> float conversion (stb_image.h:7130-7155) and 8-bit PNG decoding (zlib inflate + the five
// PNG filters; alpha dropped for STBI_rgb).
#pragma once

#include <string>
#include <vector>

#include "scene.hpp"

namespace pt {

extern bool g_quiet;  // suppress the reference's progress prints (library / test use)

bool fileExists(const std::string& path);

Image loadImage(const std::string& path);             // input.cpp:132-147; width 0 on failure
Image loadEnvironmentImage(const std::string& path);  // input.cpp:117-130; width 0 on failure
// input.cpp:12-33; empty on failure.  Binary files are memory-mapped and converted in one pass;
// stl_reader's sort-based vertex merge (stl_reader.h:464) is not repeated: the only thing the reference
// keeps from it is "a triangle that names one vertex twice is dropped", which is decided per triangle.
std::vector<Triangle> loadSTL(const std::string& path);
// Writes a binary STL (80-byte header, count, 50 bytes per facet with a zero normal): the synthetic
// stand-in of the absent scan as a FILE, so that the ingest path can be exercised and timed.
bool writeBinarySTL(const std::string& path, const std::vector<Triangle>& tris);
std::vector<Triangle> loadObj(const std::string& path, std::vector<vec3>& normals,
                              std::vector<vec2>& texCoords);  // input.cpp:35-114

// tinyobjloader's decimal parser (exposed for tests).
bool parseDecimal(const char* s, const char* s_end, double* result);

// Stand-ins (labelled as such wherever they are used).
// 505,848-triangle closed bust-like surface, z-up, bounding box about 110 x 110 x 160, fixed
// analytic displacement -- the triangle count of the Menelaus scan after stl_reader (README.md:58).
std::vector<Triangle> standInBust(uint triangleTarget = 505848u);
// 2048 x 1024 analytic evening sky: gradient + sun lobe (peak radiance about 50).
Image standInSky(uint width = 2048u, uint height = 1024u);

}  // namespace pt
