// Post + output stage of the reference (host): heat-map colouring (main.cpp:363-374) and the BMP
// writer (output.cpp:8-28; the reference calls stb_image_write, the 24-bit BMP container is written
// directly here).
#pragma once

#include <cstdint>
#include <string>

namespace pt {

// Returns the maximum count; rgb[3*i..] = afmhot(count/max) where count > 0, else (0,0,0).
uint32_t heatmapColour(const uint32_t* counts, uint32_t nPixels, float* rgb);

// rgb: width*height*3 floats, row 0 = bottom scan line.  clamp -> (uint8_t)(c * 255.0) -> BMP rows.
bool writeBmp(const std::string& path, const float* rgb, uint32_t width, uint32_t height);

// The two halves of writeBmp, for callers that receive the pixel array ready-made from the device-side
// output stage (ptx_render_bgr8 / ptx_render_counts_bgr8): rows = bmpRowBytes(width) * height bytes,
// bottom scan line first, B G R per pixel, rows zero-padded to a multiple of 4.
inline uint32_t bmpRowBytes(uint32_t width) { return (width * 3u + 3u) & ~3u; }
void quantiseRows(const float* rgb, uint32_t width, uint32_t height, unsigned char* rows);
bool writeBmpRows(const std::string& path, const unsigned char* rows, uint32_t width, uint32_t height);

}  // namespace pt
