#include "output.hpp"

#include <cstdio>
#include <vector>

#include "colors.hpp"

namespace pt {

uint32_t heatmapColour(const uint32_t* counts, uint32_t n, float* rgb) {
  uint32_t mx = 0;
  for (uint32_t i = 0; i < n; i++) mx = counts[i] > mx ? counts[i] : mx;
  // main.cpp:364-373 works on float(count): max, inverse, afmhot(p.x * inverseMax) where p.x > 0
  float inverseMax = 1.0f / (float)mx;
  for (uint32_t i = 0; i < n; i++) {
    vec3 c(0.0f);
    if (counts[i] > 0) c = afmhot((float)counts[i] * inverseMax);
    rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
  }
  return mx;
}

void quantiseRows(const float* rgb, uint32_t W, uint32_t H, unsigned char* rows) {
  const uint32_t rowBytes = bmpRowBytes(W);
  // BMP stores bottom row first, which is image row 0 here; output.cpp:13-23 flips rows to feed a
  // top-down writer, so the file content is the same.
  for (uint32_t y = 0; y < H; y++) {
    unsigned char* row = rows + (size_t)rowBytes * y;
    for (uint32_t x = 0; x < W; x++) {
      const float* p = &rgb[3 * ((size_t)y * W + x)];
      unsigned char q[3];
      for (int c = 0; c < 3; c++) {
        float v = gclamp(p[c], 0.0f, 1.0f);
        q[c] = (unsigned char)(v * 255.0);  // output.cpp:19-21: float * double, truncated
      }
      row[3 * x + 0] = q[2]; row[3 * x + 1] = q[1]; row[3 * x + 2] = q[0];  // BGR
    }
    for (uint32_t k = 3u * W; k < rowBytes; k++) row[k] = 0;
  }
}

bool writeBmpRows(const std::string& path, const unsigned char* rows, uint32_t W, uint32_t H) {
  const uint32_t rowBytes = bmpRowBytes(W);
  const uint32_t dataBytes = rowBytes * H;
  unsigned char head[54] = {0};
  auto put16 = [&](size_t o, uint32_t v) { head[o] = v & 255; head[o + 1] = (v >> 8) & 255; };
  auto put32 = [&](size_t o, uint32_t v) { put16(o, v & 0xFFFF); put16(o + 2, v >> 16); };
  head[0] = 'B'; head[1] = 'M';
  put32(2, 54 + dataBytes); put32(10, 54);
  put32(14, 40); put32(18, W); put32(22, H); put16(26, 1); put16(28, 24); put32(34, dataBytes);
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  bool ok = fwrite(head, 1, 54, f) == 54 && fwrite(rows, 1, dataBytes, f) == dataBytes;
  fclose(f);
  return ok;
}

bool writeBmp(const std::string& path, const float* rgb, uint32_t W, uint32_t H) {
  std::vector<unsigned char> rows((size_t)bmpRowBytes(W) * H);
  quantiseRows(rgb, W, H, rows.data());
  return writeBmpRows(path, rows.data(), W, H);
}

}  // namespace pt
