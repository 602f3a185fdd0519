// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the path kernel's access shape (not part of the product).
//
// The guide (MI355X_MICROARCH.md, HBM) calibrates FETCH_SIZE only for wide coalesced streams (it reports half
// of their bytes).  The path kernel reads 64-byte records at random (one record = four dwordx4 loads of one
// lane), so the factor is measured here on kernels whose byte counts are known:
//   stream     every lane reads consecutive 16-B words, the table once       known = table bytes
//   gather64   every lane reads a random 64-B record (4 x dwordx4)           known = accesses * 64 B
//   gather48   every lane reads the first 48 B of a random 64-B record       known = accesses * 48 B
// each over a table far larger than L2 (32 MiB aggregate) so that (almost) every access leaves L2:
//   96 MiB  -- stays in the 256-MiB Infinity Cache: what the path kernel's 98-MB scene does
//   2 GiB   -- exceeds it: HBM
// One launch per kernel, each its own symbol, so that rocprofv3 --pmc rows can be matched by name.  Prints the
// known bytes and the time of each launch; tools/calibrate_fetch.py joins them with the counter CSVs.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint32_t pcg(uint32_t& s) {
  s = s * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}
template <int TABLE>  // TABLE only makes distinct symbols
__global__ void __launch_bounds__(256) streamKernel(const float4* __restrict__ table, size_t nWords, float* out) {
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < nWords; i += (size_t)gridDim.x * 256u) acc += table[i].x;
  if (acc == 123.456f) out[0] = acc;
}
template <int TABLE, int QUARTERS>
__global__ void __launch_bounds__(256) gatherKernel(const float4* __restrict__ table, uint32_t nRecords, int iters, float* out) {
  uint32_t rng = (blockIdx.x * 256u + threadIdx.x) * 9781u + 1u;
  float acc = 0.0f;
  for (int it = 0; it < iters; it++) {
    const uint32_t rec = pcg(rng) % nRecords;
    const float4* p = table + (size_t)rec * 4u;
    const float4 a = p[0], b = p[1], c = p[2];
    acc += a.x + b.y + c.z;
    if (QUARTERS == 4) acc += p[3].w;
  }
  if (acc == 123.456f) out[0] = acc;
}
template <typename F>
void timed(const char* name, double knownBytes, F launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, 0));
  launch();
  CHECK(hipGetLastError());
  CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("CALIB %s known_bytes %.0f ms %.3f useful_GBps %.1f\n", name, knownBytes, ms, knownBytes / (ms * 1e-3) / 1e9);
}
int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int grid = prop.multiProcessorCount * 8;
  float* out; CHECK(hipMalloc(&out, 4));
  const size_t small = 96ull << 20, big = 2ull << 30;
  float4* table; CHECK(hipMalloc(&table, big)); CHECK(hipMemset(table, 0, big));
  CHECK(hipDeviceSynchronize());
  const int iters = 400;
  const double accesses = (double)grid * 256 * iters;
  // warm the Infinity Cache with the small table (first 96 MiB), untimed, under its own name
  streamKernel<9><<<grid, 256>>>(table, small / 16, out);
  CHECK(hipDeviceSynchronize());
  timed("stream_mall96", (double)small, [&] { streamKernel<0><<<grid, 256>>>(table, small / 16, out); });
  timed("gather64_mall96", accesses * 64, [&] { gatherKernel<0, 4><<<grid, 256>>>(table, (uint32_t)(small / 64), iters, out); });
  timed("gather48_mall96", accesses * 48, [&] { gatherKernel<0, 3><<<grid, 256>>>(table, (uint32_t)(small / 64), iters, out); });
  timed("stream_hbm2g", (double)big, [&] { streamKernel<1><<<grid, 256>>>(table, big / 16, out); });
  timed("gather64_hbm2g", accesses * 64, [&] { gatherKernel<1, 4><<<grid, 256>>>(table, (uint32_t)(big / 64), iters, out); });
  timed("gather48_hbm2g", accesses * 48, [&] { gatherKernel<1, 3><<<grid, 256>>>(table, (uint32_t)(big / 64), iters, out); });
  return 0;
}
