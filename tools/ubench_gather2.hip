// What does a divergent gather cost on gfx950's vector memory path?  Table resident in L2 (1 MiB).
// Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint32_t pcg(uint32_t& s) {
  s = s * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}
template <int V>
__global__ void __launch_bounds__(256) gather(const float4* __restrict__ table, uint32_t nRecords, int iters, float* out) {
  uint32_t rng = (blockIdx.x * 256u + threadIdx.x) * 9781u + 1u;
  float acc = 0.0f;
  const uint32_t lane = threadIdx.x & 63u;
  for (int it = 0; it < iters; it++) {
    uint32_t rec = pcg(rng) % nRecords;
    if (V == 7) rec = __builtin_amdgcn_readfirstlane(rec);       // all lanes the same record
    if (V == 8) rec = (rec & ~15u) | (lane & 15u);                // 16 consecutive records per 16 lanes (coalesced 1 KiB)
    if (V == 9) rec = (rec & ~1u) | (lane & 1u);                  // lane pairs share a 128-B line
    const float4* p = table + (size_t)rec * 4u;
    const float* pf = reinterpret_cast<const float*>(p);
    if (V == 0 || V >= 7) { const float4 a = p[0], b = p[1], c = p[2], d = p[3]; acc += a.x + b.y + c.z + d.w; }
    if (V == 1) { const float4 a = p[0], b = p[1]; acc += a.x + b.y; }
    if (V == 2) { const float4 a = p[0]; acc += a.x; }
    if (V == 3) { acc += pf[0] + pf[4] + pf[8] + pf[12]; }
    if (V == 4) { if (lane & 1u) { const float4 a = p[0], b = p[1], c = p[2], d = p[3]; acc += a.x + b.y + c.z + d.w; } }
    if (V == 5) { const float4 a = p[0], b = p[1], c = p[2]; acc += a.x + b.y + c.z; }
    if (V == 6) { if (lane < 32u) { const float4 a = p[0], b = p[1], c = p[2], d = p[3]; acc += a.x + b.y + c.z + d.w; } }
  }
  if (acc == 123.456f) out[0] = acc;
}
int main(int argc, char** argv) {
  const size_t kb = argc > 1 ? atoi(argv[1]) : 1024;
  const uint32_t nRecords = (uint32_t)(kb * 1024 / 64);
  float4* table; float* out;
  CHECK(hipMalloc(&table, (size_t)nRecords * 64)); CHECK(hipMalloc(&out, 4)); CHECK(hipMemset(table, 0, (size_t)nRecords * 64));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 2000, waves = 4;
  typedef void (*kern_t)(const float4*, uint32_t, int, float*);
  kern_t ks[10] = {gather<0>, gather<1>, gather<2>, gather<3>, gather<4>, gather<5>, gather<6>, gather<7>, gather<8>, gather<9>};
  const char* names[10] = {"4 x dwordx4 (64 B/lane)", "2 x dwordx4 (32 B)", "1 x dwordx4 (16 B)", "4 x dword (16 B, 4 instr)", "4 x dwordx4, odd lanes only",
                           "3 x dwordx4 (48 B)", "4 x dwordx4, lanes 0-31 only", "4 x dwordx4, all lanes same record", "4 x dwordx4, 16 consecutive records / 16 lanes",
                           "4 x dwordx4, lane pairs share a 128-B line"};
  printf("table %zu KiB, %d CUs, %d waves/SIMD\n", kb, cus, waves);
  for (int v = 0; v < 10; v++) {
    const int grid = cus * waves;
    hipLaunchKernelGGL(ks[v], dim3(grid), dim3(256), 0, 0, table, nRecords, 10, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(ks[v], dim3(grid), dim3(256), 0, 0, table, nRecords, iters, out);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-48s %8.3f ms  %6.1f cycles/wave-iteration/CU\n", names[v], ms, ms * 1e-3 * 2.4e9 / ((double)iters * waves * 4));
  }
  return 0;
}
