// Gather microbenchmark for gfx950: every lane needs one random 64-byte record per iteration (the
// access pattern of the traversal loop).  Variants:
//   A  each lane loads its own record with 4 x global_load_dwordx4 (64 lanes x 4 instructions, every
//      lane-access on its own cache line)
//   B  quad-cooperative: in instruction j, lanes 4m..4m+3 load the four 16-byte quarters of the record
//      needed by lane 16j+m (addresses handed over with ds_bpermute) -- same bytes, but every
//      instruction touches 16 lines instead of 64; data stays in the loading lanes (load side only)
//   C  as B through LDS-DMA (global_load_lds_dwordx4), owners read their record back with ds_read_b128
//   D  as B, records handed to their owners through LDS with ds_write_b128 / ds_read_b128
// Not part of the product.  hipcc -O3 --offload-arch=gfx950 tools/ubench_gather.hip -o tools/bin/ubench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t& s) {
  s = s * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}

template <int VARIANT>
__global__ void __launch_bounds__(256) gather(const float4* __restrict__ table, uint32_t nRecords, int iters, float* out) {
  __shared__ float4 stage[4 * 256];  // 16 KiB: 4 KiB per wave
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  float4* wstage = stage + wave * 256;
  uint32_t rng = (blockIdx.x * 256u + threadIdx.x) * 9781u + 1u;
  float acc = 0.0f;
  for (int it = 0; it < iters; it++) {
    const uint32_t rec = pcg(rng) % nRecords;  // the record this lane needs
    if (VARIANT == 0) {
      const float4* p = table + (size_t)rec * 4u;
      const float4 a = p[0], b = p[1], c = p[2], d = p[3];
      acc += a.x + b.y + c.z + d.w;
    } else if (VARIANT == 1) {
      float s = 0.0f;
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++) {
        const uint32_t owner = 16u * j + (lane >> 2);
        const uint32_t orec = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner * 4u), (int)rec);
        const float4 q = table[(size_t)orec * 4u + (lane & 3u)];
        s += q.x + q.w;
      }
      acc += s;
    } else if (VARIANT == 2) {
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++) {
        const uint32_t owner = 16u * j + (lane >> 2);
        const uint32_t orec = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner * 4u), (int)rec);
        const float4* src = table + (size_t)orec * 4u + (lane & 3u);
        // LDS destination = base + lane * 16: owner m of this instruction gets bytes [64m, 64m+64)
        __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(wstage + 64u * j), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const float4* mine = wstage + 4u * lane;
      const float4 a = mine[0], b = mine[1], c = mine[2], d = mine[3];
      acc += a.x + b.y + c.z + d.w;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      float4 q[4];
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++) {
        const uint32_t owner = 16u * j + (lane >> 2);
        const uint32_t orec = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner * 4u), (int)rec);
        q[j] = table[(size_t)orec * 4u + (lane & 3u)];
      }
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++) wstage[64u * j + lane] = q[j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const float4* mine = wstage + 4u * lane;
      const float4 a = mine[0], b = mine[1], c = mine[2], d = mine[3];
      acc += a.x + b.y + c.z + d.w;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char** argv) {
  const size_t kb = argc > 1 ? atoi(argv[1]) : 65536; const size_t mb = kb / 1024;
  const uint32_t nRecords = (uint32_t)(kb * 1024 / 64);
  float4* table;
  float* out;
  CHECK(hipMalloc(&table, (size_t)nRecords * 64));
  CHECK(hipMalloc(&out, 4));
  CHECK(hipMemset(table, 0, (size_t)nRecords * 64));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 2000;
  typedef void (*kern_t)(const float4*, uint32_t, int, float*);
  kern_t ks[4] = {gather<0>, gather<1>, gather<2>, gather<3>};
  const char* names[4] = {"A per-lane 4 x dwordx4", "B quad-cooperative (load side only)", "C quad-cooperative LDS-DMA + ds_read_b128",
                          "D quad-cooperative regs + ds_write/read_b128"};
  printf("table %zu MiB (%u records of 64 B), %d CUs\n", mb, nRecords, cus);
  for (int waves = 4; waves <= 4; waves *= 2) {
    for (int v = 0; v < 4; v++) {
      const int grid = cus * waves;
      hipLaunchKernelGGL(ks[v], dim3(grid), dim3(256), 0, 0, table, nRecords, 10, out);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(ks[v], dim3(grid), dim3(256), 0, 0, table, nRecords, iters, out);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double recs = (double)grid * 256 * iters;
      printf("waves/SIMD %d  %-48s %8.3f ms  %7.1f G records/s  %7.1f GB/s  %6.1f cycles/wave-gather/CU\n", waves, names[v], ms,
             recs / ms / 1e6, recs * 64 / ms / 1e6, ms * 1e-3 * 2.4e9 / ((double)iters * waves * 4));
    }
  }
  return 0;
}
