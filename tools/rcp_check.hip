// Exhaustive check (all 2^32 float bit patterns) of cheap reciprocal sequences against the correctly
// rounded 1.0f / x (-fhip-fp32-correctly-rounded-divide-sqrt), per biased exponent of x.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/rcp_check.hip -o tools/bin/rcp_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float rcp1(float x) {  // one Newton step
  const float r = __builtin_amdgcn_rcpf(x);
  const float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcp2(float x) {  // two Newton steps
  float r = __builtin_amdgcn_rcpf(x);
  float e = __builtin_fmaf(-x, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}

__device__ __forceinline__ bool same(float a, float b) {
  return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
}

__global__ void __launch_bounds__(256) check(unsigned long long* bad1, unsigned long long* bad2, uint32_t* firstBad1) {
  // thread t handles bit patterns t, t + stride, ...
  const uint32_t stride = gridDim.x * blockDim.x;
  uint32_t bits = blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t n = 0; n < (1ull << 32) / stride; n++, bits += stride) {
    const float x = __uint_as_float(bits);
    const float ref = 1.0f / x;
    const uint32_t ex = (bits >> 23) & 0xFFu;
    if (!same(rcp1(x), ref)) {
      atomicAdd(&bad1[ex], 1ull);
      atomicMin(&firstBad1[ex], bits & 0x7FFFFFFFu);
    }
    if (!same(rcp2(x), ref)) atomicAdd(&bad2[ex], 1ull);
  }
}

int main() {
  unsigned long long *b1, *b2;
  uint32_t* f1;
  CHECK(hipMalloc(&b1, 256 * 8));
  CHECK(hipMalloc(&b2, 256 * 8));
  CHECK(hipMalloc(&f1, 256 * 4));
  CHECK(hipMemset(b1, 0, 256 * 8));
  CHECK(hipMemset(b2, 0, 256 * 8));
  CHECK(hipMemset(f1, 0xFF, 256 * 4));
  hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, b1, b2, f1);  // 2^20 threads x 4096 values
  CHECK(hipDeviceSynchronize());
  unsigned long long h1[256], h2[256];
  uint32_t hf[256];
  CHECK(hipMemcpy(h1, b1, sizeof(h1), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(h2, b2, sizeof(h2), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hf, f1, sizeof(hf), hipMemcpyDeviceToHost));
  unsigned long long t1 = 0, t2 = 0;
  printf("biased exponent: mismatches of rcp+1 Newton step / rcp+2 steps (of 2^24 patterns each, both signs)\n");
  for (int e = 0; e < 256; e++) {
    t1 += h1[e];
    t2 += h2[e];
    if (h1[e] || h2[e]) printf("  e=%3d  one-step %10llu (first 0x%08x)   two-step %10llu\n", e, h1[e], hf[e], h2[e]);
  }
  printf("total: one-step %llu, two-step %llu\n", t1, t2);
  return 0;
}
