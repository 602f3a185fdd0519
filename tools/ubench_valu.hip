// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD for the
// instruction kinds the traversal loop is made of, at 1..8 waves per SIMD.  Decides whether packed
// f32 (v_pk_*) pays for the slab / triangle arithmetic.  Not part of the product.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// 8 independent accumulators, 4 rounds = 32 instructions per asm block
#define R8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define R32(T) R8(T) R8(T) R8(T) R8(T)

#define T_FMA(i)  "v_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define T_MUL(i)  "v_mul_f32 %" #i ", %8, %" #i "\n"
#define T_ADD(i)  "v_add_f32 %" #i ", %8, %" #i "\n"
#define T_MIN(i)  "v_min_f32 %" #i ", %8, %" #i "\n"
#define T_MAX3(i) "v_max3_f32 %" #i ", %8, %9, %" #i "\n"
#define T_AND(i)  "v_and_b32 %" #i ", %8, %" #i "\n"
#define T_ADDU(i) "v_add_u32 %" #i ", %8, %" #i "\n"
#define T_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define T_MOV(i)  "v_mov_b32 %" #i ", %8\n"
#define T_CND(i)  "v_cndmask_b32 %" #i ", %8, %" #i ", vcc\n"
#define T_CMP(i)  "v_cmp_lt_f32 vcc, %8, %" #i "\n"
#define T_CMPS(i) "v_cmp_lt_f32 s[20:21], %8, %" #i "\n"
#define T_RCP(i)  "v_rcp_f32 %" #i ", %" #i "\n"
#define T_MADU(i) "v_mad_u32_u24 %" #i ", %8, %9, %" #i "\n"
#define T_MULLO(i) "v_mul_lo_u32 %" #i ", %8, %" #i "\n"
#define T_PKFMA(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define T_PKMUL(i) "v_pk_mul_f32 %" #i ", %8, %" #i "\n"
#define T_PKADD(i) "v_pk_add_f32 %" #i ", %8, %" #i "\n"
#define T_PKADDN(i) "v_pk_add_f32 %" #i ", %8, %" #i " op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]\n"
#define T_CMPCND(i) "v_cmp_lt_f32 vcc, %8, %" #i "\n v_cndmask_b32 %" #i ", %9, %" #i ", vcc\n"
#define T_SNOP(i) "s_nop 0\n"
#define T_SMOV(i) "s_mov_b32 s20, s21\n"
#define T_VS(i)   "v_add_f32 %" #i ", %8, %" #i "\n s_add_u32 s20, s20, 1\n"

enum { FMA, MUL, ADD, MIN, MAX3, AND, ADDU, LSHL, MOV, CND, CMP, CMPS, RCP, MADU, MULLO, PKFMA, PKMUL, PKADD, PKADDN, CMPCND, SNOP, SMOV, VS, NOPS };
static const char* names[] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_min_f32", "v_max3_f32", "v_and_b32", "v_add_u32", "v_lshlrev_b32",
                              "v_mov_b32", "v_cndmask_b32(vcc)", "v_cmp_lt_f32->vcc", "v_cmp_lt_f32->sgpr", "v_rcp_f32", "v_mad_u32_u24",
                              "v_mul_lo_u32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_add_f32+mods", "v_cmp+v_cndmask (pair)",
                              "s_nop 0", "s_mov_b32", "v_add_f32+s_add_u32 (pair)"};

template <int OP>
__global__ void __launch_bounds__(256) bench(float* out, int iters, float b, float c) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  v2f p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a4}, p4 = {a4, a5}, p5 = {a5, a6}, p6 = {a6, a7}, p7 = {a7, a0};
  v2f pb = {b, c}, pc = {c, b};
#define SCALAR_CASE(E, T) \
  if (OP == E) { for (int i = 0; i < iters; i++) asm volatile(R32(T) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21"); }
#define PACKED_CASE(E, T) \
  if (OP == E) { for (int i = 0; i < iters; i++) asm volatile(R32(T) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc) : "vcc"); }
  SCALAR_CASE(FMA, T_FMA) SCALAR_CASE(MUL, T_MUL) SCALAR_CASE(ADD, T_ADD) SCALAR_CASE(MIN, T_MIN) SCALAR_CASE(MAX3, T_MAX3)
  SCALAR_CASE(AND, T_AND) SCALAR_CASE(ADDU, T_ADDU) SCALAR_CASE(LSHL, T_LSHL) SCALAR_CASE(MOV, T_MOV) SCALAR_CASE(CND, T_CND)
  SCALAR_CASE(CMP, T_CMP) SCALAR_CASE(CMPS, T_CMPS) SCALAR_CASE(RCP, T_RCP) SCALAR_CASE(MADU, T_MADU) SCALAR_CASE(MULLO, T_MULLO)
  SCALAR_CASE(CMPCND, T_CMPCND) SCALAR_CASE(SNOP, T_SNOP) SCALAR_CASE(SMOV, T_SMOV) SCALAR_CASE(VS, T_VS)
  PACKED_CASE(PKFMA, T_PKFMA) PACKED_CASE(PKMUL, T_PKMUL) PACKED_CASE(PKADD, T_PKADD) PACKED_CASE(PKADDN, T_PKADDN)
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y +
            p6.x + p6.y + p7.x + p7.y;
  if (r == 123.456f) out[0] = r;
}

typedef void (*kern_t)(float*, int, float, float);
template <int OP> static kern_t get() { return bench<OP>; }
template <int... I> static void fill(kern_t* t, std::integer_sequence<int, I...>) { ((t[I] = get<I>()), ...); }

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate / 1e6;
  printf("device %s, %d CUs, clockRate %.3f GHz\n", prop.name, cus, ghz);
  float* out;
  CHECK(hipMalloc(&out, 4));
  kern_t table[NOPS];
  fill(table, std::make_integer_sequence<int, NOPS>{});
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 4000;
  const int waves[] = {1, 2, 4, 8};
  printf("%-28s", "cycles/instr/SIMD at waves:");
  for (int w : waves) printf(" %7d", w);
  printf("\n");
  for (int op = 0; op < NOPS; op++) {
    printf("%-28s", names[op]);
    for (int w : waves) {
      const int grid = cus * w;  // 4 waves per block -> one per SIMD per block
      hipLaunchKernelGGL(table[op], dim3(grid), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);  // warm
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(table[op], dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const int perAsm = (op == CMPCND || op == VS) ? 64 : 32;
      const double instr = (double)iters * perAsm * w;  // per SIMD
      printf(" %7.2f", ms * 1e-3 * ghz * 1e9 / instr);
      fflush(stdout);
    }
    printf("\n");
  }
  return 0;
}
