#!/usr/bin/env python3
"""Writes tools/ubench_mix.hip: issue cost of MIXES of the two VALU classes that tools/ubench_valu2.hip found on gfx950
(plain fp32 / integer ops at ~2.5 cycles per wave64 instruction per SIMD; min/max, compares, selects, conversions and
three-operand integer ops at ~4.3), and of the packed-fp32 instructions.  Answers: do the two classes overlap (time =
max) or add (time = sum)?  Not part of the product.
    python tools/gen_ubench3.py && hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_mix.hip -o tools/bin/ubench_mix
"""
import os

F = "v_fma_f32 %{i}, %16, %17, %{i}"
A = "v_add_f32 %{i}, %16, %{i}"
M = "v_min_f32 %{i}, %16, %{i}"
C = "v_cndmask_b32_e64 %{i}, %16, %{i}, s[20:21]"
CV = "v_cndmask_b32_e32 %{i}, %16, %{i}, vcc"
P = "v_cmp_lt_f32_e64 s[22:23], %16, %{i}"
PV = "v_cmp_lt_f32_e32 vcc, %16, %{i}"
U = "v_cvt_f32_ubyte1_e32 %{i}, %{i}"
PRO = "v_cmp_lt_f32 s[20:21], %16, %0\\n v_cmp_lt_f32 vcc, %17, %0"

def mix(*parts):
    return "\\n ".join(parts), len(parts)


# whole sequences: the four-element sorting network of a wide step (keys %0-%3, payloads %4-%7)
def seq(lines):
    return "\\n ".join(l.strip() for l in lines.strip().splitlines())
SORT_A = seq("""
v_cmp_gt_f32 vcc, %0, %1
s_nop 1
v_cndmask_b32 v200, %0, %1, vcc
v_cndmask_b32 v201, %1, %0, vcc
v_cndmask_b32 v202, %4, %5, vcc
v_cndmask_b32 v203, %5, %4, vcc
v_cmp_gt_f32 vcc, %2, %3
s_nop 1
v_cndmask_b32 v204, %2, %3, vcc
v_cndmask_b32 v205, %3, %2, vcc
v_cndmask_b32 v206, %6, %7, vcc
v_cndmask_b32 v207, %7, %6, vcc
v_cmp_gt_f32 vcc, v200, v204
s_nop 1
v_cndmask_b32 v208, v204, v200, vcc
v_cndmask_b32 %4, v202, v206, vcc
v_cndmask_b32 v209, v206, v202, vcc
v_cmp_gt_f32 vcc, v201, v205
s_nop 1
v_cndmask_b32 v210, v201, v205, vcc
v_cndmask_b32 v211, v203, v207, vcc
v_cndmask_b32 %7, v207, v203, vcc
v_cmp_gt_f32 vcc, v210, v208
s_nop 1
v_cndmask_b32 %5, v211, v209, vcc
v_cndmask_b32 %6, v209, v211, vcc
""")
SORT_B = seq("""
v_cmp_gt_f32 vcc, %0, %1
v_min_f32 v200, %0, %1
v_max_f32 v201, %0, %1
v_cndmask_b32 v202, %4, %5, vcc
v_min_f32 v204, %2, %3
v_cndmask_b32 v203, %5, %4, vcc
v_cmp_gt_f32 vcc, %2, %3
v_max_f32 v205, %2, %3
v_max_f32 v208, v200, v204
v_cndmask_b32 v206, %6, %7, vcc
v_min_f32 v210, v201, v205
v_cndmask_b32 v207, %7, %6, vcc
v_cmp_gt_f32 vcc, v200, v204
v_cmp_gt_f32 s[20:21], v201, v205
v_cmp_gt_f32 s[22:23], v210, v208
v_cndmask_b32 %4, v202, v206, vcc
v_cndmask_b32_e64 v211, v203, v207, s[20:21]
v_cndmask_b32 v209, v206, v202, vcc
v_cndmask_b32_e64 %7, v207, v203, s[20:21]
v_cndmask_b32_e64 %5, v211, v209, s[22:23]
v_cndmask_b32_e64 %6, v209, v211, s[22:23]
""")
SORT_C = SORT_A.replace("v_cmp_gt_f32 vcc,", "v_cmp_gt_f32 s[20:21],").replace(", vcc", ", s[20:21]").replace("v_cndmask_b32 ", "v_cndmask_b32_e64 ")
# E: B with every select e32 on vcc, separated by the min / max of the keys and by v_mov fillers where nothing else is left
SORT_E = seq("""
v_cmp_gt_f32 vcc, %0, %1
v_min_f32 v200, %0, %1
v_max_f32 v201, %0, %1
v_cndmask_b32 v202, %4, %5, vcc
v_min_f32 v204, %2, %3
v_cndmask_b32 v203, %5, %4, vcc
v_cmp_gt_f32 vcc, %2, %3
v_max_f32 v205, %2, %3
v_max_f32 v208, v200, v204
v_cndmask_b32 v206, %6, %7, vcc
v_min_f32 v210, v201, v205
v_cndmask_b32 v207, %7, %6, vcc
v_cmp_gt_f32 vcc, v200, v204
v_cmp_gt_f32 s[20:21], v201, v205
v_cmp_gt_f32 s[22:23], v210, v208
v_cndmask_b32 %4, v202, v206, vcc
v_mov_b32 v212, v212
v_cndmask_b32 v209, v206, v202, vcc
s_mov_b64 vcc, s[20:21]
v_mov_b32 v212, v212
v_mov_b32 v212, v212
v_cndmask_b32 v211, v203, v207, vcc
v_mov_b32 v212, v212
v_cndmask_b32 %7, v207, v203, vcc
s_mov_b64 vcc, s[22:23]
v_mov_b32 v212, v212
v_mov_b32 v212, v212
v_cndmask_b32 %5, v211, v209, vcc
v_mov_b32 v212, v212
v_cndmask_b32 %6, v209, v211, vcc
""")
# (label, template (per accumulator), instructions per copy, prologue, wide: 64-bit accumulators)
OPS = [
    ("fma", *mix(F), PRO, False),
    ("min", *mix(M), PRO, False),
    ("fma min", *mix(F, M), PRO, False),
    ("fma fma min", *mix(F, A, M), PRO, False),
    ("fma fma fma min", *mix(F, A, F, M), PRO, False),
    ("fma min min", *mix(F, M, M), PRO, False),
    ("cndmask_e64", *mix(C), PRO, False),
    ("cndmask_e32 vcc", *mix(CV), PRO, False),
    ("fma cndmask_e64", *mix(F, C), PRO, False),
    ("fma cndmask_e32", *mix(F, CV), PRO, False),
    ("fma fma cndmask_e32", *mix(F, A, CV), PRO, False),
    ("cmp_e64 sgpr", *mix(P), PRO, False),
    ("cmp_e32 vcc", *mix(PV), PRO, False),
    ("fma cmp_e64", *mix(F, P), PRO, False),
    ("fma cmp_e32 cndmask_e32", *mix(F, PV, CV), PRO, False),
    ("fma fma cmp_e32 cndmask_e32", *mix(F, A, PV, CV), PRO, False),
    ("cvt_f32_ubyte1", *mix(U), PRO, False),
    ("fma cvt_f32_ubyte1", *mix(F, U), PRO, False),
    ("fma fma cvt_f32_ubyte1", *mix(F, A, U), PRO, False),
    ("min3_f32", *mix("v_min3_f32 %{i}, %16, %17, %{i}"), PRO, False),
    ("max3_f32", *mix("v_max3_f32 %{i}, %16, %17, %{i}"), PRO, False),
    ("fma min3", *mix(F, "v_min3_f32 %{i}, %16, %17, %{i}"), PRO, False),
    ("mov_b32", *mix("v_mov_b32 %{i}, %16"), PRO, False),
    ("fma mov", *mix(F, "v_mov_b32 %{i}, %{i}"), PRO, False),
    ("mad_u64_u32", *mix("v_mad_u64_u32 %{j}, vcc, %18, %18, %{j}"), PRO, True),
    ("lshl_add_u64", *mix("v_lshl_add_u64 %{j}, %{j}, 2, %{j}"), PRO, True),
    ("pk_fma_f32", *mix("v_pk_fma_f32 %{j}, %19, %20, %{j}"), PRO, True),
    ("pk_mul_f32", *mix("v_pk_mul_f32 %{j}, %19, %{j}"), PRO, True),
    ("pk_add_f32", *mix("v_pk_add_f32 %{j}, %19, %{j}"), PRO, True),
    ("pk_add_f32 neg (a - b)", *mix("v_pk_add_f32 %{j}, %19, %{j} neg_lo:[0,1] neg_hi:[0,1]"), PRO, True),
    ("pk_mov_b32", *mix("v_pk_mov_b32 %{j}, %19, %20"), PRO, True),
    ("pk_fma min(lo)", *mix("v_pk_fma_f32 %{j}, %19, %20, %{j}", "v_min_f32 %{i}, %16, %{i}"), PRO, True),
    ("pk_mul pk_add min(lo) min(lo)", *mix("v_pk_mul_f32 %{j}, %19, %{j}", "v_pk_add_f32 %{j}, %20, %{j}", "v_min_f32 %{i}, %16, %{i}", "v_max_f32 %{i}, %16, %{i}"), PRO, True),
    ("fma + s_mov", *mix(F, "s_mov_b32 s24, s25"), PRO, False),
    ("min + s_mov", *mix(M, "s_mov_b32 s24, s25"), PRO, False),
    ("fma + s_and_b64", *mix(F, "s_and_b64 s[24:25], s[20:21], vcc"), PRO, False),
    ("fma fma + s_and_b64", *mix(F, A, "s_and_b64 s[24:25], s[20:21], vcc"), PRO, False),
    ("fma min + s_and_b64", *mix(F, M, "s_and_b64 s[24:25], s[20:21], vcc"), PRO, False),
    # runs of selects behind one compare, as a compare-exchange of (key, payload) pairs is compiled
    ("cmp_e32 cnd_e32 x2", *mix(PV, CV, CV), PRO, False),
    ("cmp_e32 cnd_e32 x4", *mix(PV, CV, CV, CV, CV), PRO, False),
    ("cmp_e32 cnd fma cnd fma", *mix(PV, CV, F, CV, A), PRO, False),
    ("cmp_e32 cnd fma cnd fma cnd fma cnd fma", *mix(PV, CV, F, CV, A, CV, F, CV, A), PRO, False),
    ("cmp_e32 cnd nop cnd nop cnd nop cnd", *mix(PV, CV, "s_nop 0", CV, "s_nop 0", CV, "s_nop 0", CV), PRO, False),
    ("cnd_e32 cnd_e32 fma", *mix(CV, CV, F), PRO, False),
    ("cnd_e32 fma fma", *mix(CV, F, A), PRO, False),
    ("cnd_e32 mov", *mix(CV, "v_mov_b32 %{i}, %{i}"), PRO, False),
    ("cnd_e32 (other dst)", *mix("v_cndmask_b32_e32 %{i}, %16, %17, vcc"), PRO, False),
    ("cmp_e64 cnd_e64 x4", *mix(P, C.replace("s[20:21]", "s[22:23]"), C.replace("s[20:21]", "s[22:23]"), C.replace("s[20:21]", "s[22:23]"), C.replace("s[20:21]", "s[22:23]")), PRO, False),
    ("min max (keys of a compare-exchange)", *mix("v_min_f32 %{i}, %16, %{i}", "v_max_f32 %{i}, %17, %{i}"), PRO, False),
    ("min max fma fma", *mix("v_min_f32 %{i}, %16, %{i}", F, "v_max_f32 %{i}, %17, %{i}", A), PRO, False),
    ("v_swap_b32", *mix("v_swap_b32 %{i}, %{i2}"), PRO, False),
    ("cmp saveexec swap swap restore", *mix(PV, "s_and_saveexec_b64 s[24:25], vcc", "v_swap_b32 %{i}, %{i2}", "v_swap_b32 %{i}, %{i2}", "s_mov_b64 exec, s[24:25]"), PRO, False),
    ("and_b32 and_b32 or_b32 (select by mask)", *mix("v_and_b32 %{i}, %16, %{i}", "v_and_b32 %{i}, %17, %{i}", "v_or_b32 %{i}, %16, %{i}"), PRO, False),
    ("bfi_b32", *mix("v_bfi_b32 %{i}, %18, %16, %{i}"), PRO, False),
    ("fma bfi", *mix(F, "v_bfi_b32 %{i}, %18, %16, %{i}"), PRO, False),
    ("cmp_e32 + 4 x (cnd_e32 sub)", *mix(PV, CV, "v_sub_f32 %{i}, %16, %{i}", CV, "v_sub_f32 %{i}, %16, %{i}", CV, "v_sub_f32 %{i}, %16, %{i}", CV, "v_sub_f32 %{i}, %16, %{i}"), PRO, False),
    ("max_f32 x2 then cnd (dep)", *mix("v_max_f32 %{i}, %16, %{i}", "v_max_f32 %{i}, %17, %{i}", CV), PRO, False),
    ("cvt fma sub mul (dequantise + slab)", *mix(U, F, "v_sub_f32 %{i}, %{i}, %16", "v_mul_f32 %{i}, %17, %{i}"), PRO, False),
    ("sdwa: fma with byte operand", *mix("v_mul_f32_sdwa %{i}, %{i}, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD"), PRO, False),
    ("cvt_f32_ubyte0 sdwa-free (and + cvt_u32)", *mix("v_and_b32 %{i}, 0xff, %{i}", "v_cvt_f32_u32 %{i}, %{i}"), PRO, False),
    ("SORT A: as compiled (5 cmp, 16 cnd_e32 in runs)", SORT_A, 21, PRO, "seq"),
    ("SORT B: min/max keys, e32 / e64 selects apart", SORT_B, 21, PRO, "seq"),
    ("SORT C: A with e64 selects on an SGPR pair", SORT_C, 21, PRO, "seq"),
    ("SORT E: B, e32 selects only, v_mov fillers", SORT_E, 21, PRO, "seq"),
]
ROUNDS = 8

out = []
out.append("""// GENERATED by tools/gen_ubench3.py -- issue cost of instruction MIXES on gfx950 (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int OP>
__global__ void __launch_bounds__(256) bench(float* out, int iters, float b, float c) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double w0 = a0, w1 = a1, w2 = a2, w3 = a3, w4 = a4, w5 = a5, w6 = a6, w7 = a7, wb = b, wc = c;
  uint32_t ib = threadIdx.x | 1u;
""")
for n, (label, tmpl, per, pro, wide) in enumerate(OPS):
    body = pro + "\\n "
    if wide == "seq":
        body += (tmpl + "\\n ") * (ROUNDS * 8)
    else:
      for r in range(ROUNDS):
        for i in range(8):
            body += tmpl.replace("{i}", str(i)).replace("{i2}", str((i + 4) % 8)).replace("{j}", str(i + 8)) + "\\n "
    out.append(f'  if (OP == {n}) {{ for (int i = 0; i < iters; i++) asm volatile("{body}" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), '
               f'"+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7) '
               f': "v"(b), "v"(c), "v"(ib), "v"(wb), "v"(wc) : "vcc", "scc", "memory", "s20", "s21", "s22", "s23", "s24", "s25", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212"); }}\n')
out.append("""  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7);
  if (r == 123.456f) out[0] = r;
}
typedef void (*kern_t)(float*, int, float, float);
""")
out.append(f"static const int NOPS = {len(OPS)};\n")
out.append("static const char* names[] = {" + ", ".join('"' + o[0] + '"' for o in OPS) + "};\n")
out.append("static const int perCopy[] = {" + ", ".join(str(o[2]) for o in OPS) + "};\n")
out.append("static kern_t table[] = {" + ", ".join(f"bench<{n}>" for n in range(len(OPS))) + "};\n")
out.append(f"""
int main() {{
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate / 1e6;
  printf("%d CUs, clockRate %.3f GHz (cycles below assume this clock)\\n", cus, ghz);
  float* out;
  CHECK(hipMalloc(&out, 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 1500;
  const int waves[] = {{2, 6, 8}};
  printf("%-34s", "cycles per COPY of the mix, per SIMD, at waves:");
  for (int w : waves) printf(" %7d", w);
  printf("   (instructions per copy)\\n");
  for (int op = 0; op < NOPS; op++) {{
    printf("%-34s", names[op]);
    for (int w : waves) {{
      const int grid = cus * w;  // 4 waves per block -> one per SIMD per block
      hipLaunchKernelGGL(table[op], dim3(grid), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(table[op], dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double copies = (double)iters * {ROUNDS * 8} * w;  // per SIMD
      printf(" %7.2f", ms * 1e-3 * ghz * 1e9 / copies);
      fflush(stdout);
    }}
    printf("   (%d)\\n", perCopy[op]);
  }}
  return 0;
}}
""")
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench_mix.hip")
open(path, "w").write("".join(out))
