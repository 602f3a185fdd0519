#!/usr/bin/env python3
"""Static instruction counts of the path kernel's traversal iteration, per region, from the gfx950
assembly (compiled with -DPTX_ISA_MARKERS).  usage: tools/count_isa.py [extra -D flags]"""
import os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
       f"-I{REPO}/include", "-fhip-fp32-correctly-rounded-divide-sqrt", "-DPTX_ISA_MARKERS", "--cuda-device-only", "-S",
       f"{REPO}/path-tracer_amd/csrc/ptx_backend.hip", "-o", f"{tmp}/k.s"] + sys.argv[1:]
subprocess.check_call(cmd)
text = open(f"{tmp}/k.s").read()
# ISA_KERNEL=radiance: the phase kernel (experiments/radiance_kernel.hpp); default: pathKernel, radiance, binary (ISA_WIDE=1: wide)
if os.environ.get("ISA_KERNEL") == "radiance":
    pat = r"^_ZN3ptx14radianceKernelILb0EEE.*?:\n(.*?)s_endpgm"
else:
    pat = r"^_ZN3ptx10pathKernelILi1ELb0ELb" + ("1" if os.environ.get("ISA_WIDE") else "0") + r"E.*?:\n(.*?)s_endpgm"
m = re.search(pat, text, re.S | re.M)
lines = m.group(1).splitlines()
region = "PRE"
# issue cost per wave64 instruction per SIMD, measured with tools/ubench_valu2.hip (gpurun_out/ubench_valu2.txt)
FULL = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_lshrrev_b32", "v_not_b32"}
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32"}
cyc = {}
ops = {}
counts = {}
order = []
for ln in lines:
    t = ln.strip()
    mm = re.match(r"; PTXMARK (\w+)", t)
    if mm:
        region = mm.group(1)
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "mem"
    if kind == "valu":
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        w = 2.5 if base in FULL else (8.3 if base in TRANS else 4.3)
        cyc[region] = cyc.get(region, 0.0) + w
        ops.setdefault(region, {}).setdefault(base, 0)
        ops[region][base] += 1
    if region not in counts:
        counts[region] = {"valu": 0, "salu": 0, "mem": 0, "nop": 0, "scratch": 0}
        order.append(region)
    if op == "s_nop":
        counts[region]["nop"] += 1
    counts[region][kind] += 1
    if op.startswith("scratch_"):
        counts[region]["scratch"] += 1
tot = {"valu": 0, "salu": 0, "mem": 0, "nop": 0, "scratch": 0}
for r in order:
    c = counts[r]
    print(f"{r:12s} valu {c['valu']:4d}  salu {c['salu']:4d} (nop {c['nop']:3d})  mem {c['mem']:3d} (scratch {c['scratch']:2d})  valu-cycles {cyc.get(r, 0.0):7.1f}")
    if os.environ.get("ISA_OPS") and r not in ("PRE", "TRAV_END"):
        print("      " + ", ".join(f"{k}:{v}" for k, v in sorted(ops.get(r, {}).items(), key=lambda kv: -kv[1])))
    if r not in ("PRE", "TRAV_END"):
        for k in tot: tot[k] += c[k]
print("iteration (TRAV_BEGIN..TRAV_END):", tot)
