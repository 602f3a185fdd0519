#!/usr/bin/env python3
"""Static instruction counts of the path kernel's traversal iteration, per region, from the gfx950
assembly (compiled with -DPTX_ISA_MARKERS).  usage: tools/count_isa.py [extra -D flags]"""
import os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
       f"-I{REPO}/include", "-fhip-fp32-correctly-rounded-divide-sqrt", "-DPTX_ISA_MARKERS", "--cuda-device-only", "-S",
       f"{REPO}/path-tracer_amd/csrc/ptx_backend.hip", "-o", f"{tmp}/k.s"] + sys.argv[1:]
subprocess.check_call(cmd)
text = open(f"{tmp}/k.s").read()
m = re.search(r"^_ZN3ptx10pathKernelILi1ELb0ELb0E.*?:\n(.*?)s_endpgm", text, re.S | re.M)
lines = m.group(1).splitlines()
region = "PRE"
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
    if region not in counts:
        counts[region] = {"valu": 0, "salu": 0, "mem": 0, "nop": 0}
        order.append(region)
    if op == "s_nop":
        counts[region]["nop"] += 1
    counts[region][kind] += 1
tot = {"valu": 0, "salu": 0, "mem": 0, "nop": 0}
for r in order:
    c = counts[r]
    print(f"{r:12s} valu {c['valu']:4d}  salu {c['salu']:4d} (nop {c['nop']:3d})  mem {c['mem']:3d}")
    if r not in ("PRE", "TRAV_END"):
        for k in tot: tot[k] += c[k]
print("iteration (TRAV_BEGIN..TRAV_END):", tot)
