#!/usr/bin/env python3
"""Registers, scratch, LDS and spills of every path kernel, from the compiler's own metadata: compiles
csrc/ptx_backend.hip for gfx950 to assembly with the flags of the Makefile (no GPU needed) and prints one JSON
object per kernel; with -DPTX_ISA_MARKERS (pass --regions) also lists the scratch instructions per region of
the loop.  usage: tools/kernel_resources.py [--regions] [extra -D flags]

The shipped radiance kernel sits exactly at the 80-VGPR budget of 6 waves/SIMD: after ANY change to kernels.hpp /
shade.hpp / trace.hpp run this (tests/test_kernel_resources.py does) -- a spill inside the traversal loop costs
more than the sixth wave gains."""
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-I" + os.path.join(REPO, "include")]


def compile_to_asm(extra=()):
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "--cuda-device-only", "-S",
                           os.path.join(REPO, "path-tracer_amd", "csrc", "ptx_backend.hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def kernels(text):
    """{demangled-ish name: {field: int}} from the amdhsa metadata at the end of the assembly."""
    res = {}
    for block in text.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        f = {k: int(re.search(r"\." + k + r":\s+(\d+)", block).group(1))
             for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                       "group_segment_fixed_size")}
        m = re.match(r"_ZN3ptx10pathKernelILi(\d)ELb(\d)ELb(\d)E", name)
        if m:
            name = "pathKernel<%s, %s, %s>" % ("radiance" if m.group(1) == "1" else "counts",
                                               "stats" if m.group(2) == "1" else "plain", "wide" if m.group(3) == "1" else "binary")
        m = re.match(r"_ZN3ptx(\d+)(queueKernel|poolKernel|pacedPoolKernel)ILb(\d)E", name)
        if m:
            name = "%s<%s>" % (m.group(2), "stats" if m.group(3) == "1" else "plain")
        f["waves_per_simd"] = min(8, 512 // max(f["vgpr_count"], 1) if f["vgpr_count"] > 64 else 8)
        res[name] = f
    return res


def scratch_by_region(text, mangled="_ZN3ptx10pathKernelILi1ELb0ELb1E"):
    m = re.search(r"^" + mangled + r".*?:\n(.*?)s_endpgm", text, re.S | re.M)
    region, out = "PRE", []
    for line in m.group(1).splitlines():
        mm = re.search(r"PTXMARK (\w+)", line)
        if mm:
            region = mm.group(1)
        elif "scratch_" in line:
            out.append((region, line.strip()))
    return out


def pool_spill_ops_by_region(text, mangled="_ZN3ptx10poolKernelILb0E"):
    """{region: count} of spill traffic (scratch_*, v_readlane / v_writelane) in poolKernel, regions by its PTXMARKs
    (compile with -DPTX_ISA_MARKERS): PRE, SERVICE, TRAV_BEGIN (the traversal step), TRAV_END."""
    m = re.search(r"^" + mangled + r".*?:\n(.*?)s_endpgm", text, re.S | re.M)
    region, out = "PRE", {}
    for line in m.group(1).splitlines():
        mm = re.search(r"PTXMARK (\w+)", line)
        if mm:
            region = mm.group(1)
        elif "scratch_" in line or "v_readlane_b32" in line or "v_writelane_b32" in line:
            out[region] = out.get(region, 0) + 1
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    regions = "--regions" in args
    extra = [a for a in args if a != "--regions"] + (["-DPTX_ISA_MARKERS"] if regions else [])
    text = compile_to_asm(extra)
    for name, f in kernels(text).items():
        if "Kernel<" in name:
            print(json.dumps({"kernel": name, **f}))
    if regions:
        for region, line in scratch_by_region(text):
            print(f"  {region:12s} {line}")
        print("poolKernel spill traffic by region:", pool_spill_ops_by_region(text))
        print("pacedPoolKernel spill traffic by region:", pool_spill_ops_by_region(text, "_ZN3ptx15pacedPoolKernelILb0E"))
