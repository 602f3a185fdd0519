#!/usr/bin/env python3
"""Copies what is judged from a tools/profile.sh output tree into profiles/<round>/: the kernel-trace statistics,
the path kernel's rows of every PMC pass, and the summary.  usage: tools/collect_profile.py gpurun_out/prof_r02 profiles/r02 100"""
import csv
import glob
import os
import shutil
import subprocess
import sys

src, dst, spp = sys.argv[1], sys.argv[2], sys.argv[3]
here = os.path.dirname(os.path.abspath(__file__))
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    rows, header = [], None
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        r = list(csv.reader(open(f)))
        header = r[0]
        rows += [x for x in r[1:] if "pathKernel" in x[header.index("Kernel_Name")] or "poolKernel" in x[header.index("Kernel_Name")]]
    if header:
        with open(os.path.join(dst, os.path.basename(d) + "_radianceKernel.csv"), "w", newline="") as o:
            w = csv.writer(o)
            w.writerow(header)
            w.writerows(rows)
with open(os.path.join(dst, f"pmc_summary_{spp}spp.json"), "w") as o:
    o.write(subprocess.check_output([sys.executable, os.path.join(here, "pmc_summary.py"), src], text=True))
print(sorted(os.listdir(dst)))
