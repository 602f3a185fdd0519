#!/usr/bin/env python3
"""Summarises rocprofv3 output of tools/profile.sh: per-kernel time from the kernel trace and the PMC
counters of the path kernel.  usage: tools/pmc_summary.py gpurun_out/prof_<tag>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(root):
    out = {}
    for f in glob.glob(os.path.join(root, "trace", "*", "*_kernel_stats.csv")):
        rows = list(csv.DictReader(open(f)))
        out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")} for r in rows]
    counters = defaultdict(float)
    meta = {}
    for f in glob.glob(os.path.join(root, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "pathKernel<1, false" not in r["Kernel_Name"] and "poolKernel<false>" not in r["Kernel_Name"]:
                continue  # the un-instrumented radiance kernel only (poolKernel, or pathKernel with PTX_QUEUE=0)
            meta["Kernel_Name"] = r["Kernel_Name"]
            counters[r["Counter_Name"]] += float(r["Counter_Value"])
            meta.update({k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size") if k in r})
    out["counters"] = dict(counters)
    out["dispatch"] = meta
    c = counters
    d = {}
    if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
        d["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0) if c["SQ_ACTIVE_INST_VALU"] else None
    if c.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS"):
            if k in c:
                d[k + "/WAVE_CYCLES"] = c[k] / c["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in c:
        d["fetch_bytes_x2_corrected"] = c["FETCH_SIZE"] * 1024 * 2  # KiB units; gfx950 reports half of wide reads
        d["fetch_bytes_raw"] = c["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in c:
        d["write_bytes"] = c["WRITE_SIZE"] * 1024
    out["derived"] = d
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
