#!/usr/bin/env python3
"""FETCH_SIZE calibration: joins the known byte counts printed by tools/ubench_fetch_calib with rocprofv3's counter
CSVs of the same run.  usage: tools/calibrate_fetch.py <stdout of the ubench> <dir with pmc csv trees> > fetch_calibration.json

factor = known bytes / (FETCH_SIZE * 1024): what FETCH_SIZE has to be multiplied with for that access shape.
The symbol -> case map follows the template arguments in tools/ubench_fetch_calib.hip."""
import csv
import glob
import json
import os
import re
import sys

log, root = sys.argv[1], sys.argv[2]
known = {}
for line in open(log):
    m = re.match(r"CALIB (\S+) known_bytes (\d+) ms ([0-9.]+) useful_GBps ([0-9.]+)", line)
    if m:
        known[m.group(1)] = {"known_bytes": float(m.group(2)), "ms": float(m.group(3)), "useful_GBps": float(m.group(4))}
symbol = {"streamKernel<0>": "stream_mall96", "streamKernel<1>": "stream_hbm2g",
          "gatherKernel<0, 4>": "gather64_mall96", "gatherKernel<0, 3>": "gather48_mall96",
          "gatherKernel<1, 4>": "gather64_hbm2g", "gatherKernel<1, 3>": "gather48_hbm2g"}
counters = {}
for f in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        for sym, case in symbol.items():
            if sym in r["Kernel_Name"]:
                counters.setdefault(case, {}).setdefault(r["Counter_Name"], 0.0)
                counters[case][r["Counter_Name"]] += float(r["Counter_Value"])
out = {}
for case, k in known.items():
    c = counters.get(case, {})
    e = dict(k)
    e["counters"] = c
    if c.get("FETCH_SIZE"):
        e["fetch_size_bytes"] = c["FETCH_SIZE"] * 1024
        e["factor"] = k["known_bytes"] / (c["FETCH_SIZE"] * 1024)
    if c.get("TCC_EA0_RDREQ_sum"):
        e["known_bytes_per_rdreq"] = k["known_bytes"] / c["TCC_EA0_RDREQ_sum"]
    if c.get("TCC_EA0_RDREQ_128B_sum") is not None:
        n = c["TCC_EA0_RDREQ_128B_sum"] + c.get("TCC_EA0_RDREQ_64B_sum", 0.0) + c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        e["requests_128B_share"] = c["TCC_EA0_RDREQ_128B_sum"] / n if n else None
        e["moved_bytes"] = 128.0 * c["TCC_EA0_RDREQ_128B_sum"] + 64.0 * c.get("TCC_EA0_RDREQ_64B_sum", 0.0) + 32.0 * c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        e["moved_GBps"] = e["moved_bytes"] / (k["ms"] * 1e-3) / 1e9
    out[case] = e
print(json.dumps(out, indent=1))
