#!/usr/bin/env python3
"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh.
HBM bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: gfx950's FETCH_SIZE reports half the bytes
of wide (16 B/lane) reads (MI355X_MICROARCH.md, HBM section) -- every read of the path kernel is a
dwordx4 load -- and WRITE_SIZE is exact.  usage: tools/make_traffic.py gpurun_out/prof_<tag> <spp>"""
import json
import subprocess
import sys
import os

root, spp = sys.argv[1], int(sys.argv[2])
here = os.path.dirname(os.path.abspath(__file__))
s = json.loads(subprocess.check_output([sys.executable, os.path.join(here, "pmc_summary.py"), root]))
d = s["derived"]
out = {
    "workload": f"scene1_1500x800_{spp}spp_6b",
    "kernel": "ptx::pathKernel<1, false, true>",
    "hbm_bytes_per_launch": d["fetch_bytes_x2_corrected"] + d["write_bytes"],
    "fetch_size_raw_bytes": d["fetch_bytes_raw"], "fetch_correction": 2.0, "write_bytes": d["write_bytes"],
    "l2_hit_rate": d.get("l2_hit_rate"), "valu_lane_utilisation": d.get("valu_lane_utilisation"),
    "source": os.path.basename(root),
}
print(json.dumps(out, indent=1))
