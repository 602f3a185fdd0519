#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/profile.sh and the FETCH_SIZE calibration of tools/calib.sh.

Traffic beyond L2 per launch = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024.  FETCH_SIZE tallies every memory-side
read request of the L2 at 64 B, but on gfx950 every such request is a 128-byte line (MI355X_MICROARCH.md, HBM
section: double it).  Checked here for the path kernel's access shape, which the guide leaves uncalibrated
(tools/ubench_fetch_calib.hip, profiles/r02/calib/): a gather of random 64-byte records makes one request per
record, TCC_EA0_RDREQ_128B counts all of them and TCC_EA0_RDREQ_64B none, exactly as for a coalesced stream, and
the path kernel's own passes show the same (pmc_tcc2).  So x2 is the bytes MOVED for this kernel too; a gather
of 64-byte records USES half of them (FETCH_SIZE x 1.0 equals the known useful bytes of the gather), the other
half of each line is useful only when the neighbouring record is wanted while the line is cached.  Infinity-Cache
hits are included in the counter (the scene is 98 MB: most of this traffic never reaches HBM).
The file is stamped with the build id of the library the passes ran (prof_<tag>/build_id.txt, written by tools/profile.sh from
ptx_build_id()), the kernel's name and the stream count: bench.py reports these figures only next to that build.
usage: tools/make_traffic.py gpurun_out/prof_<tag> <spp> <rays per launch> <streams> [profiles/r02/calib/fetch_calibration.json]"""
import json
import os
import subprocess
import sys

root, spp, rays = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
streams = int(sys.argv[4])
calib_path = sys.argv[5] if len(sys.argv) > 5 else None
build_id = open(os.path.join(root, "build_id.txt")).read().strip()
here = os.path.dirname(os.path.abspath(__file__))
s = json.loads(subprocess.check_output([sys.executable, os.path.join(here, "pmc_summary.py"), root]))
c, d = s["counters"], s["derived"]
factor, calib = 2.0, None
if calib_path:
    cal = json.load(open(calib_path))
    # known USEFUL bytes / (FETCH_SIZE * 1024) per access shape, and bytes per memory-side request
    calib = {k: {"useful_bytes_per_fetch_size_byte": round(v["factor"], 4), "useful_bytes_per_request": round(v.get("known_bytes_per_rdreq", 0.0), 2),
                 "requests_128B_share": v.get("requests_128B_share")} for k, v in cal.items() if "factor" in v}
fetch = d["fetch_bytes_raw"] * factor
n128, n64, n32 = c.get("TCC_EA0_RDREQ_128B_sum"), c.get("TCC_EA0_RDREQ_64B_sum"), c.get("TCC_EA0_RDREQ_32B_sum")
out = {
    "workload": f"scene1_1500x800_{spp}spp_6b",
    "kernel": "ptx::poolKernel<false>" if "poolKernel" in s["dispatch"].get("Kernel_Name", "") else "ptx::pathKernel<1, false, true>",
    "build_id": build_id, "sub_streams": streams,
    "rays_per_launch": rays,
    "fetch_size_raw_bytes": d["fetch_bytes_raw"], "fetch_factor": factor,
    "read_requests": {"all": c.get("TCC_EA0_RDREQ_sum"), "128B": n128, "64B": n64, "32B": n32},
    "read_bytes_from_request_sizes": (128.0 * n128 + 64.0 * n64 + 32.0 * n32) if n128 is not None else None,
    "fetch_calibration": calib, "write_bytes": d["write_bytes"],
    "hbm_bytes_per_launch": fetch + d["write_bytes"],
    "hbm_bytes_per_ray": (fetch + d["write_bytes"]) / rays,
    "includes_infinity_cache_hits": True,
    "l2_hit_rate": d.get("l2_hit_rate"),
    "valu_insts_per_launch": c.get("SQ_INSTS_VALU"), "valu_insts_per_ray": c.get("SQ_INSTS_VALU", 0.0) / rays,
    "valu_lane_utilisation": d.get("valu_lane_utilisation"),
    "lane_accesses_per_launch": c.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
    "scratch_size": s["dispatch"].get("Scratch_Size"), "vgprs": s["dispatch"].get("VGPR_Count"), "lds_bytes": s["dispatch"].get("LDS_Block_Size"),
    "source": os.path.basename(root),
}
print(json.dumps(out, indent=1))
