#!/usr/bin/env python3
"""How concentrated are node steps on few NodePairs?  (GPU box; build: tools/build_variants.sh hist "-DPTX_VISIT_HIST")"""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PTX_BACKEND_LIB"] = os.path.join(REPO, "path-tracer_amd", "lib", "libptx_backend_hist.so")
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd", "py"))
import ptamd
scene = int(sys.argv[1]) if len(sys.argv) > 1 else 1
s = ptamd.HostScene(scene)
d = ptamd.DeviceScene(s.desc_ptr)
p = ptamd.make_params(1500, 800, spp=2, max_bounces=6)
st = d.measure(s.camera, p)
lib = ctypes.CDLL(os.environ["PTX_BACKEND_LIB"])
n = 1 << 20
buf = np.zeros(n, dtype=np.uint32)
assert lib.ptx_debug_visit_hist(buf.ctypes.data_as(ctypes.c_void_p), n) == 0
tot = int(buf.sum(dtype=np.uint64))
print("node steps", st.node_steps, "histogram total", tot, "pairs visited", int((buf > 0).sum()), "tlas nodes", s.info.tlas_nodes)
srt = np.sort(buf)[::-1].astype(np.float64)
cum = np.cumsum(srt) / tot
for k in (32, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 2048, 4096, 16384):
    print(f"  hottest {k:6d} pairs ({k * 64 // 1024:5d} KiB): {cum[k - 1] * 100:6.2f} % of node steps")
ntl = (s.info.tlas_nodes - 1) // 2
print(f"  TLAS pairs [0,{ntl}) carry {buf[:ntl].sum(dtype=np.uint64) / tot * 100:.2f} %")
np.save(os.path.join(REPO, "gpurun_out", f"visit_hist_scene{scene}.npy"), buf[: 1 << 20])
for k in (64, 128, 192, 256, 384):
    print(f"  first {k} pairs of the array (ptx_create's static ordering): {buf[:k].sum(dtype=np.uint64) / tot * 100:.2f} %")
