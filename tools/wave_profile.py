#!/usr/bin/env python3
"""Wave-level execution profile of the path kernel's main loop (run on the GPU box).
Needs a profiling build:  tools/build_variants.sh prof "-DPTX_WAVE_PROFILE"
usage: tools/wave_profile.py [--scene 1] [--spp 4] [--lib prof]"""
import argparse, ctypes, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--scene", type=int, default=1)
ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--lib", default="prof")
ap.add_argument("--streams", type=int, default=0)
ap.add_argument("--width", type=int, default=1500)
ap.add_argument("--height", type=int, default=800)
a = ap.parse_args()
os.environ["PTX_BACKEND_LIB"] = os.path.join(REPO, "path-tracer_amd", "lib", f"libptx_backend_{a.lib}.so")
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd", "py"))
import ptamd
s = ptamd.HostScene(a.scene)
d = ptamd.DeviceScene(s.desc_ptr)
p = ptamd.make_params(a.width, a.height, spp=a.spp, max_bounces=6, seed_mode=1 if a.streams > 1 else 0, sub_streams=a.streams)
lib = ctypes.CDLL(os.environ["PTX_BACKEND_LIB"])
buf = (ctypes.c_ulonglong * 32)()
lib.ptx_debug_wave_profile(buf)  # clear
st = d.measure(s.camera, p)
assert lib.ptx_debug_wave_profile(buf) == 0
# experiments/radiance_kernel.hpp (wide layout, default) / kernels.hpp pathKernel (PTX_RAD_LEGACY=1 or PTX_WIDE=0)
legacy = os.environ.get("PTX_RAD_LEGACY") == "1" or os.environ.get("PTX_WIDE") == "0"
names = (["loop", "shade", "step", "leaving", "enter", "tri", "node", "node-exact"] if legacy else
         ["loop", "shade", "node", "tri", "enter", "tail only", "deep stack", "exit (in tail)"])
iters = buf[0]
out = {"rays": st.rays, "node_steps": st.node_steps, "tri_tests": st.tri_tests, "inst_entries": st.inst_entries, "render_s": st.render_s}
print(json.dumps(out))
print(f"{'body':12s} {'iterations':>14s} {'frac of loop':>13s} {'lanes/iter':>11s} {'lane total':>14s}")
for k, n in enumerate(names):
    it, ln = buf[2 * k], buf[2 * k + 1]
    print(f"{n:12s} {it:14d} {it / max(iters, 1):13.3f} {ln / max(it, 1):11.2f} {ln:14d}")
tn = ["shade+checks", "decode+issue", "wait (vmcnt/lgkmcnt)", "exit", "enter", "tri", "node", "merge"]
if os.environ.get("PTX_EXCHANGE") == "1":
    tn = ["loop top + shading", "post", "barrier after post", "serve: lists, request, fetch issue", "serve: wait for the record", "serve: compute + result", "barrier after serve", "merge"]
tt = sum(buf[16 + k] for k in range(8))
if tt:
    print(f"\nwall-clock ticks per region, summed over waves (total {tt:.4g}; per loop iteration {tt / max(iters, 1):.0f})")
    for k, n in enumerate(tn):
        print(f"  {n:24s} {buf[16 + k] / tt:7.3f}   {buf[16 + k] / max(iters, 1):8.1f} ticks/iteration")

if not legacy and hasattr(lib, "ptx_debug_sp_hist"):
    h = (ctypes.c_ulonglong * 128)()
    if lib.ptx_debug_sp_hist(h) == 0:
        for name, base in (("node", 0), ("tri", 64)):
            tot = sum(h[base + i] for i in range(64))
            acc = 0
            print(f"\nstack pointer at {name} steps (entries in use incl. bottom marker): sp  share  cumulative")
            for i in range(64):
                if h[base + i]:
                    acc += h[base + i]
                    print(f"  {i:3d} {h[base + i] / max(tot, 1):8.4f} {acc / max(tot, 1):8.4f}")
