#!/usr/bin/env python3
"""Where a wave of poolKernel spends its time (a -DPTX_POOL_PROFILE build: PTX_BACKEND_LIB=.../libptx_backend_pprof.so).
usage: tools/pool_profile.py [spp] [streams] [scene]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "path-tracer_amd", "py"))
import ptamd

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 34
scene = int(sys.argv[3]) if len(sys.argv) > 3 else 1
os.environ["PTX_QUEUE"] = "1"
s = ptamd.HostScene(scene, cache_dir="/tmp/pt_scene_cache")
d = ptamd.DeviceScene(s.desc_ptr)
p = ptamd.make_params(1500, 800, spp=spp, max_bounces=6, seed_mode=1 if streams > 1 else 0, sub_streams=streams)
lib = ptamd.backend_lib()
out = (C.c_ulonglong * 32)()
img, st = d.render(s.camera, p)
lib.ptx_debug_pool_profile(out)  # clear what the warm-up left
img, st = d.render(s.camera, p)
lib.ptx_debug_pool_profile(out)
v = list(out)
ticks = sum(v[:5])
names = ["loop top", "delivery", "shading batch", "refill", "traversal step"]
print(f"poolKernel, scene {scene}, 1500x800, {spp} spp, {streams} streams, chains/lane {os.environ.get('PTX_POOL_CHAINS', '3')}: {st.render_s * 1e3:.1f} ms")
for n, t in zip(names, v[:5]):
    print(f"  {n:16s} {100.0 * t / ticks:5.1f} % of wave time")
it, sv, ba, sh, dl, tk, ls, vac = v[8:16]
print(f"  iterations {it}, services {sv} (one per {it / max(sv, 1):.1f} iterations), batches {ba} with {sh / max(ba, 1):.1f} chains, "
      f"rays delivered {dl} ({dl / max(sv, 1):.1f} per service), taken {tk}")
print(f"  lanes stepping per iteration {ls / max(it, 1):.1f}, lanes without a ray per iteration {vac / max(it, 1):.1f}")
print(f"  ticks per iteration: traversal {v[4] / max(it, 1):.0f}; per service: delivery {v[1] / max(sv, 1):.0f}, refill {v[3] / max(sv, 1):.0f}; per batch: {v[2] / max(ba, 1):.0f}")
longest, waves, sq = v[16], v[17], v[18]
if waves:
    mean = ticks / waves
    var = sq * 1024.0 * 1024.0 / waves - mean * mean
    print(f"  waves {waves}: mean duration {mean:.4g} ticks, longest {longest:.4g} ({longest / mean:.3f} x mean), standard deviation {max(var, 0.0) ** 0.5 / mean:.3f} of the mean")
if waves and hasattr(lib, "ptx_debug_pool_waves"):
    import numpy as np
    n = int(min(waves, 8192))
    w = np.zeros((n, 8), dtype=np.uint64)
    lib.ptx_debug_pool_waves(w.ctypes.data_as(C.c_void_p), C.c_uint(n))
    w = w.astype(np.float64)
    dur, its, steps, batches, shaded, trav, start, services = w.T
    order = np.argsort(-dur)
    t0 = start.min()
    print("  the ten longest waves and the median wave (ticks, iterations, ticks per iteration, lanes stepping per iteration, batches, chains per batch, start):")
    for i in list(order[:10]) + [order[n // 2]]:
        print(f"    wave {i:5d}: {dur[i]:.4g} ticks, {its[i]:.4g} iterations, {dur[i] / its[i]:7.0f} ticks each, {steps[i] / its[i]:5.1f} lanes, "
              f"{batches[i]:.0f} batches of {shaded[i] / max(batches[i], 1):.1f}, started at {start[i] - t0:.3g}")
    q = np.quantile(dur, [0.01, 0.1, 0.5, 0.9, 0.99, 1.0])
    print("  duration quantiles 1/10/50/90/99/100 %:", " ".join(f"{x:.4g}" for x in q))
    print("  iterations quantiles:", " ".join(f"{x:.4g}" for x in np.quantile(its, [0.01, 0.1, 0.5, 0.9, 0.99, 1.0])))
    print("  lane-steps quantiles:", " ".join(f"{x:.4g}" for x in np.quantile(steps, [0.01, 0.1, 0.5, 0.9, 0.99, 1.0])))
