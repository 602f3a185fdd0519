#!/usr/bin/env python3
"""Kernel time of small frames (the fixed cost of a launch): tools/small_frames.py [scene]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "path-tracer_amd", "py"))
import ptamd
scene = int(sys.argv[1]) if len(sys.argv) > 1 else 1
s = ptamd.HostScene(scene, cache_dir="/tmp/pt_scene_cache"); d = ptamd.DeviceScene(s.desc_ptr)
for q in ("1", "0"):
    os.environ["PTX_QUEUE"] = q
    for (w, h, spp, streams) in [(8, 8, 1, 0), (64, 64, 1, 0), (256, 256, 1, 0), (1500, 800, 1, 0), (1500, 800, 2, 2), (1500, 800, 4, 4), (1500, 800, 8, 8)]:
        p = ptamd.make_params(w, h, spp=spp, max_bounces=6, seed_mode=1 if streams > 1 else 0, sub_streams=streams)
        ts = []
        for i in range(4):
            img, st = d.render(s.camera, p); ts.append(st.render_s * 1e3)
        print(f"{'pool' if q == '1' else 'path'} kernel  {w}x{h} {spp} spp {streams} streams: {min(ts):.3f} ms (first {ts[0]:.3f})", flush=True)
