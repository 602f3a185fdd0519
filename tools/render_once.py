#!/usr/bin/env python3
"""One frame of the bench workload through ptx_render (for rocprofv3 passes): tools/render_once.py spp streams [scene] [frames]
PTX_QUEUE / PTX_POOL_CHAINS / PTX_BACKEND_LIB from the environment."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "path-tracer_amd", "py"))
import ptamd

spp, streams = int(sys.argv[1]), int(sys.argv[2])
scene = int(sys.argv[3]) if len(sys.argv) > 3 else 1
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 1
s = ptamd.HostScene(scene, cache_dir="/tmp/pt_scene_cache")
d = ptamd.DeviceScene(s.desc_ptr)
p = ptamd.make_params(1500, 800, spp=spp, max_bounces=6, seed_mode=1 if streams > 1 else 0, sub_streams=streams)
for i in range(frames):
    img, st = d.render(s.camera, p)
    print("render_ms", round(st.render_s * 1e3, 2), flush=True)
