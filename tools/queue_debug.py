#!/usr/bin/env python3
"""Event counters of queueKernel (a -DPTX_QUEUE_DEBUG build: PTX_BACKEND_LIB=.../libptx_backend_qdbg.so).
usage: tools/queue_debug.py [spp] [streams] [scene]   (PTX_QUEUE_SHARDS / PTX_QUEUE_CHAINS from the environment)"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "path-tracer_amd", "py"))
import numpy as np
import ptamd

NAMES = ["services", "pop_tried", "pop_lost", "rays_taken", "batch_tried", "batch_lost", "batches", "batch_lanes",
         "ray_rereads", "shade_rereads", "idle_rounds", "iterations", "starved", "delivered"]
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 0
scene = int(sys.argv[3]) if len(sys.argv) > 3 else 1
os.environ["PTX_QUEUE"] = "1"
s = ptamd.HostScene(scene, cache_dir="/tmp/pt_scene_cache")
d = ptamd.DeviceScene(s.desc_ptr)
p = ptamd.make_params(1500, 800, spp=spp, max_bounces=6, seed_mode=1 if streams > 1 else 0, sub_streams=streams)
img, st = d.render(s.camera, p)
img, st = d.render(s.camera, p)
out = (C.c_uint * 16)()
lib = ptamd.backend_lib()
lib.ptx_debug_queue_counters.argtypes = [C.c_void_p, C.c_void_p]
rc = lib.ptx_debug_queue_counters(d._h, out)
print(json.dumps({"shards": os.environ.get("PTX_QUEUE_SHARDS"), "chains": os.environ.get("PTX_QUEUE_CHAINS"), "ms": round(st.render_s * 1e3, 2), "rc": rc,
                  **{n: out[i] for i, n in enumerate(NAMES)}}))
