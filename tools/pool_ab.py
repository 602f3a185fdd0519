#!/usr/bin/env python3
"""Timing of the pool kernel against pathKernel on the bench frame (GPU box).
usage: tools/pool_ab.py spp streams variant[:ENV=V...] ...   (variant "base" = libptx_backend.so; PTX_QUEUE is set per run:
a variant name starting with "path" runs pathKernel)"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json, zlib
if os.environ.get("AB_TORCH_FIRST") == "1":
    import torch
sys.path.insert(0, os.path.join(%(repo)r, "path-tracer_amd", "py"))
import numpy as np, ptamd
s = ptamd.HostScene(%(scene)d, cache_dir="/tmp/pt_scene_cache"); d = ptamd.DeviceScene(s.desc_ptr)
streams = %(streams)d
p = ptamd.make_params(1500, 800, spp=%(spp)d, max_bounces=6, seed_mode=1 if streams > 1 else 0, sub_streams=streams)
ts = []
for i in range(3):
    img, st = d.render(s.camera, p); ts.append(round(st.render_s * 1e3, 2))
print(json.dumps({"ms": ts, "crc": zlib.crc32(np.ascontiguousarray(img).tobytes())}))
'''
spp, streams = int(sys.argv[1]), int(sys.argv[2])
scene = int(os.environ.get("AB_SCENE", "1"))
for v in sys.argv[3:]:
    parts = v.split(":")
    name = parts[0]
    lib = name[4:] if name.startswith("path") else name
    lib = "libptx_backend.so" if lib in ("", "base") else f"libptx_backend_{lib}.so"
    env = dict(os.environ, PTX_BACKEND_LIB=os.path.join(REPO, "path-tracer_amd", "lib", lib), PTX_QUEUE="0" if name.startswith("path") else "1")
    for kv in parts[1:]:
        k, val = kv.split("=", 1)
        env[k] = val
    try:
        r = subprocess.run([sys.executable, "-c", CHILD % dict(repo=REPO, spp=spp, streams=streams, scene=scene)], env=env, capture_output=True, text=True, timeout=180)
    except subprocess.TimeoutExpired:
        print(v, "TIMED OUT -- stopping", flush=True)
        sys.exit(124)
    if r.returncode != 0:
        print(v, "rc", r.returncode, r.stderr[-600:], flush=True)
        if r.returncode < 0 or r.returncode in (134, 139):
            sys.exit(1)
        continue
    print(f"{spp:4d} spp {streams:3d} streams  {v:40s}", r.stdout.strip().splitlines()[-1], flush=True)
