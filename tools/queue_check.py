#!/usr/bin/env python3
"""queueKernel against pathKernel on the GPU box: same seeds, images must be bit-identical.
usage: tools/queue_check.py [--big]   (each case in a child process with its own time limit; stops at the first
case that crashes or hangs)"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json, time, zlib
sys.path.insert(0, os.path.join(%(repo)r, "path-tracer_amd", "py"))
import numpy as np, ptamd
scene, w, h, spp, bounces, streams, iters = %(scene)d, %(w)d, %(h)d, %(spp)d, %(bounces)d, %(streams)d, %(iters)d
s = ptamd.HostScene(scene, cache_dir="/tmp/pt_scene_cache"); d = ptamd.DeviceScene(s.desc_ptr)
p = ptamd.make_params(w, h, spp=spp, max_bounces=bounces, seed_mode=1 if streams > 1 else 0, sub_streams=streams)
out = {}
imgs = {}
for mode in ("0", "1"):
    os.environ["PTX_QUEUE"] = mode
    ts = []
    for i in range(iters):
        img, st = d.render(s.camera, p); ts.append(round(st.render_s * 1e3, 3))
    imgs[mode] = img
    out["ms_q" + mode] = ts
    if %(measure)d:
        m = d.measure(s.camera, p)
        out["stats_q" + mode] = [m.rays, m.node_steps, m.tri_tests, m.inst_entries, m.hits, m.misses]
diff = imgs["0"] != imgs["1"]
out["pixels_differ"] = int(diff.any(axis=2).sum())
out["max_abs"] = float(np.abs(imgs["0"] - imgs["1"]).max())
out["crc"] = [zlib.crc32(np.ascontiguousarray(imgs[m]).tobytes()) for m in ("0", "1")]
print(json.dumps(out))
'''

CASES = [  # scene, w, h, spp, bounces, streams, iters, seconds
    (2, 64, 48, 1, 1, 0, 1, 90),
    (2, 160, 96, 4, 4, 0, 1, 60),
    (2, 160, 96, 9, 6, 4, 1, 60),
    (0, 240, 128, 8, 6, 0, 1, 90),
    (1, 240, 128, 8, 6, 3, 2, 90),
    (1, 333, 77, 5, 3, 0, 1, 60),
    (1, 1500, 800, 10, 6, 0, 2, 120),
    (1, 1500, 800, 10, 6, 4, 2, 120),
]
BIG = [
    (1, 1500, 800, 100, 6, 34, 2, 200),
    (1, 1500, 800, 100, 6, 0, 2, 200),
]


def main():
    cases = CASES + (BIG if "--big" in sys.argv else [])
    if "--only-big" in sys.argv:
        cases = BIG
    measure = 1 if "--measure" in sys.argv else 0
    bad = 0
    for c in cases:
        code = CHILD % dict(repo=REPO, scene=c[0], w=c[1], h=c[2], spp=c[3], bounces=c[4], streams=c[5], iters=c[6], measure=measure)
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=c[7])
        except subprocess.TimeoutExpired:
            print("CASE", c, "TIMED OUT -- stopping", flush=True)
            sys.exit(124)
        if r.returncode != 0:
            print("CASE", c, "rc", r.returncode, r.stderr[-1500:], flush=True)
            if r.returncode < 0 or r.returncode in (134, 139):
                sys.exit(r.returncode if r.returncode > 0 else 128 - r.returncode)
            bad += 1
            continue
        j = json.loads(r.stdout.strip().splitlines()[-1])
        bad += j["pixels_differ"] != 0
        print("CASE", c, json.dumps(j), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
