#!/usr/bin/env python3
"""A/B timing of backend builds on the bench workload (run on the GPU box).
usage: tools/ab.py [--spp N] [--rounds R] name1 name2 ...   (name -> path-tracer_amd/lib/libptx_backend_<name>.so,
"base" -> libptx_backend.so).  Each variant runs in its own process; rounds are interleaved."""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os, json, time
sys.path.insert(0, os.path.join(%(repo)r, "path-tracer_amd", "py")); sys.path.insert(0, os.path.join(%(repo)r, "tests"))
import numpy as np, ptamd
scene_id, spp, check = %(scene)d, %(spp)d, %(check)d
s = ptamd.HostScene(scene_id); d = ptamd.DeviceScene(s.desc_ptr)
ss = int(os.environ.get("AB_STREAMS", "0"))
p = ptamd.make_params(1500, 800, spp=spp, max_bounces=6, seed_mode=1 if ss > 1 else 0, sub_streams=ss)
ts = []
for i in range(%(iters)d):
    img, st = d.render(s.camera, p); ts.append(st.render_s)
pc = ptamd.make_params(1500, 800)
c, stc = d.render_counts(s.camera, pc)
out = {"render_ms": [round(t*1e3, 2) for t in ts], "counts_ms": round(stc.render_s*1e3, 3), "counts_sum": int(c.sum()), "counts_max": int(c.max()), "img_mean": float(img.mean())}
import zlib
out["img_crc"] = zlib.crc32(np.ascontiguousarray(img).tobytes())
if %(measure)d:
    m = d.measure(s.camera, p)
    out.update(rays=m.rays, S=m.node_steps, T=m.tri_tests, I=m.inst_entries)
    np.save(os.path.join(%(repo)r, "gpurun_out", "ab_img_" + os.path.basename(os.environ["PTX_BACKEND_LIB"]) + ".npy"), img)
if check:
    import oracle_binding as ob
    ref, _ = ob.render_counts(s, pc)
    out["counts_exact"] = bool(np.array_equal(ref, c))
print(json.dumps(out))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--scene", type=int, default=1)
    ap.add_argument("--check", type=int, default=1)
    ap.add_argument("--measure", type=int, default=0, help="instrumented launch (step counts) and the image saved under gpurun_out/")
    ap.add_argument("names", nargs="+")
    a = ap.parse_args()
    res = {n: [] for n in a.names}
    for r in range(a.rounds):
        for n in a.names:
            parts = n.split(":")  # name[:ENV=VALUE...]
            lib = "libptx_backend.so" if parts[0] == "base" else f"libptx_backend_{parts[0]}.so"
            env = dict(os.environ, PTX_BACKEND_LIB=os.path.join(REPO, "path-tracer_amd", "lib", lib))
            for kv in parts[1:]:
                k, v = kv.split("=", 1)
                env[k] = v
            code = CHILD % {"repo": REPO, "scene": a.scene, "spp": a.spp, "check": a.check if r == 0 else 0, "iters": a.iters, "measure": a.measure if r == 0 else 0}
            out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
            if out.returncode != 0:
                print(n, "FAILED", out.stderr[-800:], flush=True)
                continue
            j = json.loads(out.stdout.strip().splitlines()[-1])
            res[n].append(j)
            print(r, n, json.dumps(j), flush=True)
    print("\nsummary (min render ms over all rounds):")
    for n in a.names:
        if res[n]:
            print(f"  {n:28s} {min(min(j['render_ms']) for j in res[n]):9.2f} ms   counts {min(j['counts_ms'] for j in res[n]):7.3f} ms  exact={res[n][0].get('counts_exact')}")


if __name__ == "__main__":
    main()
