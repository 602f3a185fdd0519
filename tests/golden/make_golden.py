#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle + host pipeline.

Run here (CPU only) AFTER tests/test_pins.py passes, i.e. after the oracle and the builder reproduce
the reference's published scene-2 numbers (README.md:60).  The reference itself cannot be executed
(no nvcc, no GLM, assets missing), so these vectors are outputs of the pinned restatement, not of the
reference binary; they freeze its behaviour so that later edits (and the HIP path) are checked
against the same numbers on any machine.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402
import ptamd  # noqa: E402
import ctypes as C  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    out = {}
    # (iv) first 16 PCG outputs from the three seeds the reference uses (gpuRender.cu:130,
    # main.cpp:186, sampleScenes.cpp:85)
    pcg = {}
    for seed in (1023, 1031, 7142):
        st = C.c_uint32(seed)
        pcg[str(seed)] = [int(ob.lib().oracle_pcg_hash(C.byref(st))) for _ in range(16)]
    out["pcg_hash"] = pcg

    s = ptamd.HostScene(2)
    # (i) scene 2 BVH
    out["scene2_bvh"] = {"triangles": int(s.info.triangles_per_mesh), "blas_nodes": int(s.info.blas_nodes),
                         "tlas_nodes": int(s.info.tlas_nodes), "blas_depth": int(s.info.blas_depth),
                         "blas_sha256": sha(s.blas_nodes), "triangles_sha256": sha(s.triangles),
                         "instance_sha256": sha(s.instances)}
    # (ii) scene 2 heat maps
    c256, _ = ob.render_counts(s, ptamd.make_params(256, 256))
    assert c256.max() < 65536
    np.savez_compressed(os.path.join(HERE, "scene2_counts_256.npz"), counts=c256.astype(np.uint16))
    cfull, _ = ob.render_counts(s, ptamd.make_params(1500, 800))
    out["scene2_counts_1500x800"] = {"max": int(cfull.max()), "sum": int(cfull.sum()), "sha256": sha(cfull)}
    c_odd, _ = ob.render_counts(s, ptamd.make_params(37, 29))
    out["scene2_counts_37x29"] = {"max": int(c_odd.max()), "sum": int(c_odd.sum()), "sha256": sha(c_odd)}
    # (vi) a 64x64 radiance image, reference GPU seeding (deterministic for any thread count)
    p = ptamd.make_params(64, 64, spp=256, max_bounces=4)
    img, st = ob.render(s, p, seed_mode=ptamd.SEED_REFERENCE_GPU)
    np.save(os.path.join(HERE, "scene2_radiance_64_256spp.npy"), img.astype(np.float32))
    out["scene2_radiance_64_256spp"] = {"rays": int(st.rays), "mean": float(img.mean())}
    # the reference CPU seeding with one thread (`-d 1 -t 1`), BASELINE config 0 shape at reduced size
    p = ptamd.make_params(32, 32, spp=16, max_bounces=4)
    img, st = ob.render(s, p, seed_mode=ob.SEED_REFERENCE_CPU, threads=1)
    np.save(os.path.join(HERE, "scene2_radiance_32_16spp_cpuseed.npy"), img.astype(np.float32))
    s.close()

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1)[:600])


if __name__ == "__main__":
    main()
