"""The radiance path's TLAS carries tight boxes (csrc/tight_bounds.hpp): bounds of every instance's transformed vertices,
computed on the GPU in ptx_create, the reference's tree refitted bottom-up.  Checked here: the device's bounds are the
host loop's bit for bit, no tight box leaves the reference's box, every leaf holds its instances; and the images are
the ones the reference's boxes give (same hits): PTX_TIGHT_TLAS=0 in a child process renders the comparison frame."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import ptamd

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("preset", [0, 1, 2])
def test_tight_boxes_are_exact_and_inside_the_reference_s(gpu_scenes, preset):
    hosts, _ = gpu_scenes
    lib = ptamd.backend_lib()
    lib.ptx_check_tight_tlas.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    rep = (C.c_uint64 * 6)()
    assert lib.ptx_check_tight_tlas(hosts[preset].desc_ptr, 0, rep) == 0, lib.ptx_last_error()
    n, checked, differ, outside, leaf_misses, ratio = [int(v) for v in rep]
    print(f"scene {preset}: {n} instances, {checked} recomputed on the host, leaf volume tight / reference {ratio / 1e6:.3f}")
    assert n == hosts[preset].desc.instance_count and checked >= min(n, 64)
    assert differ == 0 and outside == 0 and leaf_misses == 0
    assert ratio <= 1_000_000
    if preset == 1:
        assert ratio < 600_000  # rotated busts: the reference's boxes have about twice the volume


def test_box_corners_stand_in_for_the_vertices_of_very_large_scenes(tmp_path):
    """Beyond 4e10 triangle-instance pairs ptx_create bounds an instance by the corners of its BLAS boxes eight levels
    down instead of its vertices (forced here with PTX_TIGHT_EXACT_BUDGET=0): still conservative, still the same image."""
    code = r"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, %(py)r)
import ptamd
s = ptamd.HostScene(1, scatter_instances=400)
lib = ptamd.backend_lib()
lib.ptx_check_tight_tlas.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
rep = (C.c_uint64 * 6)()
assert lib.ptx_check_tight_tlas(s.desc_ptr, 0, rep) == 0, lib.ptx_last_error()
d = ptamd.DeviceScene(s.desc_ptr)
img, _ = d.render(s.camera, ptamd.make_params(160, 96, spp=4, max_bounces=6))
np.save(%(out)r, img)
print(*[int(v) for v in rep])
"""
    res = {}
    for name, env in (("standin", {"PTX_TIGHT_EXACT_BUDGET": "0"}), ("exact", {})):
        out = str(tmp_path / f"{name}.npy")
        r = subprocess.run([sys.executable, "-c", code % {"py": os.path.join(os.path.dirname(HERE), "path-tracer_amd", "py"), "out": out}],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stderr[-1500:]
        res[name] = (np.load(out), [int(v) for v in r.stdout.split()[-6:]])
    for name in res:
        n, checked, differ, outside, leaf_misses, ratio = res[name][1]
        assert n == 400 and checked >= 64 and differ == 0 and outside == 0 and leaf_misses == 0, (name, res[name][1])
    assert res["exact"][1][5] < res["standin"][1][5] < 1_000_000     # looser than the vertices, tighter than the reference
    assert float((res["exact"][0] == res["standin"][0]).all(axis=2).mean()) >= 0.999


CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(py)r)
import ptamd
s = ptamd.HostScene(%(preset)d)
d = ptamd.DeviceScene(s.desc_ptr)
p = ptamd.make_params(240, 128, spp=8, max_bounces=6, seed_mode=ptamd.SEED_STREAM, sub_streams=4)
img, _ = d.render(s.camera, p)
m = d.measure(s.camera, p)
np.save(%(out)r, img)
print(m.rays, m.node_steps, m.tri_tests, m.inst_entries)
"""


@pytest.mark.parametrize("preset", [0, 1])
def test_images_do_not_depend_on_the_boxes(tmp_path, preset):
    res = {}
    for name, env in (("tight", {}), ("reference", {"PTX_TIGHT_TLAS": "0"})):
        out = str(tmp_path / f"{name}.npy")
        code = CHILD % {"py": os.path.join(os.path.dirname(HERE), "path-tracer_amd", "py"), "preset": preset, "out": out}
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stderr[-1500:]
        res[name] = (np.load(out), [int(v) for v in r.stdout.split()[-4:]])
    (a, sa), (b, sb) = res["tight"], res["reference"]
    same = float((a == b).all(axis=2).mean())
    print(f"scene {preset}: {100 * same:.3f} % of pixels bit-identical; instance entries {sa[3]} vs {sb[3]}, node steps {sa[1]} vs {sb[1]}")
    # the same hits: only exact ties in t between two instances, reached in another order, may resolve differently
    assert same >= 0.999 and float(np.abs(a - b).mean()) < 1e-4
    assert abs(sa[0] - sb[0]) <= max(4, sb[0] // 10000)          # the same rays
    assert sa[3] <= sb[3] and sa[1] <= sb[1]                      # fewer entries, fewer steps
    if preset == 1:
        assert sa[3] < 0.8 * sb[3]
