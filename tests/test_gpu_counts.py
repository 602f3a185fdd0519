"""`-a` heat map parity, HIP (through the C ABI) vs the CPU oracle: BIT-EXACT integers
(BASELINE.json configs[2])."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
import ptamd

pytestmark = pytest.mark.gpu

with open(os.path.join(ptamd.GOLDEN_DIR, "golden.json")) as f:
    G = json.load(f)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("preset,size", [(2, (1500, 800)), (0, (1500, 800)), (1, (1500, 800)),
                                         (2, (37, 29)), (2, (1, 1)), (2, (8, 8)), (2, (333, 1)), (2, (1, 77)),
                                         (0, (65, 63)), (1, (129, 31))])
def test_counts_bit_exact(gpu_scenes, preset, size):
    hosts, devs = gpu_scenes
    p = ptamd.make_params(*size)
    ref, st_ref = ob.render_counts(hosts[preset], p)
    got, _ = devs[preset].render_counts(hosts[preset].camera, p)
    assert got.dtype == np.uint32 and got.shape == ref.shape
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} of {ref.size} pixels differ"
    st = devs[preset].measure(hosts[preset].camera, p, heatmap=True)
    # the work counters agree exactly as well (S, T, I)
    assert (st.rays, st.node_steps, st.tri_tests, st.inst_entries, st.box_hits) == \
           (st_ref.rays, st_ref.node_steps, st_ref.tri_tests, st_ref.inst_entries, st_ref.box_hits)


def test_counts_against_committed_golden(gpu_scenes):
    hosts, devs = gpu_scenes
    gold = np.load(os.path.join(ptamd.GOLDEN_DIR, "scene2_counts_256.npz"))["counts"].astype(np.uint32)
    got, _ = devs[2].render_counts(hosts[2].camera, ptamd.make_params(256, 256))
    assert np.array_equal(got, gold)
    got, _ = devs[2].render_counts(hosts[2].camera, ptamd.make_params(1500, 800))
    g = G["scene2_counts_1500x800"]
    assert (int(got.max()), int(got.sum()), sha(got)) == (g["max"], g["sum"], g["sha256"])
    assert int(got.max()) == 123  # README.md:60


def test_counts_are_repeatable_and_size_independent_properties(gpu_scenes):
    hosts, devs = gpu_scenes
    p = ptamd.make_params(1500, 800)
    a, _ = devs[1].render_counts(hosts[1].camera, p)
    b, _ = devs[1].render_counts(hosts[1].camera, p)
    assert np.array_equal(a, b)                      # idempotent
    # the frame has cheap and expensive pixels
    assert a.min() < a.max() and a.max() > 100


def test_tile_partition_reassembles_exactly(gpu_scenes):
    hosts, devs = gpu_scenes
    full, _ = devs[1].render_counts(hosts[1].camera, ptamd.make_params(301, 203))
    total = np.zeros_like(full)
    for parts in (2, 3, 8):
        total[:] = 0
        for r in range(parts):
            c, _ = devs[1].render_counts(hosts[1].camera, ptamd.make_params(301, 203, part_index=r, part_count=parts))
            assert ((c == 0) | (total == 0)).all()   # disjoint support
            total += c
        assert np.array_equal(total, full)


def test_bad_params_are_rejected(gpu_scenes):
    hosts, devs = gpu_scenes
    with pytest.raises(ptamd.BackendError):
        devs[2].render_counts(hosts[2].camera, ptamd.make_params(0, 10))
    with pytest.raises(ptamd.BackendError):
        devs[2].render_counts(hosts[2].camera, ptamd.make_params(8, 8, part_index=3, part_count=2))
    with pytest.raises(ptamd.BackendError):
        devs[2].render(hosts[2].camera, ptamd.make_params(8, 8, spp=0))
    with pytest.raises(ptamd.BackendError):
        devs[2].render_counts(hosts[2].camera, ptamd.make_params(5000, 5000))  # > 2^24 pixels
