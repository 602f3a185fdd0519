"""The oracle against closed forms (CPU): the floor under a uniform sky of test_gpu_analytic.py.  Every sample of every
pixel returns albedo * sky whatever the random numbers were, so the frame is grade(albedo * sky) to rounding: a known
answer for the oracle's Lambert, miss, accumulation and tone-mapping chain that comes from the reference's formulas
(gpuRender.cu:24-116, :158-166), not from the oracle itself."""
import numpy as np
import pytest

import oracle_binding as ob
import ptamd
from test_gpu_analytic import FloorScene, grade


@pytest.mark.parametrize("albedo,sky", [((0.5, 0.25, 0.75), (0.8, 0.8, 0.8)), ((1.0, 1.0, 1.0), (2.0, 0.5, 0.1))])
def test_floor_under_a_uniform_sky_oracle(albedo, sky):
    sc = FloorScene(albedo, sky)
    a, L = np.array(albedo), np.array(sky)
    for seed_mode in (ptamd.SEED_REFERENCE_GPU, 100):  # the GPU kernel's seeding and the CPU path's (-d 1)
        for spp, bounces, expect in ((1, 6, grade(a * L)), (5, 2, grade(a * L)), (3, 1, grade(a))):
            img, st = ob.render(sc, ptamd.make_params(64, 40, spp=spp, max_bounces=bounces), seed_mode=seed_mode, threads=2)
            assert np.abs(img - expect).max() < 2e-6
            assert st.hits == 64 * 40 * spp and st.misses == (0 if bounces == 1 else 64 * 40 * spp)


def test_mirror_floor_energy_oracle():
    for roughness, lo, hi in ((0.05, 0.97, 1.0), (0.5, 0.6, 0.7)):
        sc = FloorScene((1.0, 1.0, 1.0), (1.0, 1.0, 1.0), metalness=1.0, roughness=roughness)
        acc, _ = ob.accumulate(sc, ptamd.make_params(64, 48, spp=32, max_bounces=2), seed_mode=ptamd.SEED_REFERENCE_GPU, threads=2)
        mean = acc[..., :3].sum() / (3.0 * acc[..., 3].sum())
        assert lo < mean <= hi + 1e-3
