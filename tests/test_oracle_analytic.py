"""The oracle against closed forms (CPU): the floor under a uniform sky of test_gpu_analytic.py.  Every sample of every
pixel returns albedo * sky whatever the random numbers were, so the frame is grade(albedo * sky) to rounding: a known
answer for the oracle's Lambert, miss, accumulation and tone-mapping chain that comes from the reference's formulas
(gpuRender.cu:24-116, :158-166), not from the oracle itself."""
import numpy as np
import pytest

import oracle_binding as ob
import ptamd
from test_gpu_analytic import (BoxScene, bumpy_mirror_floor, check_furnace, FURNACE, LOBE_CASES, LOBE_RES, LOBE_SPP, lobe_scene, MIRROR_RES, MODEL_RES, SKY_RES, SKY_VIEWS, TEX_RES, FloorScene, check_near_mirror, grade,
                               near_mirror_floor, placed_floor, sky_view, textured_floor)


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


@pytest.mark.parametrize("target,up,fov", SKY_VIEWS)
def test_sky_only_views_show_the_environment_map_oracle(target, up, fov):
    """Camera basis, pixel centres, image orientation and the orientation of the environment map against the
    reference's formulas restated in numpy (test_gpu_analytic.py: primary_directions, environment_texels)."""
    sc, expect, sky = sky_view(target, up, fov)
    img, _ = ob.render(sc, ptamd.make_params(*SKY_RES, spp=1, max_bounces=6), seed_mode=ptamd.SEED_REFERENCE_GPU, threads=2)
    bad = int(((np.abs(img - expect.astype(np.float32)).max(axis=2) >= 2e-6) & sky).sum())
    assert bad <= sky.sum() // 1000, bad
    # upside down or mirrored it is another picture: the check has teeth
    assert ((np.abs(img[::-1] - expect).max(axis=2) >= 2e-6) & sky).mean() > 0.2
    assert ((np.abs(img[:, ::-1] - expect).max(axis=2) >= 2e-6) & sky).mean() > 0.2


def test_textured_emissive_floor_oracle():
    """Texture coordinates, wrap-around, texel addressing and the emissive term against numpy (geometry.cpp:69-87,
    image.hpp:27-35, gpuRender.cu:60-66 and :107)."""
    sc, expect, sure = textured_floor()
    img, _ = ob.render(sc, ptamd.make_params(*TEX_RES, spp=1, max_bounces=6), seed_mode=ptamd.SEED_REFERENCE_GPU, threads=2)
    err = np.abs(img - expect.astype(np.float32)).max(axis=2)
    assert float(err[sure].max()) < 2e-6, (int((err[sure] >= 2e-6).sum()), int(sure.sum()))
    # the texture laid the other way round is another picture
    assert (np.abs(img - expect[:, ::-1]).max(axis=2)[sure] >= 2e-6).mean() > 0.3


def test_placed_floor_instance_transform_oracle():
    """Instance transform (inverse(model) for the ray, its transpose for the normal) against the silhouette and the
    colours computed in numpy."""
    sc, expect, sure, inside = placed_floor()
    img, st = ob.render(sc, ptamd.make_params(*MODEL_RES, spp=1, max_bounces=6), seed_mode=ptamd.SEED_REFERENCE_GPU, threads=2)
    err = np.abs(img - expect.astype(np.float32)).max(axis=2)
    assert float(err[sure].max()) < 2e-6, (int((err[sure] >= 2e-6).sum()), int(sure.sum()))
    assert abs(int(st.hits) - int(inside.sum())) <= int((~sure).sum()) and st.misses == MODEL_RES[0] * MODEL_RES[1]


def test_near_mirror_floor_reflection_fresnel_and_smith_oracle():
    """The GGX branch in its mirror limit: reflection direction, Schlick's Fresnel with F0 = albedo and Smith's G, as
    functions of the angle of incidence, against numpy."""
    sc, expect, floor = near_mirror_floor()
    img, _ = ob.render(sc, ptamd.make_params(*MIRROR_RES, spp=1, max_bounces=2), seed_mode=ptamd.SEED_REFERENCE_GPU, threads=2)
    check_near_mirror(img, expect, floor)
    # the check has teeth: the same frame with the environment mirrored left to right is far off
    assert np.median(np.abs(img[:, ::-1] - expect).max(axis=2)[floor & floor[:, ::-1]]) > 1e-2


@pytest.mark.parametrize("metalness,roughness", LOBE_CASES)
def test_sampled_mean_is_the_integral_of_the_lobe_oracle(metalness, roughness):
    """Cosine-weighted and GGX sampling, the tangent frame and the sample weights: the mean of 196,608 samples against
    the quadrature of the lobe times the environment (test_gpu_analytic.py: lobe_scene)."""
    sc, expect = lobe_scene(metalness, roughness)
    acc, _ = ob.accumulate(sc, ptamd.make_params(*LOBE_RES, spp=LOBE_SPP, max_bounces=2), seed_mode=ptamd.SEED_REFERENCE_GPU, threads=4)
    got = acc[..., :3].sum(axis=(0, 1)).astype(np.float64) / float(acc[..., 3].sum())
    assert np.abs(got / expect - 1.0).max() < 0.01, got / expect


def test_interpolated_normals_on_a_mirror_floor_oracle():
    """Barycentric order of the per-corner normals (geometry.cpp:69-87 with Moeller-Trumbore's u, v) through the
    reflection they cause."""
    sc, expect, sure = bumpy_mirror_floor()
    img, _ = ob.render(sc, ptamd.make_params(*MIRROR_RES, spp=1, max_bounces=2), seed_mode=ptamd.SEED_REFERENCE_GPU, threads=2)
    check_near_mirror(img, expect, sure)


def test_furnace_box_oracle():
    sc = BoxScene(*FURNACE)
    stats = {}

    def render(p):
        img, st = ob.render(sc, p, seed_mode=ptamd.SEED_REFERENCE_GPU, threads=2)
        stats["st"] = st
        return img

    check_furnace(render, lambda p: stats["st"])
