"""The C ABI takes any camera, the presets fix one: parity away from it, and on the rays the presets never shoot.

* other view points (inside the model, far away, narrow and wide fields of view, a tilted `up`): `-a` counts
  bit-exact against the oracle, radiance with identical seeds within the tolerance of test_gpu_radiance.py;
* axis-aligned views at even resolutions: the centre column / row of primary rays has direction components
  that are EXACTLY zero, so the slab test meets 0 * inf = NaN and GLM's min / max semantics decide
  (intersection.hpp:48-75; trace.hpp EXACT_NAN / wideStep<SAFE>): the wave-uniform exact path of both layouts
  is what runs here, and the counts still have to be bit-exact."""
import types

import numpy as np
import pytest

import oracle_binding as ob
import ptamd

pytestmark = pytest.mark.gpu


def camera(position, target, up=(0.0, 1.0, 0.0), fov=45.0):
    c = ptamd.Camera()
    c.position[:] = position
    c.target[:] = target
    n = float(np.linalg.norm(up))
    c.up[:] = [u / n for u in up]
    c.fov_deg = fov
    return c


def with_camera(scene, cam):
    """What oracle_binding reads of a HostScene, with another camera."""
    return types.SimpleNamespace(desc_ptr=scene.desc_ptr, camera=cam)


CAMERAS = {
    "behind": ((-0.6, 0.35, 0.9), (0.0, 0.1, 0.0), (0, 1, 0), 45.0),
    "inside": ((0.02, 0.05, 0.03), (0.3, 0.1, -0.2), (0, 1, 0), 70.0),
    "far_narrow": ((4.0, 2.5, -6.0), (0.0, 0.0, 0.0), (0, 1, 0), 8.0),
    "wide_tilted": ((0.4, 0.9, -0.5), (0.0, 0.0, 0.0), (0.3, 1.0, 0.2), 120.0),
    "from_below": ((0.1, -1.2, 0.1), (0.0, 0.0, 0.0), (0, 0, 1), 45.0),
}


@pytest.mark.parametrize("preset", [2, 0, 1])
@pytest.mark.parametrize("name", sorted(CAMERAS))
def test_counts_bit_exact_from_other_view_points(gpu_scenes, preset, name):
    hosts, devs = gpu_scenes
    cam = camera(*CAMERAS[name])
    p = ptamd.make_params(320, 200)
    ref, st_ref = ob.render_counts(with_camera(hosts[preset], cam), p)
    got, _ = devs[preset].render_counts(cam, p)
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} of {ref.size} pixels differ"
    st = devs[preset].measure(cam, p, heatmap=True)
    assert (st.node_steps, st.tri_tests, st.inst_entries, st.box_hits) == \
           (st_ref.node_steps, st_ref.tri_tests, st_ref.inst_entries, st_ref.box_hits)


@pytest.mark.parametrize("preset,axis", [(2, "z"), (2, "x"), (0, "z"), (1, "z"), (1, "x")])
def test_axis_aligned_views_with_exactly_zero_direction_components(gpu_scenes, preset, axis):
    hosts, devs = gpu_scenes
    pos = (0.0, 0.0, -2.5) if axis == "z" else (2.5, 0.0, 0.0)
    cam = camera(pos, (0.0, 0.0, 0.0))
    p = ptamd.make_params(128, 96)  # even: pixel (64, y) has d.x == 0 in view space, pixel (x, 48) d.y == 0
    ref, st_ref = ob.render_counts(with_camera(hosts[preset], cam), p)
    got, _ = devs[preset].render_counts(cam, p)
    assert ref.max() > 0, "the view sees nothing: the test would be empty"
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} of {ref.size} pixels differ"
    # radiance, one sample: no jitter (gpuRender.cu:134), so the primary rays keep their exact zeros and the wide
    # layout's SAFE step runs; identical seeds -> the same image within the same-seed tolerance
    pr = ptamd.make_params(128, 96, spp=1, max_bounces=4)
    rref, _ = ob.render(with_camera(hosts[preset], cam), pr, seed_mode=0)
    rgot, _ = devs[preset].render(cam, pr)
    diff = np.abs(rgot - rref).max(axis=2)
    # the centre column and row are the rays in question: every one of them must agree (a bounced path can
    # decorrelate by a 1-ulp libm difference on scene 1, hence the frame-wide allowance only there)
    line = np.concatenate([diff[:, 64], diff[48, :]])
    allowed = 0 if preset != 1 else 4
    assert int((line > 2e-3).sum()) <= allowed, f"{int((line > 2e-3).sum())} centre-line pixels differ"
    assert float((diff <= 2e-3).mean()) >= (0.99 if preset != 1 else 0.93)


@pytest.mark.parametrize("preset", [2, 0])
def test_radiance_with_identical_seeds_from_other_view_points(gpu_scenes, preset):
    hosts, devs = gpu_scenes
    for name in ("behind", "inside", "wide_tilted"):
        cam = camera(*CAMERAS[name])
        p = ptamd.make_params(160, 100, spp=4, max_bounces=6)
        ref, _ = ob.render(with_camera(hosts[preset], cam), p, seed_mode=0)
        got, _ = devs[preset].render(cam, p)
        diff = np.abs(got - ref)
        rmse = float(np.sqrt(np.mean(diff ** 2)))
        frac = float((diff.max(axis=2) <= 2e-3).mean())
        bias = float((got.astype(np.float64) - ref).mean())
        print(f"scene {preset} {name}: rmse {rmse:.2e}, within 2e-3: {frac:.4f}, mean diff {bias:+.2e}")
        # scene 2 (diffuse, textured) agrees to rounding.  On scene 0 the metallic busts fill these views: a 1-ulp
        # difference between the device's and glibc's sinf / cosf / powf flips a discrete decision on a few per cent
        # of the specular chains, and such a pixel then holds another, equally valid, sample:
        # most pixels still agree to rounding, the rest differ without bias
        tol_rmse, tol_frac = (5e-3, 0.99) if preset == 2 else (4e-2, 0.95)
        assert np.isfinite(got).all() and rmse < tol_rmse and frac >= tol_frac and abs(bias) < 2e-3, (name, rmse, frac, bias)
