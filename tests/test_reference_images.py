"""Pins against the two pictures the reference holds of its own scene-2 output (README.md:50-51, reduced to the
fixtures of tests/golden/reference_images/, see make_fixtures.py there).

* `images/obj_bvh.png` is a screen shot of the reference's `-a` heat map.  The heat map depends on nothing that this
  checkout lacks, so the picture must be OUR picture: the oracle's counts (and, on the GPU, the HIP kernel's, which
  are bit-identical to them), coloured by the host stage, scaled and cropped the way the screen shot was, correlate
  with it to 0.99+.
* `images/obj.png` is the reference's render, at the benchmark resolution.  Its light comes from an HDR that the
  checkout lacks (the stand-in sky is darker and bluer), so absolute values cannot agree; what is compared is the
  local structure that does not depend on the light -- texture detail, edges, silhouettes: log-luminance minus its
  local mean -- which pins camera, projection, geometry, the instance matrix, texture coordinates and the image
  orientation to the pixel: the correlation peaks at zero shift and drops by a third when the picture is moved by
  one sample (4 pixels) in any direction."""
import os

import numpy as np
import pytest
from scipy import ndimage as ndi

import oracle_binding as ob
import ptamd

PIL_Image = pytest.importorskip("PIL.Image")
FIX = os.path.join(ptamd.GOLDEN_DIR, "reference_images")
W, H = 1500, 800


def fixture(name):
    return np.asarray(PIL_Image.open(os.path.join(FIX, name))).astype(np.float64) / 255.0


def box4(lum):
    h, w = lum.shape[0] // 4 * 4, lum.shape[1] // 4 * 4
    return lum[:h, :w].reshape(h // 4, 4, w // 4, 4).mean(axis=(1, 3))


def corr(a, b):
    return float(np.corrcoef(a.ravel(), b.ravel())[0, 1])


def heat_map_correlation(counts):
    """counts: (H, W) uint32, row 0 = bottom scan line (as every entry point returns them)."""
    rgb, _ = ptamd.heatmap_colour(counts)                       # main.cpp:363-374 (afmhot of count / max)
    lum = np.clip(np.asarray(rgb).reshape(H, W, 3), 0.0, 1.0).mean(axis=2)[::-1]
    # the viewer scaled the 1500x800 picture by 1.829 and the shot cut 12 / 8 pixels off its left / top (found by a
    # search over scale and offset when the fixture was made: correlation 0.9966 there, < 0.98 two pixels away)
    big = np.asarray(PIL_Image.fromarray((lum * 255.0).astype(np.uint8)).resize((2744, 1463), PIL_Image.BILINEAR)).astype(np.float64) / 255.0
    shot = fixture("obj_bvh_lum_681x360.png")
    ours = box4(big[8:8 + 1441, 12:12 + 2724])
    return corr(ours[:shot.shape[0]], shot)


def structure(lum4):
    g = np.log(lum4 + 0.02)
    return g - ndi.gaussian_filter(g, 2.0)


def picture_correlations(image):
    """image: (H, W, 3) float, row 0 = bottom.  Returns (correlation at zero shift, best correlation one sample away)."""
    ours = structure(box4(image[::-1].mean(axis=2)))
    ref = structure(fixture("obj_lum_375x200.png"))
    m = (slice(8, -8), slice(8, -8))
    at0 = corr(ours[m], ref[m])
    moved = max(corr(ours[m], np.roll(np.roll(ref, dy, 0), dx, 1)[m])
                for dy, dx in ((0, 1), (1, 0), (0, -1), (-1, 0), (1, 1), (-1, -1), (1, -1), (-1, 1)))
    return at0, moved


def test_heat_map_is_the_reference_s_screen_shot(scene2):
    counts, _ = ob.render_counts(scene2, ptamd.make_params(W, H))
    assert int(counts.max()) == 123                              # README.md:60
    c = heat_map_correlation(counts)
    print(f"heat map vs images/obj_bvh.png: correlation {c:.4f}")
    assert c > 0.99


def test_render_has_the_structure_of_the_reference_s_picture(scene2):
    img, _ = ob.render(scene2, ptamd.make_params(W, H, spp=8, max_bounces=6), seed_mode=ptamd.SEED_REFERENCE_GPU)
    at0, moved = picture_correlations(img)
    print(f"render (8 spp) vs images/obj.png: structure correlation {at0:.3f}, best one sample away {moved:.3f}")
    assert at0 > 0.75 and at0 - moved > 0.2
    # upside down or mirrored it is another picture altogether
    assert picture_correlations(img[::-1])[0] < 0.2 and picture_correlations(img[:, ::-1])[0] < 0.2


@pytest.mark.gpu
def test_gpu_heat_map_and_render_against_the_reference_s_pictures(gpu_scenes):
    hosts, devs = gpu_scenes
    counts, _ = devs[2].render_counts(hosts[2].camera, ptamd.make_params(W, H))
    assert heat_map_correlation(counts) > 0.99
    # 128 samples: the noise is gone, what is left of 1 - correlation is the light (HDR absent)
    for seed_mode, streams in ((ptamd.SEED_REFERENCE_GPU, 0), (ptamd.SEED_STREAM, 16)):
        img, _ = devs[2].render(hosts[2].camera, ptamd.make_params(W, H, spp=128, max_bounces=6, seed_mode=seed_mode, sub_streams=streams))
        at0, moved = picture_correlations(img)
        print(f"HIP render (128 spp, streams {streams}) vs images/obj.png: structure correlation {at0:.3f}, one sample away {moved:.3f}")
        assert at0 > 0.87 and at0 - moved > 0.25
