"""Device-side output stage: ptx_render_counts_bgr8 / ptx_render_bgr8 return the BMP
pixel array; it has to be bit-identical to what the host stage (main.cpp:363-374 heat-map colouring,
output.cpp:13-23 clamp / 8-bit quantise / flip) makes of the same frame."""
import os

import numpy as np
import pytest

import ptamd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("preset,size", [(2, (1500, 800)), (1, (750, 400)), (0, (333, 77)), (2, (37, 29)), (2, (1, 1))])
def test_heat_map_bytes_equal_the_host_stage(gpu_scenes, preset, size):
    hosts, devs = gpu_scenes
    p = ptamd.make_params(size[0], size[1])
    counts, _ = devs[preset].render_counts(hosts[preset].camera, p)
    rgb, mx = ptamd.heatmap_colour(counts)
    want = ptamd.quantise_rows(rgb)
    got, got_max, st = devs[preset].render_counts_bgr8(hosts[preset].camera, p)
    assert got_max == mx == counts.max()
    assert got.shape == want.shape and got.shape[1] % 4 == 0
    assert np.array_equal(got, want)
    assert st.render_s > 0


def test_radiance_bytes_equal_the_host_stage(gpu_scenes):
    hosts, devs = gpu_scenes
    p = ptamd.make_params(150, 80, spp=8, max_bounces=4)
    img, _ = devs[2].render(hosts[2].camera, p)
    got, _ = devs[2].render_bgr8(hosts[2].camera, p)
    assert np.array_equal(got, ptamd.quantise_rows(img))
    # scene 1 has out-of-range radiance before the clamp
    img, _ = devs[1].render(hosts[1].camera, p)
    got, _ = devs[1].render_bgr8(hosts[1].camera, p)
    assert np.array_equal(got, ptamd.quantise_rows(img))


def test_bmp_file_from_device_rows_equals_the_host_writer(gpu_scenes, tmp_path):
    hosts, devs = gpu_scenes
    p = ptamd.make_params(101, 50)
    counts, _ = devs[2].render_counts(hosts[2].camera, p)
    rgb, _ = ptamd.heatmap_colour(counts)
    a, b = str(tmp_path / "host.bmp"), str(tmp_path / "device.bmp")
    ptamd.write_bmp(a, rgb)
    rows, _, _ = devs[2].render_counts_bgr8(hosts[2].camera, p)
    assert ptamd.host_lib().pth_write_bmp_rows(b.encode(), rows.ctypes.data, 101, 50) == 0
    assert open(a, "rb").read() == open(b, "rb").read()
    assert os.path.getsize(a) == 54 + 304 * 50


def test_row_bytes_too_small_is_rejected(gpu_scenes):
    import ctypes as C
    hosts, devs = gpu_scenes
    p = ptamd.make_params(16, 16)
    buf = np.zeros(16 * 48, dtype=np.uint8)
    mx = C.c_uint32()
    rc = ptamd.backend_lib().ptx_render_counts_bgr8(devs[2]._h, C.byref(hosts[2].camera), C.byref(p), buf.ctypes.data_as(C.c_void_p),
                                                    47, C.byref(mx), None)
    assert rc != 0
