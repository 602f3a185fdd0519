"""ptx_accumulate_device_noalloc (include/ptx_backend.h): the entry point that allocates nothing.

Same kernels as ptx_accumulate_device, scratch (work counter, traversal-stack overflow area, stream
planes) in the library's static per-GPU array instead of hipMalloc'ed buffers.  Bar: bit-identical
accumulation buffers (float work, but the same kernels in the same order on the same inputs: nothing
may differ), on two calls in a row, the second with a smaller input (the array keeps the first call's
contents), plus a clear refusal beyond the array's size.
"""
import numpy as np
import pytest

import ptamd

pytestmark = pytest.mark.gpu


def _both(dev, host, p, torch):
    a = torch.full((p.height, p.width, 4), -7.0, dtype=torch.float32, device="cuda")
    b = torch.full((p.height, p.width, 4), -9.0, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    dev.accumulate_device(host.camera, p, a.data_ptr(), stream)
    dev.accumulate_device_noalloc(host.camera, p, b.data_ptr(), stream)
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


def test_noalloc_equals_the_allocating_entry_point(gpu_scenes):
    torch = pytest.importorskip("torch")
    hosts, devs = gpu_scenes
    # scene 1: deep TLAS + BLAS (the stack overflow area is in use), RNG sub-streams (planes in the scratch array)
    big = ptamd.make_params(320, 200, spp=12, max_bounces=6, seed_mode=ptamd.SEED_STREAM, sub_streams=4)
    small = ptamd.make_params(96, 64, spp=5, max_bounces=6, seed_mode=ptamd.SEED_STREAM, sub_streams=2)
    for p in (big, small):  # two calls in a row, the second smaller
        ref, got = _both(devs[1], hosts[1], p, torch)
        assert np.isfinite(got).all()
        assert np.all(got[..., 3] == p.spp)
        assert np.array_equal(ref, got)
    # one stream per pixel, reference seeding: the kernel writes d_accum itself (no planes); another scene handle
    # shares the same array
    p1 = ptamd.make_params(128, 96, spp=4, max_bounces=4)
    ref, got = _both(devs[2], hosts[2], p1, torch)
    assert np.array_equal(ref, got)
    # and once more on the first scene, smaller again
    tiny = ptamd.make_params(33, 17, spp=3, max_bounces=6, seed_mode=ptamd.SEED_STREAM, sub_streams=3)
    ref, got = _both(devs[1], hosts[1], tiny, torch)
    assert np.array_equal(ref, got)


def test_noalloc_refuses_what_does_not_fit(gpu_scenes):
    torch = pytest.importorskip("torch")
    hosts, devs = gpu_scenes
    # planes: 25 streams x 1500 x 800 x 16 B = 480 MB > the 256 MiB array
    p = ptamd.make_params(1500, 800, spp=100, max_bounces=6, seed_mode=ptamd.SEED_STREAM, sub_streams=25)
    buf = torch.zeros((800, 1500, 4), dtype=torch.float32, device="cuda")
    with pytest.raises(ptamd.BackendError, match="scratch"):
        devs[1].accumulate_device_noalloc(hosts[1].camera, p, buf.data_ptr(), None)
    # the same frame with one stream per pixel fits (stack overflow area only)
    p = ptamd.make_params(1500, 800, spp=1, max_bounces=2, seed_mode=ptamd.SEED_STREAM)
    devs[1].accumulate_device_noalloc(hosts[1].camera, p, buf.data_ptr(), None)
    torch.cuda.synchronize()
    assert np.all(buf[..., 3].cpu().numpy() == 1)


def test_noalloc_allocates_nothing(gpu_scenes):
    """The property itself, not only the result: on a FRESH scene handle (no launch yet: its own overflow area and planes are
    still unallocated, so a call that fell back to them would have to hipMalloc megabytes) the device's free memory is the
    same before and after the call, while the allocating entry point on another fresh handle takes memory for the same frame."""
    torch = pytest.importorskip("torch")
    hosts, _ = gpu_scenes
    p = ptamd.make_params(640, 400, spp=8, max_bounces=6, seed_mode=ptamd.SEED_STREAM, sub_streams=4)
    out = torch.zeros((p.height, p.width, 4), dtype=torch.float32, device="cuda")
    out2 = torch.zeros_like(out)
    fresh, other = ptamd.DeviceScene(hosts[1].desc_ptr), ptamd.DeviceScene(hosts[1].desc_ptr)
    try:
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        fresh.accumulate_device_noalloc(hosts[1].camera, p, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        assert free1 == free0, (free0, free1)
        other.accumulate_device(hosts[1].camera, p, out2.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        free2, _ = torch.cuda.mem_get_info()
        assert free2 < free1 - (4 << 20), (free1, free2)   # planes (16 MB) + stack overflow area + the chain records
        assert np.array_equal(out.cpu().numpy(), out2.cpu().numpy())
        assert ptamd.noalloc_scratch_bytes() == 256 << 20
    finally:
        fresh.close()
        other.close()
