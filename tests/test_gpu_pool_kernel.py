"""poolKernel (csrc/pool_kernel.hpp), the shipped radiance kernel: a private pool of chains per wave, rays traced by whatever
lane is free, shading in batches of 64.  Per chain the random numbers, the order of the additions and every rounding are
those of pathKernel (csrc/kernels.hpp, the kernel of rounds 1 and 2, still the `-a` kernel and PTX_QUEUE=0), so for equal
seeds the two must agree BIT FOR BIT -- which also makes every parity test of pathKernel against the oracle a test of this
kernel.  All through the C ABI; the kernel is chosen per launch by the environment."""
import os

import numpy as np
import pytest

import ptamd

pytestmark = pytest.mark.gpu


def render_with(dev, host, p, kernel, env=None):
    saved = {k: os.environ.get(k) for k in ["PTX_QUEUE", *(env or {})]}
    os.environ["PTX_QUEUE"] = "1" if kernel == "pool" else "0"
    os.environ.update(env or {})
    try:
        return dev.render(host.camera, p)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


CASES = [  # scene, width, height, spp, bounces, streams
    (2, 64, 48, 1, 1, 0),      # one sample: no jitter (gpuRender.cu:134)
    (2, 160, 96, 4, 4, 0),
    (2, 160, 96, 9, 6, 4),     # sub-streams of unequal length (3, 2, 2, 2)
    (2, 37, 29, 5, 0, 0),      # no bounce at all: every path returns white (gpuRender.cu:27-29)
    (0, 240, 128, 8, 6, 0),
    (1, 240, 128, 8, 6, 3),
    (1, 333, 77, 5, 3, 0),     # ragged frame
    (1, 1, 1, 7, 6, 0),        # one pixel: 63 of a wave's lanes never get a chain
    (1, 200, 120, 40, 6, 0),   # long items: scattered over the waves, three chains per lane, urgent list (pool_kernel.hpp)
    (1, 200, 120, 40, 6, 13),
    (1, 1500, 800, 16, 6, 0),  # the reference's seeding at full size: a static partition, every wave of the chip in the priority feedback
]


@pytest.mark.parametrize("scene,w,h,spp,bounces,streams", CASES)
def test_pool_kernel_equals_path_kernel_bit_for_bit(gpu_scenes, scene, w, h, spp, bounces, streams):
    hosts, devs = gpu_scenes
    p = ptamd.make_params(w, h, spp=spp, max_bounces=bounces, seed_mode=ptamd.SEED_STREAM if streams > 1 else ptamd.SEED_REFERENCE_GPU,
                          sub_streams=streams)
    a, _ = render_with(devs[scene], hosts[scene], p, "path")
    b, _ = render_with(devs[scene], hosts[scene], p, "pool")
    assert np.isfinite(b).all()
    assert np.array_equal(a, b), int((a != b).any(axis=2).sum())


@pytest.mark.parametrize("env", [{"PTX_POOL_CHAINS": "1"}, {"PTX_POOL_CHAINS": "3", "PTX_SCATTER": "1"}, {"PTX_POOL_CHAINS": "2", "PTX_SCATTER": "0"},
                                 {"PTX_POOL_PACED": "0"}])  # the default of this frame is pacedPoolKernel (a static partition)
def test_pool_kernel_result_does_not_depend_on_its_scheduling(gpu_scenes, env):
    """Chains per lane, the order in which pixels are handed to the waves, which chains are served first and which waves the
    issue arbiter prefers decide WHEN a ray is traced, never what it returns."""
    hosts, devs = gpu_scenes
    p = ptamd.make_params(320, 200, spp=24, max_bounces=6)
    ref, _ = render_with(devs[1], hosts[1], p, "path")
    got, _ = render_with(devs[1], hosts[1], p, "pool", env)
    assert np.array_equal(ref, got)


def test_pool_kernel_counts_the_same_work(gpu_scenes):
    """The instrumented variants of both kernels: rays, node steps, triangle tests, instance entries, hits and misses are
    properties of the paths, not of the kernel that traces them."""
    hosts, devs = gpu_scenes
    p = ptamd.make_params(400, 240, spp=6, max_bounces=6, seed_mode=ptamd.SEED_STREAM, sub_streams=2)
    saved = os.environ.get("PTX_QUEUE")
    try:
        os.environ["PTX_QUEUE"] = "0"
        a = devs[1].measure(hosts[1].camera, p)
        os.environ["PTX_QUEUE"] = "1"
        b = devs[1].measure(hosts[1].camera, p)
    finally:
        if saved is None:
            os.environ.pop("PTX_QUEUE", None)
        else:
            os.environ["PTX_QUEUE"] = saved
    for k in ("rays", "node_steps", "tri_tests", "inst_entries", "hits", "misses", "box_hits"):
        assert getattr(a, k) == getattr(b, k), k
    assert b.hits + b.misses == b.rays and b.rays >= 400 * 240 * 6


def test_config2_scene0_full_size(gpu_scenes):
    """BASELINE.json configs[1]: scene 0 (three instances of the bust), 1500x800, 100 spp, 6 bounces on one GPU, at full size,
    with the reference's seeding.  No oracle at this size: properties -- finite, in [0, 1], every pixel carries 100 samples,
    the frame is repeatable to the bit and equal to pathKernel's, and hits + misses = rays, between one and six per path."""
    torch = pytest.importorskip("torch")
    hosts, devs = gpu_scenes
    p = ptamd.make_params(1500, 800, spp=100, max_bounces=6)
    img, st = render_with(devs[0], hosts[0], p, "pool")
    again, _ = render_with(devs[0], hosts[0], p, "pool")
    old, st_old = render_with(devs[0], hosts[0], p, "path")
    assert np.isfinite(img).all() and img.min() >= 0.0 and img.max() <= 1.0 and img.std() > 0.01
    assert np.array_equal(img, again) and np.array_equal(img, old)
    accum = torch.zeros((800, 1500, 4), dtype=torch.float32, device="cuda")
    devs[0].accumulate_device(hosts[0].camera, p, accum.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    a = accum.cpu().numpy()
    assert (a[..., 3] == 100.0).all() and np.isfinite(a).all() and (a[..., :3] >= 0.0).all()
    m = devs[0].measure(hosts[0].camera, p)
    paths = 1500 * 800 * 100
    assert m.hits + m.misses == m.rays and paths <= m.rays <= 6 * paths
    print(f"config 2: {m.rays / 1e6:.1f} M rays, poolKernel {st.render_s * 1e3:.1f} ms = {m.rays / st.render_s / 1e6:.0f} Mrays/s, "
          f"pathKernel {st_old.render_s * 1e3:.1f} ms")
