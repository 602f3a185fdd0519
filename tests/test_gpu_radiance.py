"""Radiance parity, HIP (through the C ABI) vs the CPU oracle.

Tolerances (stated here as BASELINE.json's north_star asks):
 * same seeding on both sides (reference GPU seeding, 1023 + pixel): the two sides draw the same
   random numbers, so images agree to float rounding except where a 1-ulp difference between
   glibc's and the device library's cosf/sinf/powf/atan2f/acosf flips a discrete decision on a path.
   Scene 2 / scene 0: per-pixel RMSE < 2e-3 and >= 99% of pixels within 2e-3.
 * different seeding (reference CPU `-d 1 -t 1` seeding vs GPU seeding): statistical agreement only.
   At 256 spp on scene 2: per-pixel RMSE < 0.03, |mean bias| < 3e-3 (tonemapped [0,1] units).
"""
import os

import numpy as np
import pytest

import oracle_binding as ob
import ptamd

pytestmark = pytest.mark.gpu


def rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


@pytest.mark.parametrize("preset,size,spp,bounces,tol_rmse,tol_frac", [
    (2, (128, 96), 8, 4, 2e-3, 0.99),
    (2, (512, 512), 16, 4, 2e-3, 0.99),   # BASELINE.json configs[0] geometry
    (0, (150, 80), 8, 6, 2e-3, 0.99),
    # scene 1 is looser for ONE reason, and it is shown, not assumed: glibc's and the device library's cosf / sinf / atan2f /
    # acosf / powf differ by an ulp now and then, and a later bounce on a curved mirror turns that into another hit.  With one
    # implementation of those five functions on both sides the accumulation buffers of this scene are identical BIT FOR BIT
    # (tests/test_gpu_shared_libm.py: 0 of 48,000 pixels differ; 12,112 with each side's own libm).  "libm only".
    (1, (150, 80), 4, 6, 3e-2, 0.85),
    (2, (64, 64), 1, 4, 2e-3, 0.99),      # spp == 1: no jitter (gpuRender.cu:134)
    (2, (33, 17), 5, 1, 2e-3, 0.99),      # one bounce: hit shading only, never the environment after a hit
])
def test_same_seed_images_agree(gpu_scenes, preset, size, spp, bounces, tol_rmse, tol_frac):
    hosts, devs = gpu_scenes
    p = ptamd.make_params(size[0], size[1], spp=spp, max_bounces=bounces)
    ref, st_ref = ob.render(hosts[preset], p, seed_mode=ptamd.SEED_REFERENCE_GPU)
    got, _ = devs[preset].render(hosts[preset].camera, p)
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert got.min() >= 0.0 and got.max() <= 1.0
    d = np.abs(got - ref).max(axis=2)
    assert rmse(got, ref) < tol_rmse
    assert (d < 2e-3).mean() >= tol_frac
    st = devs[preset].measure(hosts[preset].camera, p)
    assert abs(int(st.rays) - int(st_ref.rays)) <= max(8, int(0.02 * st_ref.rays))
    assert st.hits + st.misses == st.rays


def test_against_committed_golden_image(gpu_scenes):
    hosts, devs = gpu_scenes
    gold = np.load(os.path.join(ptamd.GOLDEN_DIR, "scene2_radiance_64_256spp.npy"))
    got, _ = devs[2].render(hosts[2].camera, ptamd.make_params(64, 64, spp=256, max_bounces=4))
    assert rmse(got, gold) < 2e-3
    assert (np.abs(got - gold).max(axis=2) < 2e-3).mean() >= 0.99


def test_statistical_agreement_with_the_cpu_seeding(gpu_scenes):
    hosts, devs = gpu_scenes
    p = ptamd.make_params(96, 96, spp=256, max_bounces=4)
    ref, _ = ob.render(hosts[2], p, seed_mode=ob.SEED_REFERENCE_CPU, threads=1)   # `-d 1 -t 1`
    got, _ = devs[2].render(hosts[2].camera, p)
    assert rmse(got, ref) < 0.03
    assert abs(float(got.mean() - ref.mean())) < 3e-3


def test_zero_bounces_and_device_buffers(gpu_scenes):
    torch = pytest.importorskip("torch")
    hosts, devs = gpu_scenes
    # max_bounces == 0: every path returns white (gpuRender.cu:27-29)
    got, _ = devs[2].render(hosts[2].camera, ptamd.make_params(16, 16, spp=3, max_bounces=0))
    one = np.zeros(3, dtype=np.float32)
    import ctypes as C
    s = np.array([3.0, 3.0, 3.0], dtype=np.float32)
    ob.lib().oracle_tonemap(s.ctypes.data, C.c_float(3.0), one.ctypes.data)
    assert np.abs(got - one).max() < 1e-6
    # accumulate + resolve on caller-owned device memory == ptx_render
    p = ptamd.make_params(80, 48, spp=6, max_bounces=4)
    acc = torch.zeros((48, 80, 4), dtype=torch.float32, device="cuda")
    out = torch.zeros((48, 80, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    devs[2].accumulate_device(hosts[2].camera, p, acc.data_ptr(), stream)
    devs[2].resolve_device(acc.data_ptr(), 80, 48, out.data_ptr(), stream)
    torch.cuda.synchronize()
    direct, _ = devs[2].render(hosts[2].camera, p)
    assert np.array_equal(out.cpu().numpy(), direct)
    assert (acc[..., 3].cpu().numpy() == 6).all()
    # and matches the oracle's un-normalised sums
    ref, _ = ob.accumulate(hosts[2], p, seed_mode=ptamd.SEED_REFERENCE_GPU)
    a = acc.cpu().numpy()
    assert (np.abs(a - ref).max(axis=2) < 1e-2).mean() >= 0.99


def test_sample_slice_streams(gpu_scenes):
    torch = pytest.importorskip("torch")
    import multigpu
    hosts, devs = gpu_scenes
    W, H, SPP, B = 96, 64, 8, 4
    acc = np.zeros((H, W, 4), dtype=np.float32)
    ref = np.zeros((H, W, 4), dtype=np.float32)
    buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    for r in range(3):
        p = ptamd.make_params(W, H, spp=SPP, max_bounces=B, seed_mode=ptamd.SEED_STREAM, part_index=r, part_count=3)
        devs[2].accumulate_device(hosts[2].camera, p, buf.data_ptr(), None)
        torch.cuda.synchronize()
        acc += buf.cpu().numpy()
        a, _ = ob.accumulate(hosts[2], p, seed_mode=ptamd.SEED_STREAM)
        ref += a
    got = ob.resolve(acc)
    exp = ob.resolve(ref)
    assert rmse(got, exp) < 2e-3
    # stream 0 == reference seeding
    p = ptamd.make_params(W, H, spp=SPP, max_bounces=B, seed_mode=ptamd.SEED_STREAM, part_index=0, part_count=3)
    devs[2].accumulate_device(hosts[2].camera, p, buf.data_ptr(), None)
    torch.cuda.synchronize()
    s0 = buf.cpu().numpy().copy()
    p = ptamd.make_params(W, H, spp=SPP, max_bounces=B)
    devs[2].accumulate_device(hosts[2].camera, p, buf.data_ptr(), None)
    torch.cuda.synchronize()
    assert np.array_equal(s0, buf.cpu().numpy())


def test_sub_streams_equal_the_sum_of_their_streams(gpu_scenes):
    """ptx_params.sub_streams: S RNG streams per pixel inside one call == S single-stream calls with
    stream ids part_index*S + k and the same split of the samples, added in stream order."""
    torch = pytest.importorskip("torch")
    hosts, devs = gpu_scenes
    W, H, SPP, B, S, PART = 96, 64, 10, 4, 4, 1
    buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    p = ptamd.make_params(W, H, spp=SPP, max_bounces=B, seed_mode=ptamd.SEED_STREAM, part_index=PART, part_count=2,
                          sub_streams=S)
    devs[2].accumulate_device(hosts[2].camera, p, buf.data_ptr(), None)
    torch.cuda.synchronize()
    got = buf.cpu().numpy().copy()
    assert np.all(got[..., 3] == SPP)
    gpu_sum = np.zeros((H, W, 4), dtype=np.float32)
    cpu_sum = np.zeros((H, W, 4), dtype=np.float32)
    for k in range(S):
        n = SPP // S + (1 if k < SPP % S else 0)
        pk = ptamd.make_params(W, H, spp=n, max_bounces=B, seed_mode=ptamd.SEED_STREAM, part_index=PART * S + k,
                               part_count=2 * S)
        devs[2].accumulate_device(hosts[2].camera, pk, buf.data_ptr(), None)
        torch.cuda.synchronize()
        gpu_sum += buf.cpu().numpy()
        a, _ = ob.accumulate(hosts[2], pk, seed_mode=ptamd.SEED_STREAM)
        cpu_sum += a
    assert np.array_equal(got, gpu_sum)  # same streams, same order of additions
    assert rmse(ob.resolve(got), ob.resolve(cpu_sum)) < 2e-3
    # the host-buffer entry point goes through the same planes + combine path
    img, _ = devs[2].render(hosts[2].camera, p)
    assert np.abs(img - ob.resolve(got)).max() < 1e-5
    # more streams than samples: clamped, no empty streams
    p1 = ptamd.make_params(W, H, spp=2, max_bounces=B, seed_mode=ptamd.SEED_STREAM, sub_streams=16)
    devs[2].accumulate_device(hosts[2].camera, p1, buf.data_ptr(), None)
    torch.cuda.synchronize()
    assert np.all(buf.cpu().numpy()[..., 3] == 2)
    # the reference's seeding has one stream per pixel
    with pytest.raises(Exception):
        devs[2].render(hosts[2].camera, ptamd.make_params(W, H, spp=4, max_bounces=B, sub_streams=2))


def test_single_rank_orchestration_on_gpu(gpu_scenes):
    pytest.importorskip("torch")
    import multigpu
    hosts, devs = gpu_scenes
    be = multigpu.HipBackend(hosts[2], 0)
    img = multigpu.render_radiance(be, ptamd.make_params, 64, 48, 4, 4, 0, 1)
    cnt = multigpu.render_counts(be, ptamd.make_params, 64, 48, 0, 1)
    direct, _ = devs[2].render(hosts[2].camera, ptamd.make_params(64, 48, spp=4, max_bounces=4))
    assert np.array_equal(multigpu.to_numpy(img), direct)
    ref, _ = ob.render_counts(hosts[2], ptamd.make_params(64, 48))
    assert np.array_equal(multigpu.to_numpy(cnt).astype(np.uint32), ref)
    be.close()


def test_repeated_calls_are_bit_identical(gpu_scenes):
    """No float atomics anywhere: lanes pull work in a different order on every launch, the result may not
    depend on it (one lane per item, planes added in stream order)."""
    torch = pytest.importorskip("torch")
    hosts, devs = gpu_scenes
    buf = torch.zeros((90, 160, 4), dtype=torch.float32, device="cuda")
    for preset, kw in ((1, dict(seed_mode=ptamd.SEED_STREAM, sub_streams=5)), (2, dict())):
        p = ptamd.make_params(160, 90, spp=10, max_bounces=6, **kw)
        outs = []
        for _ in range(3):
            devs[preset].accumulate_device(hosts[preset].camera, p, buf.data_ptr(), None)
            torch.cuda.synchronize()
            outs.append(buf.cpu().numpy().copy())
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    c0, _ = devs[1].render_counts(hosts[1].camera, ptamd.make_params(160, 90))
    c1, _ = devs[1].render_counts(hosts[1].camera, ptamd.make_params(160, 90))
    assert np.array_equal(c0, c1)
