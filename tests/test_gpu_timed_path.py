"""Parity of the configurations that are TIMED and of BASELINE.json's largest shape (configs[3], [4]).

bench.py times scene 1 on the wide (four-wide, quantised) BVH layout with RNG sub-streams; BASELINE's
last configuration is scene 1 at 3840x2160, 8 bounces, split over 8 GPUs.  The tests here run exactly
those code paths through the C ABI against the CPU oracle:

 * same seeds: sub-stream launch on scene 1 == the sum of the oracle's streams.  Tolerance (tonemapped
   [0,1] units): per-pixel RMSE < 3e-2, |mean difference| < 2e-3, and at most 2 % of the PATHS may part
   from the oracle's: a pixel of n samples must be within 2e-3 with probability >= 0.98^n (scene 0:
   >= 99 % of the pixels outright).  Scene 1 is 10,000 small curved busts, half of them metal (GGX,
   roughness 0.05-0.2): a 1-ulp difference between glibc's and the device library's cosf/sinf/powf
   moves the next bounce, every curved mirror amplifies it, and after a few bounces the path hits
   another triangle (measured: 0.6 % of the paths at 6 bounces, 1.4 % at 8).  Such a path is another
   sample of the SAME distribution, which is what the high-spp tests below check.  That libm is the ONLY cause is shown by
   tests/test_gpu_shared_libm.py: built with one implementation of those functions, oracle and kernels produce identical
   accumulation buffers, bit for bit, on scenes 0-2 (and the shipped four-wide layout alone changes 18 of 48,000 pixels of
   scene 1, none of scenes 0 and 2).  The tolerances here are "libm only".
 * different seeds, high spp (north_star: "a stated per-pixel RMSE at high spp"): the GPU frame with
   the bench's seeding (RNG sub-streams, wide layout) against the oracle with the reference CPU path's
   own seeding (`-d 1 -t 1`), scenes 0 and 1, 512 spp, 6 bounces.  Two independent estimates can only
   agree to their Monte-Carlo noise, so the noise is measured (a second GPU frame with other streams)
   and the bar is relative to it:  RMSE(gpu, cpu) <= 1.15 * RMSE(gpu, gpu') + 1e-3,  and
   |mean(gpu) - mean(cpu)| < 1.5e-3  (a systematic error would show in the mean long before it shows
   in the RMSE), and per-channel |mean difference| < 2.5e-3.
 * 3840x2160 x 8 bounces: heat map bit-exact at full size, radiance properties at full size, same-seed
   radiance at a reduced size with 8 bounces.
"""
import numpy as np
import pytest

import oracle_binding as ob
import ptamd

pytestmark = pytest.mark.gpu


def rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def _accumulate(dev, host, p, torch, noalloc=False):
    buf = torch.zeros((p.height, p.width, 4), dtype=torch.float32, device="cuda")
    (dev.accumulate_device_noalloc if noalloc else dev.accumulate_device)(host.camera, p, buf.data_ptr(), None)
    torch.cuda.synchronize()
    return buf.cpu().numpy()


@pytest.mark.parametrize("preset,size,spp,streams,part", [
    (1, (300, 160), 16, 8, 0),    # the bench's scene, wide layout, 8 streams of 2 samples
    (1, (150, 80), 25, 25, 3),    # one sample per stream (jitter still on: the frame has 25), another part
    (0, (150, 80), 12, 5, 1),     # uneven split: streams of 3, 3, 2, 2, 2
])
def test_sub_streams_on_the_timed_scene_match_the_oracle_streams(gpu_scenes, preset, size, spp, streams, part):
    torch = pytest.importorskip("torch")
    hosts, devs = gpu_scenes
    W, H = size
    B = 6
    p = ptamd.make_params(W, H, spp=spp, max_bounces=B, seed_mode=ptamd.SEED_STREAM, part_index=part, part_count=4,
                          sub_streams=streams)
    got = _accumulate(devs[preset], hosts[preset], p, torch)
    assert np.all(got[..., 3] == spp)
    cpu = np.zeros((H, W, 4), dtype=np.float32)
    rays = 0
    for k in range(streams):
        n = spp // streams + (1 if k < spp % streams else 0)
        pk = ptamd.make_params(W, H, spp=n, max_bounces=B, seed_mode=ptamd.SEED_STREAM, part_index=part * streams + k,
                               part_count=4 * streams, frame_spp=spp)
        a, st = ob.accumulate(hosts[preset], pk, seed_mode=ptamd.SEED_STREAM)
        cpu += a
        rays += st.rays
    a, b = ob.resolve(got), ob.resolve(cpu)
    d = np.abs(a - b).max(axis=2)
    print(f"scene {preset} {W}x{H} {spp} spp in {streams} streams: rmse {rmse(a, b):.2e}, within 2e-3: {(d < 2e-3).mean():.4f}, "
          f"mean diff {float(a.mean() - b.mean()):+.2e}")
    assert rmse(a, b) < 3e-2
    assert (d < 2e-3).mean() >= (0.98 ** spp if preset == 1 else 0.99)
    assert abs(float(a.mean() - b.mean())) < 2e-3
    st = devs[preset].measure(hosts[preset].camera, p)
    assert abs(int(st.rays) - int(rays)) <= max(8, int(0.02 * rays))


@pytest.mark.parametrize("preset,S", [(0, 16), (1, 16), (1, 256)])   # 256 streams of two samples: the bench's short items
def test_high_spp_agreement_with_the_cpu_path_seeding(gpu_scenes, preset, S):
    torch = pytest.importorskip("torch")
    hosts, devs = gpu_scenes
    W, H, SPP, B = 120, 64, 512, 6
    cpu, _ = ob.render(hosts[preset], ptamd.make_params(W, H, spp=SPP, max_bounces=B), seed_mode=ob.SEED_REFERENCE_CPU,
                       threads=1)  # `-d 1 -t 1`: the only deterministic form of the reference CPU path
    frames = []
    for part in (0, 1):
        p = ptamd.make_params(W, H, spp=SPP, max_bounces=B, seed_mode=ptamd.SEED_STREAM, part_index=part, part_count=2,
                              sub_streams=S)
        frames.append(ob.resolve(_accumulate(devs[preset], hosts[preset], p, torch)))
    gpu, gpu2 = frames
    noise = rmse(gpu, gpu2)
    err = rmse(gpu, cpu)
    bias = float(gpu.mean() - cpu.mean())
    chan = [float(gpu[..., c].mean() - cpu[..., c].mean()) for c in range(3)]
    print(f"scene {preset}: rmse(gpu,cpu) {err:.4f}  rmse(gpu,gpu') {noise:.4f}  mean diff {bias:+.2e}  per channel {chan}")
    assert err <= 1.15 * noise + 1e-3
    assert abs(bias) < 1.5e-3
    assert max(abs(c) for c in chan) < 2.5e-3


def test_c5_shape_heat_map_is_bit_exact_at_3840x2160(gpu_scenes):
    hosts, devs = gpu_scenes
    p = ptamd.make_params(3840, 2160)
    ref, st_ref = ob.render_counts(hosts[1], p, threads=16)
    got, _ = devs[1].render_counts(hosts[1].camera, p)
    assert np.array_equal(ref, got)
    st = devs[1].measure(hosts[1].camera, p, heatmap=True)
    assert (st.node_steps, st.tri_tests, st.inst_entries) == (st_ref.node_steps, st_ref.tri_tests, st_ref.inst_entries)
    # tile partition over 8 parts at this size reassembles exactly
    total = np.zeros_like(got)
    for part in range(8):
        c, _ = devs[1].render_counts(hosts[1].camera, ptamd.make_params(3840, 2160, part_index=part, part_count=8))
        total += c
    assert np.array_equal(total, got)


def test_c5_shape_radiance_at_3840x2160_8_bounces(gpu_scenes):
    """One GPU's share of BASELINE's last configuration, cut down in samples only: 8.29 M pixels x 2 RNG
    streams (16.6 M work items, pixel index and stream packed in one dword), 8 bounces."""
    torch = pytest.importorskip("torch")
    hosts, devs = gpu_scenes
    W, H = 3840, 2160
    p = ptamd.make_params(W, H, spp=2, max_bounces=8, seed_mode=ptamd.SEED_STREAM, part_index=5, part_count=8, sub_streams=2,
                          frame_spp=1024)
    a = _accumulate(devs[1], hosts[1], p, torch)
    assert np.isfinite(a).all() and (a[..., :3] >= 0).all()
    assert np.all(a[..., 3] == 2)  # sample counts exact in channel 3
    img = ob.resolve(a)
    assert img.min() >= 0.0 and img.max() <= 1.0 and img.std() > 0.01
    b = _accumulate(devs[1], hosts[1], p, torch)
    assert np.array_equal(a, b)  # repeatable: no float atomics, planes added in stream order
    # every ray ends in a shaded hit or an environment lookup; every path has at least its camera ray
    st = devs[1].measure(hosts[1].camera, p)
    assert st.hits + st.misses == st.rays and st.rays >= 2 * W * H


@pytest.mark.parametrize("preset,size,spp", [(1, (192, 108), 4), (2, (96, 54), 8)])
def test_same_seed_images_agree_with_8_bounces(gpu_scenes, preset, size, spp):
    hosts, devs = gpu_scenes
    p = ptamd.make_params(size[0], size[1], spp=spp, max_bounces=8)
    ref, st_ref = ob.render(hosts[preset], p, seed_mode=ptamd.SEED_REFERENCE_GPU)
    got, _ = devs[preset].render(hosts[preset].camera, p)
    d = np.abs(got - ref).max(axis=2)
    print(f"scene {preset} 8 bounces: rmse {rmse(got, ref):.2e}, within 2e-3: {(d < 2e-3).mean():.4f}")
    tol_rmse, tol_frac = (3e-2, 0.98 ** spp) if preset == 1 else (2e-3, 0.99)  # see the module docstring
    assert rmse(got, ref) < tol_rmse and (d < 2e-3).mean() >= tol_frac
    st = devs[preset].measure(hosts[preset].camera, p)
    assert abs(int(st.rays) - int(st_ref.rays)) <= max(8, int(0.02 * st_ref.rays))
