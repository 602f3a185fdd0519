"""The C++ side of the boundary on the GPU: the PathTracer binary (host/main.cpp -> renderGPU / renderGPURows ->
C ABI) and the one-process multi-GPU entry points (ptx_render_multi, ptx_render_counts_multi).

The reference's main.cpp:240-379 prints a fixed set of lines and writes output.bmp; the binary is run in a
scratch directory and its file is compared byte for byte with what the same C-ABI calls return through ctypes.
-g N is rehearsed on this one-GPU box with PT_MULTI_SAME_DEVICE=1 (every part on device 0: the split, the RNG
streams, the device-to-device gather and the root's combine are the real code; only the copies are local).
"""
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import ptamd

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(REPO, "path-tracer_amd", "PathTracer")


def run_cli(tmp_path, *args, **env):
    e = dict(os.environ, PT_MODELS_DIR=ptamd.MODELS_DIR, PT_ENV_PATH=ptamd.ENV_PATH, **env)
    out = subprocess.run([BINARY, *args], cwd=tmp_path, env=e, capture_output=True, text=True, timeout=300)
    return out


def bmp_rows(path, width, height):
    data = np.fromfile(path, dtype=np.uint8)
    row_bytes = ptamd.host_lib().pth_bmp_row_bytes(width)
    assert data[:2].tobytes() == b"BM" and data.size == 54 + row_bytes * height
    assert int.from_bytes(data[18:22].tobytes(), "little") == width and int.from_bytes(data[22:26].tobytes(), "little") == height
    return data[54:].reshape(height, row_bytes)


def check_console(stdout, width, height, samples, bounces):
    assert f"Dimensions: [{width}, {height}]\tSamples: {samples}\tBounces: {bounces}" in stdout
    assert "Rendering..." in stdout and "Output result to output.bmp" in stdout
    # gpuRender.cu:267,283: "\nGPU data transfer time: <s> s\n" / "\nRender time: <s> s\n", seconds floored to 4 decimals
    m = re.search(r"\nRender time: ([0-9.e+-]+) s\n", stdout)
    assert m and re.search(r"\nGPU data transfer time: ([0-9.e+-]+) s\n", stdout)
    secs = float(m.group(1))
    assert secs >= 0.0 and abs(secs * 1e4 - round(secs * 1e4)) < 1e-6


@pytest.mark.parametrize("host_output", ["0", "1"])
def test_cli_heat_map_scene2(tmp_path, gpu_scenes, host_output):
    hosts, devs = gpu_scenes
    out = run_cli(tmp_path, "-p", "2", "-a", "-w", "1500", "-h", "800", PT_HOST_OUTPUT=host_output)
    assert out.returncode == 0, out.stderr
    assert "Maximum BVH tests: 123" in out.stdout  # README.md:60
    check_console(out.stdout, 1500, 800, 1, 1)      # -a forces one sample, one bounce (main.cpp:296-299)
    rows, mx, _ = devs[2].render_counts_bgr8(hosts[2].camera, ptamd.make_params(1500, 800))
    assert mx == 123
    assert np.array_equal(bmp_rows(tmp_path / "output.bmp", 1500, 800), rows)


@pytest.mark.parametrize("host_output", ["0", "1"])
def test_cli_radiance_scene2(tmp_path, gpu_scenes, host_output):
    hosts, devs = gpu_scenes
    out = run_cli(tmp_path, "-p", "2", "-w", "128", "-h", "96", "-s", "8", "-b", "4", PT_HOST_OUTPUT=host_output)
    assert out.returncode == 0, out.stderr
    check_console(out.stdout, 128, 96, 8, 4)
    rows, _ = devs[2].render_bgr8(hosts[2].camera, ptamd.make_params(128, 96, spp=8, max_bounces=4))
    assert np.array_equal(bmp_rows(tmp_path / "output.bmp", 128, 96), rows)


def test_cli_sub_streams_flag_is_the_fast_path(tmp_path, gpu_scenes):
    hosts, devs = gpu_scenes
    out = run_cli(tmp_path, "-p", "1", "-w", "160", "-h", "90", "-s", "12", "-b", "6", "-r", "4")
    assert out.returncode == 0, out.stderr
    p = ptamd.make_params(160, 90, spp=12, max_bounces=6, seed_mode=ptamd.SEED_STREAM, sub_streams=4)
    rows, _ = devs[1].render_bgr8(hosts[1].camera, p)
    assert np.array_equal(bmp_rows(tmp_path / "output.bmp", 160, 90), rows)


def test_cli_two_gpus_rehearsed_on_one(tmp_path, gpu_scenes):
    hosts, devs = gpu_scenes
    out = run_cli(tmp_path, "-p", "2", "-a", "-w", "333", "-h", "77", "-g", "2", PT_MULTI_SAME_DEVICE="1")
    assert out.returncode == 0, out.stderr
    assert "on 2 GPUs" in out.stdout
    rows, mx, _ = devs[2].render_counts_bgr8(hosts[2].camera, ptamd.make_params(333, 77))
    assert f"Maximum BVH tests: {mx}" in out.stdout
    assert np.array_equal(bmp_rows(tmp_path / "output.bmp", 333, 77), rows)  # tile split + integer sum: exact
    out = run_cli(tmp_path, "-p", "2", "-w", "96", "-h", "64", "-s", "7", "-b", "4", "-g", "2", PT_MULTI_SAME_DEVICE="1")
    assert out.returncode == 0, out.stderr
    second = ptamd.DeviceScene(hosts[2].desc_ptr)
    try:
        img, _ = ptamd.render_multi([devs[2], second], hosts[2].camera,
                                    ptamd.make_params(96, 64, spp=7, max_bounces=4, seed_mode=ptamd.SEED_STREAM))
    finally:
        second.close()
    assert np.array_equal(bmp_rows(tmp_path / "output.bmp", 96, 64), ptamd.quantise_rows(img))


def test_cli_has_no_cpu_path(tmp_path):
    out = run_cli(tmp_path, "-d", "1", "-p", "2")
    assert out.returncode != 0 and "no CPU fallback" in out.stderr


def test_render_multi_equals_its_parts_and_the_oracle(gpu_scenes):
    torch = pytest.importorskip("torch")
    hosts, devs = gpu_scenes
    W, H, SPP, B, N = 112, 80, 11, 4, 3
    extra = [ptamd.DeviceScene(hosts[2].desc_ptr) for _ in range(N - 1)]
    try:
        frame = ptamd.make_params(W, H, spp=SPP, max_bounces=B, seed_mode=ptamd.SEED_STREAM, sub_streams=2)
        img, st = ptamd.render_multi([devs[2]] + extra, hosts[2].camera, frame)
        assert st.render_s > 0
        # the same parts one by one: share i of the samples, stream set i, summed in part order, resolved on the device
        acc = np.zeros((H, W, 4), dtype=np.float32)
        cpu = np.zeros((H, W, 4), dtype=np.float32)
        buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        shares = [SPP // N + (1 if i < SPP % N else 0) for i in range(N)]
        for i, n in enumerate(shares):
            p = ptamd.make_params(W, H, spp=n, max_bounces=B, seed_mode=ptamd.SEED_STREAM, part_index=i, part_count=N,
                                  sub_streams=2, frame_spp=SPP)
            devs[2].accumulate_device(hosts[2].camera, p, buf.data_ptr(), None)
            torch.cuda.synchronize()
            acc += buf.cpu().numpy()
            for k in range(min(2, n)):  # the oracle has one stream per call: sub-stream k of part i is stream 2 i + k
                nk = n // min(2, n) + (1 if k < n % min(2, n) else 0)
                pk = ptamd.make_params(W, H, spp=nk, max_bounces=B, seed_mode=ptamd.SEED_STREAM, part_index=2 * i + k,
                                       part_count=2 * N, frame_spp=SPP)
                a, _ = ob.accumulate(hosts[2], pk, seed_mode=ptamd.SEED_STREAM)
                cpu += a
        assert np.all(acc[..., 3] == SPP)
        buf.copy_(torch.from_numpy(acc))
        out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        devs[2].resolve_device(buf.data_ptr(), W, H, out.data_ptr(), None)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), img)
        ref = ob.resolve(cpu)
        assert float(np.sqrt(np.mean((img - ref) ** 2))) < 2e-3 and (np.abs(img - ref).max(axis=2) < 2e-3).mean() >= 0.99
        # one handle: the plain call
        one, _ = ptamd.render_multi([devs[2]], hosts[2].camera, ptamd.make_params(W, H, spp=SPP, max_bounces=B))
        direct, _ = devs[2].render(hosts[2].camera, ptamd.make_params(W, H, spp=SPP, max_bounces=B))
        assert np.array_equal(one, direct)
        # more parts than samples: the empty shares contribute zeros
        few, _ = ptamd.render_multi([devs[2]] + extra, hosts[2].camera,
                                    ptamd.make_params(W, H, spp=2, max_bounces=B, seed_mode=ptamd.SEED_STREAM))
        assert np.isfinite(few).all() and few.std() > 0
        # heat map: tile partition over three handles, exact
        c, _ = ptamd.render_counts_multi([devs[2]] + extra, hosts[2].camera, ptamd.make_params(333, 77))
        full, _ = ob.render_counts(hosts[2], ptamd.make_params(333, 77))
        assert np.array_equal(c, full)
        with pytest.raises(ptamd.BackendError):  # the reference's seeding has one stream set: one GPU only
            ptamd.render_multi([devs[2]] + extra, hosts[2].camera, ptamd.make_params(W, H, spp=4, max_bounces=B))
    finally:
        for d in extra:
            d.close()
