"""bench.py end to end on the GPU at a reduced sample count: one JSON line with the fields the driver reads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    env = dict(os.environ, PTX_BENCH_CPU_BUDGET="1")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "1", "--spp", "4"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["metric"] == "Mrays/s at 1500x800, 100spp, 6 bounces, Scene 1" and j["unit"] == "Mrays/s"
    assert j["n_gpus"] == 1 and j["steps"] == 1 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["scaling"] == "strong" and j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert j["value"] > 0 and j["ms_per_step"] > 0
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # (d) the fraction is counter traffic / time / peak, and says so; SURVEY 8(d)'s algorithmic bytes are l1_level
    assert r["basis"] == "counter" and r["l1_level"]["algorithmic_bytes"] > 0 and r["memory_path"]["frac"] > 0
    assert r["kernel"] == "ptx::poolKernel<false>" and r["kernel_ms"] > 0
    # (a) counter figures are reported only next to the build they were measured on (profiles/traffic.json: build_id, kernel,
    # stream count); this run is a 4-spp frame with two streams, so they must be withheld, with the reason
    import ptamd
    assert r["build_id"] == ptamd.build_id() and r["build_id"] != "unstamped"
    assert r["from_committed_profile"] is False and r["frac"] is None and r["achieved"] is None and r["traffic"] is None and r["valu"] is None
    assert "withheld" in r["counters_note"] or "measured on build" in r["counters_note"]
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mrays/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert c["cores_available"] == os.cpu_count() and c["cores"] <= c["cores_available"]  # (c)
    # value = rays of the timed steps / wall time
    assert abs(j["value"] - j["config"]["rays_per_step"] / (j["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * j["value"]


def test_bench_line_of_the_headline_frame_carries_the_counter_figures_of_its_own_build():
    """The default run (100 spp, 100 streams).  When profiles/traffic.json was measured on the library that is loaded, the line
    carries the counter-derived figures, consistent with each other; when the sources have changed since, it withholds them."""
    import ptamd
    env = dict(os.environ, PTX_BENCH_CPU_BUDGET="1")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                          "--no-reference-seeding"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])
    r = j["roofline"]
    t = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
    assert j["config"]["rng_streams_per_pixel_per_gpu"] == 100
    if t.get("build_id") == ptamd.build_id():
        assert r["from_committed_profile"] is True and t["build_id"] in r["counters_from"]
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] <= 1.0
        assert r["traffic"] > 0 and abs(r["achieved"] - r["traffic"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
        assert r["l1_level"]["algorithmic_bytes"] > r["traffic"]
        v = r["valu"]  # (b) both peaks, both fractions
        assert 0.0 < v["frac_guide"] < v["frac"] <= 1.0 and abs(v["frac"] * v["peak"] - v["frac_guide"] * v["peak_guide"]) < 1e-6 * v["achieved"]
        assert abs(v["peak_guide"] / v["peak"] - 1.25) < 1e-9
        assert 0.0 < v["lane_utilisation"] <= 1.0 and abs(v["insts"] - v["per_ray"] * j["config"]["rays_per_step"]) < 1.0
    else:
        assert r["from_committed_profile"] is False and r["frac"] is None and r["valu"] is None and "measured on build" in r["counters_note"]


def test_bench_two_ranks_rehearsed_on_one_gpu():
    """The N > 1 path of bench.py as the driver launches it (torch.distributed.run, one rank per GPU), rehearsed with
    both ranks on this box's one GPU and gloo in place of RCCL: the samples of the frame are split over the ranks
    (strong scaling), each renders its share with its own RNG streams, the accumulation buffers are summed on rank 0,
    rank 0 prints the one line, whose rays are those of the whole frame."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PTX_BENCH_CPU_BUDGET="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--spp", "6",
           "--rehearse-on-one-gpu"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] > 0
    # 6 spp over two ranks: 1500 * 800 * 6 paths in all, between one and six rays each
    paths = 1500 * 800 * 6
    assert paths <= j["config"]["rays_per_step"] <= 6 * paths
    assert abs(j["value"] - j["config"]["rays_per_step"] / (j["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * j["value"]
