"""The host pipeline and the oracle under AddressSanitizer + UndefinedBehaviourSanitizer (host code only: the GPU side
cannot be sanitised on this pool).  tests/native/host_sanitize.cpp is linked with the sources of libpt_host and with the
oracle and walks the three presets, the flattened-scene cache (miss, hit, truncated file), the STL ingest with
degenerate facets, the output stage and both render modes of the oracle; any report makes it exit non-zero."""
import os
import shutil
import subprocess

import pytest

import ptamd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = ["scene.cpp", "loaders.cpp", "sample_scenes.cpp", "flatten.cpp", "output.cpp", "capi.cpp"]
FLAGS = ["-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
         "-I" + os.path.join(REPO, "include")]


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("gcc") is None, reason="needs gcc / g++")
def test_host_pipeline_and_oracle_under_asan_and_ubsan(tmp_path):
    obj = str(tmp_path / "oracle.o")
    exe = str(tmp_path / "host_sanitize")
    subprocess.run(["gcc", *FLAGS, "-std=c11", "-c", os.path.join(REPO, "oracle", "pt_oracle.c"), "-o", obj], check=True, timeout=300)
    subprocess.run(["g++", *FLAGS, "-std=c++17", "-o", exe, os.path.join(REPO, "tests", "native", "host_sanitize.cpp"),
                    *[os.path.join(REPO, "path-tracer_amd", "host", f) for f in HOST], obj, "-lz", "-lm", "-lpthread"], check=True, timeout=600)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", PT_BUILD_THREADS="4")
    out = subprocess.run([exe, str(tmp_path / "scratch"), ptamd.MODELS_DIR], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "host_sanitize: ok" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]


HIPCC = "/opt/rocm/bin/hipcc"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(CLANG)), reason="needs hipcc and its clang++")
def test_backend_host_side_under_asan_and_ubsan(tmp_path):
    """The host half of ptx_backend.hip -- validation of the description, the four-wide re-layout, the tight TLAS
    refit -- with the sanitizers on the host side only (-Xarch_host); no GPU is touched."""
    san = ["-fsanitize=address", "-fsanitize=undefined", "-fno-sanitize-recover=undefined"]
    inc = "-I" + os.path.join(REPO, "include")
    jobs, objs = [], []
    for f in HOST + ["../../tests/native/backend_host_sanitize.cpp"]:
        o = str(tmp_path / (os.path.basename(f) + ".o"))
        objs.append(o)
        jobs.append(subprocess.Popen([CLANG, "-O1", "-g", "-std=c++17", "-ffp-contract=off", *san, "-fno-omit-frame-pointer", inc, "-c",
                                      os.path.normpath(os.path.join(REPO, "path-tracer_amd", "host", f)), "-o", o]))
    o = str(tmp_path / "backend.o")
    objs.append(o)
    xh = [x for s in san for x in ("-Xarch_host", s)]
    jobs.append(subprocess.Popen([HIPCC, "-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                                  "-fno-slp-vectorize", "-fhip-fp32-correctly-rounded-divide-sqrt", inc, *xh, "-c",
                                  os.path.join(REPO, "path-tracer_amd", "csrc", "ptx_backend.hip"), "-o", o]))
    for j in jobs:
        assert j.wait(timeout=900) == 0
    exe = str(tmp_path / "backend_host_sanitize")
    subprocess.run([HIPCC, *san, "-fno-gpu-sanitize", "-o", exe, *objs, "-lz", "-lpthread"], check=True, timeout=300)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1", PT_BUILD_THREADS="4")
    out = subprocess.run([exe, ptamd.MODELS_DIR], env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0 and "backend_host_sanitize: ok" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
