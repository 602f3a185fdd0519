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
