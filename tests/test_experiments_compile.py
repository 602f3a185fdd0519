"""The measured-and-rejected kernels under experiments/ are not part of the shipped library; this keeps them compiling
(device code for gfx950; hipcc cross-compiles here, no GPU needed), so that the A/B records under profiles/ stay
reproducible (experiments/README.md)."""
import os
import subprocess
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O1", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-I" + os.path.join(REPO, "include"), "--cuda-device-only", "-c"]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_experimental_kernels_still_compile():
    # one compilation with all three switched on: the shipped sources plus the three experiment headers
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-DPTX_EXPERIMENT_PHASES", "-DPTX_EXPERIMENT_EXCHANGE", "-DPTX_EXPERIMENT_QUEUES",
                            "-DPTX_QUEUE_DEBUG", os.path.join(REPO, "path-tracer_amd", "csrc", "ptx_backend.hip"),
                            "-o", os.path.join(tmp, "experiments.o")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]


def test_default_build_does_not_include_them():
    src = open(os.path.join(REPO, "path-tracer_amd", "csrc", "ptx_backend.hip")).read()
    for header, guard in (("radiance_kernel.hpp", "PTX_EXPERIMENT_PHASES"), ("exchange_kernel.hpp", "PTX_EXPERIMENT_EXCHANGE"),
                          ("queue_kernel.hpp", "PTX_EXPERIMENT_QUEUES")):
        at = src.index(header)
        assert guard in src[max(0, at - 200):at], header
    assert not any(f.endswith("_kernel.hpp") and f not in ("pool_kernel.hpp",) for f in os.listdir(os.path.join(REPO, "path-tracer_amd", "csrc")))
