"""One process, two HIP runtimes: libptx_backend.so links the system's libamdhip64, the PyTorch wheel bundles its own.
Loaded torch-first (tests/conftest.py, bench.py) both use torch's; loaded backend-first, torch maps a second runtime and
then reports "No HIP GPUs are available".  ptamd detects the second mapping and says what to do (ptamd.check_hip_runtime)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY = os.path.join(REPO, "path-tracer_amd", "py")


def run(code):
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                          env=dict(os.environ, PYTHONPATH=PY + os.pathsep + os.environ.get("PYTHONPATH", "")))


def test_backend_first_then_torch_is_reported():
    r = run("import ptamd\n"
            "ptamd.backend_lib()\n"
            "print('runtimes before torch:', len(ptamd.hip_runtimes()))\n"
            "import torch\n"
            "n = len(ptamd.hip_runtimes())\n"
            "print('runtimes after torch:', n)\n"
            "try:\n"
            "    ptamd.backend_lib()\n"
            "    print('NO ERROR')\n"
            "except ptamd.BackendError as e:\n"
            "    print('ERROR:', e)\n")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "runtimes before torch: 1" in r.stdout
    if "runtimes after torch: 1" in r.stdout:   # a torch build that uses the system's runtime: nothing to report
        assert "NO ERROR" in r.stdout
    else:
        assert "ERROR: two HIP runtimes are mapped" in r.stdout and "Import torch BEFORE" in r.stdout, r.stdout


def test_torch_first_is_fine():
    r = run("import torch\n"
            "import ptamd\n"
            "ptamd.backend_lib()\n"
            "ptamd.check_hip_runtime()\n"
            "print('runtimes:', len(ptamd.hip_runtimes()))\n")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "runtimes: 1" in r.stdout, r.stdout
