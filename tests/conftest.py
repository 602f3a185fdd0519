import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd", "py"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_if_missing():
    """Libraries are built in-tree by `make` / __graft_entry__.build(); build the CPU ones on demand."""
    import subprocess
    need_host = not os.path.exists(os.path.join(REPO, "path-tracer_amd", "lib", "libpt_host.so"))
    need_oracle = not os.path.exists(os.path.join(REPO, "oracle", "libpt_oracle.so"))
    if need_host:
        subprocess.check_call(["make", "-C", REPO, "host"])
    if need_oracle:
        subprocess.check_call(["make", "-C", REPO, "oracle"])


_build_if_missing()

import ptamd  # noqa: E402


@pytest.fixture(scope="session")
def scene2():
    s = ptamd.HostScene(2)
    yield s
    s.close()


@pytest.fixture(scope="session")
def scene0():
    s = ptamd.HostScene(0)
    yield s
    s.close()


@pytest.fixture(scope="session")
def scene1():
    s = ptamd.HostScene(1)
    yield s
    s.close()


@pytest.fixture(scope="session")
def scene1_small():
    """Scene 1 with 300 instances (same RNG draw order, shorter TLAS) for CPU-side tests."""
    s = ptamd.HostScene(1, scatter_instances=300)
    yield s
    s.close()


@pytest.fixture(scope="session")
def gpu_scenes(scene0, scene1, scene2):
    """ptx_create on cuda:0 for the three presets (GPU tests only)."""
    # torch ships its own copy of the HIP runtime: when a test uses torch for device buffers it has to
    # be loaded BEFORE libptx_backend.so pulls in /opt/rocm's, or torch finds no GPU ("No HIP GPUs")
    try:
        import torch  # noqa: F401
        torch.cuda.is_available()
    except ImportError:
        pass
    devs = {0: ptamd.DeviceScene(scene0.desc_ptr), 1: ptamd.DeviceScene(scene1.desc_ptr),
            2: ptamd.DeviceScene(scene2.desc_ptr)}
    hosts = {0: scene0, 1: scene1, 2: scene2}
    yield hosts, devs
    for d in devs.values():
        d.close()
