"""tests/shared_libm/shared_libm.h on the CPU: the five functions that the radiance parity test compiles into both of its
builds are accurate to float rounding over the arguments the path feeds them (so the test builds render the same pictures),
and the shipped oracle does not contain them."""
import ctypes as C
import os

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "oracle", "libpt_oracle_sharedlibm.so")


@pytest.fixture(scope="module")
def probe():
    if not os.path.exists(LIB):
        pytest.skip("oracle/libpt_oracle_sharedlibm.so not built (make oracle)")
    lib = C.CDLL(LIB)
    lib.oracle_shared_libm_probe.argtypes = [C.c_int, C.c_float, C.c_float]
    lib.oracle_shared_libm_probe.restype = C.c_float

    def f(fn, x, y=0.0):
        return np.array([lib.oracle_shared_libm_probe(fn, float(a), float(b)) for a, b in np.broadcast(x, y)], dtype=np.float32)
    return f


def ulps(got, want64):
    want = want64.astype(np.float32)
    spacing = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / np.maximum(spacing, 1e-45)


def test_trigonometric_functions(probe):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(0.0, 2.0 * np.pi, 4000), [0.0, np.pi / 2, np.pi, 1.5 * np.pi, 2.0 * np.pi, 6.2831855]]).astype(np.float32)
    assert ulps(probe(0, x), np.cos(x.astype(np.float64))).max() <= 0.5001      # correctly rounded up to the tie
    assert ulps(probe(1, x), np.sin(x.astype(np.float64))).max() <= 0.5001
    # directions on the unit sphere, as the environment lookup passes them
    d = rng.normal(size=(4000, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    assert ulps(probe(2, d[:, 2], d[:, 0]), np.arctan2(d[:, 2].astype(np.float64), d[:, 0].astype(np.float64))).max() <= 0.5001
    assert ulps(probe(3, d[:, 1]), np.arccos(d[:, 1].astype(np.float64))).max() <= 0.5001
    assert probe(3, np.float32(1.0000001))[0] == 0.0 and abs(probe(3, np.float32(-1.0000001))[0] - np.float32(np.pi)) < 1e-6
    assert probe(2, 0.0, 0.0)[0] == 0.0
    for y, x_, want in ((1.0, 0.0, np.pi / 2), (-1.0, 0.0, -np.pi / 2), (0.0, -1.0, np.pi), (0.0, 1.0, 0.0)):
        assert abs(probe(2, y, x_)[0] - want) < 1e-6


def test_power_function(probe):
    rng = np.random.default_rng(2)
    base = np.concatenate([rng.uniform(0.0, 1.0, 4000), [0.0, 1.0, 1e-5, 0.99999, 1e-30]]).astype(np.float32)
    assert ulps(probe(4, base, 5.0), base.astype(np.float64) ** 5.0).max() <= 0.5001                          # Schlick's (1 - VdotH)^5
    g = np.float32(1.0 / 2.2)
    assert ulps(probe(4, base[base > 0], g), base[base > 0].astype(np.float64) ** float(g)).max() <= 0.5001     # gamma
    assert probe(4, 0.0, g)[0] == 0.0 and probe(4, 0.0, 5.0)[0] == 0.0
    assert probe(4, np.float32(-1e-7), 5.0)[0] < 0.0 and abs(probe(4, np.float32(-0.5), 5.0)[0] + 0.03125) < 1e-9   # odd integral exponent
    assert np.isnan(probe(4, np.float32(-0.5), g)[0])


def test_the_shipped_checker_does_not_contain_them():
    lib = C.CDLL(os.path.join(REPO, "oracle", "libpt_oracle.so"))
    assert not hasattr(lib, "oracle_shared_libm_probe")
    # and neither does the shipped backend's source path: the hook is a preprocessor switch that the default build leaves off
    mk = open(os.path.join(REPO, "Makefile")).read()
    default_rule = mk[mk.index("$(LIB)/libptx_backend.so:"):mk.index("$(LIB)/libptx_backend_sharedlibm.so:")]
    assert "PTX_SHARED_LIBM" not in default_rule
