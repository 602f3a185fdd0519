"""The C-ABI libraries load and export every symbol the headers declare (no compute calls: CPU-safe)."""
import ctypes as C
import os
import re

import pytest

import ptamd

REPO = ptamd.REPO_DIR


def declared_functions(header, prefix):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, text)))


def test_host_library_exports_header_symbols():
    lib = ptamd.host_lib()
    names = declared_functions("pt_host.h", "pth_")
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n


def test_backend_library_exports_header_symbols():
    path = os.path.join(ptamd.LIB_DIR, "libptx_backend.so")
    if not os.path.exists(path):
        pytest.skip("libptx_backend.so not built yet (make backend)")
    lib = ptamd.backend_lib()
    names = declared_functions("ptx_backend.h", "ptx_")
    assert set(names) == set(ptamd.BACKEND_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n
    assert b"gfx950" in lib.ptx_version()


def test_struct_sizes_match_the_reference_layouts():
    # dataTypes.hpp: BVHNode 32 B, GPUTriangle 36 B (packed vec3); mesh.hpp GPUMesh = 2 mat4 + 2 ids here
    assert C.sizeof(ptamd.BvhNode) == 32
    assert C.sizeof(ptamd.Triangle) == 36
    assert C.sizeof(ptamd.Instance) == 136
    assert C.sizeof(ptamd.Geometry) == 32
    assert C.sizeof(ptamd.Params) == 36


def test_backend_rejects_bad_arguments_without_a_gpu():
    path = os.path.join(ptamd.LIB_DIR, "libptx_backend.so")
    if not os.path.exists(path):
        pytest.skip("libptx_backend.so not built yet")
    lib = ptamd.backend_lib()
    h = C.c_void_p()
    assert lib.ptx_create(None, 0, C.byref(h)) == -1          # PTX_EINVAL before any HIP call
    assert b"null" in lib.ptx_last_error()
    empty = ptamd.SceneDesc()
    assert lib.ptx_create(C.byref(empty), 0, C.byref(h)) == -1


def test_product_code_does_not_reference_the_oracle():
    """oracle/ is test infrastructure: nothing under path-tracer_amd/ or include/ may use it."""
    for root in (os.path.join(REPO, "path-tracer_amd"), os.path.join(REPO, "include")):
        for dirpath, _, files in os.walk(root):
            for fn in files:
                if fn.endswith((".so", ".o", ".pyc")) or fn == "PathTracer":
                    continue
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "pt_oracle" not in text and "oracle_binding" not in text and "libpt_oracle" not in text, os.path.join(dirpath, fn)
