"""ctypes bindings of the two product libraries, for the Python tests and bench.py.

* libpt_host.so      -- host scene pipeline (include/pt_host.h): presets, BVH build, flatten.
* libptx_backend.so  -- the HIP backend behind the C ABI (include/ptx_backend.h).

Python is plumbing here, not the product: every render call goes through the C ABI exactly as the
C++ host program (path-tracer_amd/host/gpu_render.cpp) does.  There is no CPU fallback: if the
backend library is missing, or no GPU is visible, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.dirname(_HERE)
REPO_DIR = os.path.dirname(PKG_DIR)
LIB_DIR = os.path.join(PKG_DIR, "lib")
GOLDEN_DIR = os.path.join(REPO_DIR, "tests", "golden")
MODELS_DIR = os.environ.get("PT_MODELS_DIR", os.path.join(GOLDEN_DIR, "models"))
ENV_PATH = os.environ.get(
    "PT_ENV_PATH", os.path.join(GOLDEN_DIR, "environment", "evening_road_01_puresky_2k.hdr"))

PTX_NONE = 0xFFFFFFFF
SEED_REFERENCE_GPU = 0
SEED_STREAM = 1


class BvhNode(C.Structure):
    _fields_ = [("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3),
                ("left_first", C.c_uint32), ("count", C.c_uint32)]


class Triangle(C.Structure):
    _fields_ = [("v0", C.c_float * 3), ("v1", C.c_float * 3), ("v2", C.c_float * 3)]


class Geometry(C.Structure):
    _fields_ = [("node_offset", C.c_uint32), ("node_count", C.c_uint32), ("tri_offset", C.c_uint32),
                ("tri_count", C.c_uint32), ("face_normal_offset", C.c_uint32),
                ("vtx_normal_offset", C.c_uint32), ("vtx_uv_offset", C.c_uint32), ("reserved", C.c_uint32)]


class Instance(C.Structure):
    _fields_ = [("inv_model", C.c_float * 16), ("normal_matrix", C.c_float * 16),
                ("geometry_id", C.c_uint32), ("material_id", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("metalness", C.c_float), ("roughness", C.c_float),
                ("emissive", C.c_float * 3), ("f0", C.c_float * 3),
                ("albedo_texture", C.c_uint32), ("emissive_texture", C.c_uint32)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("texel_offset", C.c_uint64)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("tlas_nodes", C.POINTER(BvhNode)), ("tlas_node_count", C.c_uint32),
        ("instances", C.POINTER(Instance)), ("instance_count", C.c_uint32),
        ("geometries", C.POINTER(Geometry)), ("geometry_count", C.c_uint32),
        ("blas_nodes", C.POINTER(BvhNode)), ("blas_node_total", C.c_uint32),
        ("triangles", C.POINTER(Triangle)), ("triangle_total", C.c_uint32),
        ("face_normals", C.POINTER(C.c_float)), ("face_normal_total", C.c_uint32),
        ("vtx_normals", C.POINTER(C.c_float)), ("vtx_normal_total", C.c_uint32),
        ("vtx_uvs", C.POINTER(C.c_float)), ("vtx_uv_total", C.c_uint32),
        ("materials", C.POINTER(Material)), ("material_count", C.c_uint32),
        ("textures", C.POINTER(Texture)), ("texture_count", C.c_uint32),
        ("texels", C.POINTER(C.c_float)), ("texel_total", C.c_uint64),
        ("environment", C.POINTER(C.c_float)), ("env_width", C.c_uint32), ("env_height", C.c_uint32),
    ]


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("target", C.c_float * 3), ("up", C.c_float * 3),
                ("fov_deg", C.c_float)]


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32),
                ("max_bounces", C.c_uint32), ("seed_mode", C.c_uint32), ("part_index", C.c_uint32),
                ("part_count", C.c_uint32), ("sub_streams", C.c_uint32), ("frame_spp", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("upload_s", C.c_double), ("render_s", C.c_double), ("rays", C.c_uint64),
                ("node_steps", C.c_uint64), ("tri_tests", C.c_uint64), ("inst_entries", C.c_uint64),
                ("hits", C.c_uint64), ("misses", C.c_uint64), ("box_hits", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class HostInfo(C.Structure):
    _fields_ = [("triangles_per_mesh", C.c_uint32), ("blas_nodes", C.c_uint32), ("tlas_nodes", C.c_uint32),
                ("instances", C.c_uint32), ("blas_depth", C.c_uint32), ("tlas_depth", C.c_uint32),
                ("blas_max_leaf", C.c_uint32), ("tlas_max_leaf", C.c_uint32),
                ("stand_in_geometry", C.c_uint32), ("stand_in_environment", C.c_uint32),
                ("blas_build_s", C.c_double), ("tlas_build_s", C.c_double)]


def make_params(width, height, spp=1, max_bounces=1, seed_mode=SEED_REFERENCE_GPU, part_index=0, part_count=1,
                sub_streams=0, frame_spp=0):
    return Params(width, height, spp, max_bounces, seed_mode, part_index, part_count, sub_streams, frame_spp)


# ------------------------------------------------------------------------------------------------
# host pipeline
# ------------------------------------------------------------------------------------------------
_host = None


def host_lib():
    global _host
    if _host is None:
        path = os.path.join(LIB_DIR, "libpt_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make host` (or __graft_entry__.build()) first")
        lib = C.CDLL(path)
        lib.pth_scene_create.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.POINTER(C.c_void_p)]
        lib.pth_scene_create.restype = C.c_int
        lib.pth_scene_create_cached.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p,
                                                C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        lib.pth_scene_create_cached.restype = C.c_int
        lib.pth_write_standin_stl.argtypes = [C.c_char_p, C.c_uint32]
        lib.pth_write_standin_stl.restype = C.c_long
        lib.pth_scene_desc.argtypes = [C.c_void_p]
        lib.pth_scene_desc.restype = C.POINTER(SceneDesc)
        lib.pth_scene_camera.argtypes = [C.c_void_p, C.POINTER(Camera)]
        lib.pth_scene_info.argtypes = [C.c_void_p, C.POINTER(HostInfo)]
        lib.pth_scene_destroy.argtypes = [C.c_void_p]
        lib.pth_heatmap_colour.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.pth_heatmap_colour.restype = C.c_uint32
        lib.pth_write_bmp.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.pth_write_bmp.restype = C.c_int
        lib.pth_bmp_row_bytes.argtypes = [C.c_uint32]
        lib.pth_bmp_row_bytes.restype = C.c_uint32
        lib.pth_quantise_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.pth_quantise_rows.restype = None
        lib.pth_write_bmp_rows.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.pth_write_bmp_rows.restype = C.c_int
        lib.pth_pcg_hash.argtypes = [C.POINTER(C.c_uint32)]
        lib.pth_pcg_hash.restype = C.c_uint32
        lib.pth_random_float.argtypes = [C.POINTER(C.c_uint32)]
        lib.pth_random_float.restype = C.c_float
        lib.pth_parse_decimal.argtypes = [C.c_char_p, C.POINTER(C.c_double)]
        lib.pth_parse_decimal.restype = C.c_int
        lib.pth_mat4_inverse.argtypes = [C.c_void_p, C.c_void_p]
        lib.pth_mesh_matrices.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pth_last_error.restype = C.c_char_p
        _host = lib
    return _host


class HostScene:
    """A preset scene built by the host pipeline (getScene + completeScene + flatten)."""

    def __init__(self, preset, scatter_instances=0, sky=(0, 0), models_dir=None, env_path=None, cache_dir=None):
        lib = host_lib()
        self._h = C.c_void_p()
        self.cache_hit = False
        if cache_dir:  # flattened-scene cache (pt_host.h: pth_scene_create_cached)
            hit = C.c_int(0)
            rc = lib.pth_scene_create_cached(preset, (models_dir or MODELS_DIR).encode(), (env_path or ENV_PATH).encode(),
                                             scatter_instances, sky[0], sky[1], str(cache_dir).encode(), C.byref(hit),
                                             C.byref(self._h))
            self.cache_hit = bool(hit.value)
        else:
            rc = lib.pth_scene_create(preset, (models_dir or MODELS_DIR).encode(), (env_path or ENV_PATH).encode(),
                                      scatter_instances, sky[0], sky[1], C.byref(self._h))
        if rc != 0:
            raise RuntimeError(lib.pth_last_error().decode())
        self.preset = preset
        self.desc_ptr = lib.pth_scene_desc(self._h)
        self.desc = self.desc_ptr.contents
        self.camera = Camera()
        lib.pth_scene_camera(self._h, C.byref(self.camera))
        self.info = HostInfo()
        lib.pth_scene_info(self._h, C.byref(self.info))

    def close(self):
        if self._h:
            host_lib().pth_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # numpy views of the flattened arrays (no copies)
    def _view(self, ptr, count, dtype, shape_tail=()):
        if count == 0 or not ptr:
            return np.zeros((0,) + shape_tail, dtype=dtype)
        n = int(np.prod(shape_tail)) if shape_tail else 1
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * n * np.dtype(dtype).itemsize,))
        return arr.view(dtype).reshape((count,) + shape_tail)

    @property
    def tlas_nodes(self):
        return self._view(self.desc.tlas_nodes, self.desc.tlas_node_count, np.uint32, (8,))

    @property
    def blas_nodes(self):
        return self._view(self.desc.blas_nodes, self.desc.blas_node_total, np.uint32, (8,))

    @property
    def triangles(self):
        return self._view(self.desc.triangles, self.desc.triangle_total, np.float32, (9,))

    @property
    def instances(self):
        return self._view(self.desc.instances, self.desc.instance_count, np.float32, (34,))


def heatmap_colour(counts):
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    rgb = np.empty(counts.shape + (3,), dtype=np.float32)
    mx = host_lib().pth_heatmap_colour(counts.ctypes.data, counts.size, rgb.ctypes.data)
    return rgb, mx


def quantise_rows(rgb):
    """Host output stage: (H, W, 3) float image -> BMP pixel array (H, row_bytes) uint8."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w = rgb.shape[:2]
    rows = np.empty((h, host_lib().pth_bmp_row_bytes(w)), dtype=np.uint8)
    host_lib().pth_quantise_rows(rgb.ctypes.data, w, h, rows.ctypes.data)
    return rows


def write_bmp(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w = rgb.shape[:2]
    if host_lib().pth_write_bmp(path.encode(), rgb.ctypes.data, w, h) != 0:
        raise RuntimeError(f"cannot write {path}")


# ------------------------------------------------------------------------------------------------
# HIP backend
# ------------------------------------------------------------------------------------------------
_backend = None


class BackendError(RuntimeError):
    pass

BACKEND_SYMBOLS = [
    "ptx_device_count", "ptx_create", "ptx_render_counts", "ptx_render", "ptx_accumulate_device",
    "ptx_resolve_device", "ptx_counts_device", "ptx_measure", "ptx_last_kernel_seconds", "ptx_destroy",
    "ptx_last_error", "ptx_version", "ptx_check_wide_layout", "ptx_check_tight_tlas", "ptx_selftest_reciprocal", "ptx_render_counts_bgr8",
    "ptx_render_bgr8", "ptx_accumulate_device_noalloc", "ptx_render_multi", "ptx_render_counts_multi",
    "ptx_build_id", "ptx_noalloc_scratch_bytes",
]


def noalloc_scratch_bytes():
    """PTX_NOALLOC_SCRATCH_BYTES of the loaded library (asked of the library: no copy of the constant here)."""
    return int(backend_lib().ptx_noalloc_scratch_bytes())


def build_id():
    """ptx_build_id() of the loaded library: hash of its sources and flags (tools/source_hash.py)."""
    return backend_lib().ptx_build_id().decode()


def hip_runtimes():
    """Distinct libamdhip64 files mapped into this process (from /proc/self/maps)."""
    found = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1] if "/" in line else ""
                if "libamdhip64" in os.path.basename(path) and os.path.realpath(path) not in found:
                    found.append(os.path.realpath(path))
    except OSError:
        pass
    return found


def check_hip_runtime():
    """libptx_backend.so and PyTorch must share ONE HIP runtime.  The backend links the system's libamdhip64; the PyTorch
    wheel bundles its own.  Whichever is loaded first serves both when the other is loaded later by the same soname -- that is
    what `import torch` BEFORE ptamd.backend_lib() arranges (tests/conftest.py, bench.py, multigpu.HipBackend).  Loaded the
    other way round, the process ends up with two runtimes: torch then reports "No HIP GPUs are available" and device
    pointers of one runtime mean nothing to the other.  This says so instead."""
    libs = hip_runtimes()
    if len(libs) > 1:
        raise BackendError("two HIP runtimes are mapped into this process (" + ", ".join(libs) + "): libptx_backend.so was loaded "
                           "before PyTorch.  Import torch BEFORE the first ptamd.backend_lib() / DeviceScene in a process that "
                           "uses both, so that the backend binds to the runtime torch has loaded (INTEGRATION.md section 5).")


def backend_lib():
    global _backend
    if _backend is not None:
        check_hip_runtime()
    if _backend is None:
        # PTX_BACKEND_LIB selects an experimental build of the same library (tools/build_variants.sh)
        path = os.environ.get("PTX_BACKEND_LIB", os.path.join(LIB_DIR, "libptx_backend.so"))
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make backend` (or __graft_entry__.build()); "
                               "there is no CPU fallback for the render path")
        lib = C.CDLL(path)
        vp = C.c_void_p
        lib.ptx_device_count.argtypes = [C.POINTER(C.c_int)]
        lib.ptx_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(vp)]
        lib.ptx_render_counts.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), vp, C.POINTER(Stats)]
        lib.ptx_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), vp, C.POINTER(Stats)]
        lib.ptx_accumulate_device.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), vp, vp]
        lib.ptx_accumulate_device_noalloc.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), vp, vp]
        lib.ptx_render_multi.argtypes = [C.POINTER(vp), C.c_uint32, C.POINTER(Camera), C.POINTER(Params), vp, C.POINTER(Stats)]
        lib.ptx_render_counts_multi.argtypes = [C.POINTER(vp), C.c_uint32, C.POINTER(Camera), C.POINTER(Params), vp, C.POINTER(Stats)]
        lib.ptx_resolve_device.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, vp]
        lib.ptx_counts_device.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), vp, vp]
        lib.ptx_measure.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), C.c_int, C.POINTER(Stats)]
        lib.ptx_last_kernel_seconds.argtypes = [vp, C.POINTER(C.c_double)]
        lib.ptx_render_counts_bgr8.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), vp, C.c_uint32, C.POINTER(C.c_uint32),
                                               C.POINTER(Stats)]
        lib.ptx_render_bgr8.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), vp, C.c_uint32, C.POINTER(Stats)]
        lib.ptx_destroy.argtypes = [vp]
        lib.ptx_destroy.restype = None
        lib.ptx_last_error.restype = C.c_char_p
        lib.ptx_version.restype = C.c_char_p
        lib.ptx_build_id.restype = C.c_char_p
        lib.ptx_noalloc_scratch_bytes.restype = C.c_uint64
        for name in BACKEND_SYMBOLS:
            getattr(lib, name)
        _backend = lib
        check_hip_runtime()
    return _backend


def _check(rc, what):
    if rc != 0:
        raise BackendError(f"{what} failed ({rc}): {backend_lib().ptx_last_error().decode()}")


class DeviceScene:
    """ptx_create / ptx_destroy around a flattened scene."""

    def __init__(self, desc_ptr, device=0):
        lib = backend_lib()
        self._h = C.c_void_p()
        _check(lib.ptx_create(desc_ptr, device, C.byref(self._h)), "ptx_create")
        self.device = device

    def close(self):
        if self._h:
            backend_lib().ptx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render_counts(self, camera, params):
        out = np.empty((params.height, params.width), dtype=np.uint32)
        st = Stats()
        _check(backend_lib().ptx_render_counts(self._h, C.byref(camera), C.byref(params), out.ctypes.data, C.byref(st)),
               "ptx_render_counts")
        return out, st

    def render(self, camera, params):
        out = np.empty((params.height, params.width, 3), dtype=np.float32)
        st = Stats()
        _check(backend_lib().ptx_render(self._h, C.byref(camera), C.byref(params), out.ctypes.data, C.byref(st)),
               "ptx_render")
        return out, st

    def render_counts_bgr8(self, camera, params):
        """Device-side output stage of the heat map: (rows (H, row_bytes) uint8, maximum count, stats)."""
        rows = np.empty((params.height, host_lib().pth_bmp_row_bytes(params.width)), dtype=np.uint8)
        st = Stats()
        mx = C.c_uint32()
        _check(backend_lib().ptx_render_counts_bgr8(self._h, C.byref(camera), C.byref(params), rows.ctypes.data_as(C.c_void_p),
                                                    rows.shape[1], C.byref(mx), C.byref(st)), "ptx_render_counts_bgr8")
        return rows, mx.value, st

    def render_bgr8(self, camera, params):
        rows = np.empty((params.height, host_lib().pth_bmp_row_bytes(params.width)), dtype=np.uint8)
        st = Stats()
        _check(backend_lib().ptx_render_bgr8(self._h, C.byref(camera), C.byref(params), rows.ctypes.data_as(C.c_void_p),
                                             rows.shape[1], C.byref(st)), "ptx_render_bgr8")
        return rows, st

    def accumulate_device(self, camera, params, d_accum_ptr, stream=None):
        _check(backend_lib().ptx_accumulate_device(self._h, C.byref(camera), C.byref(params), d_accum_ptr, stream),
               "ptx_accumulate_device")

    def accumulate_device_noalloc(self, camera, params, d_accum_ptr, stream=None):
        """ptx_accumulate_device on the library's static scratch array: no hipMalloc / hipFree in the call."""
        _check(backend_lib().ptx_accumulate_device_noalloc(self._h, C.byref(camera), C.byref(params), d_accum_ptr, stream),
               "ptx_accumulate_device_noalloc")

    def resolve_device(self, d_accum_ptr, width, height, d_out_ptr, stream=None):
        _check(backend_lib().ptx_resolve_device(self._h, d_accum_ptr, width, height, d_out_ptr, stream),
               "ptx_resolve_device")

    def counts_device(self, camera, params, d_counts_ptr, stream=None):
        _check(backend_lib().ptx_counts_device(self._h, C.byref(camera), C.byref(params), d_counts_ptr, stream),
               "ptx_counts_device")

    def measure(self, camera, params, heatmap=False):
        st = Stats()
        _check(backend_lib().ptx_measure(self._h, C.byref(camera), C.byref(params), int(heatmap), C.byref(st)),
               "ptx_measure")
        return st

    def last_kernel_seconds(self):
        s = C.c_double()
        _check(backend_lib().ptx_last_kernel_seconds(self._h, C.byref(s)), "ptx_last_kernel_seconds")
        return s.value


def _handles(dscenes):
    arr = (C.c_void_p * len(dscenes))(*[d._h for d in dscenes])
    return arr


def render_multi(dscenes, camera, params):
    """ptx_render_multi: one frame over the devices of `dscenes` (one DeviceScene per device)."""
    out = np.empty((params.height, params.width, 3), dtype=np.float32)
    st = Stats()
    _check(backend_lib().ptx_render_multi(_handles(dscenes), len(dscenes), C.byref(camera), C.byref(params), out.ctypes.data,
                                          C.byref(st)), "ptx_render_multi")
    return out, st


def render_counts_multi(dscenes, camera, params):
    out = np.empty((params.height, params.width), dtype=np.uint32)
    st = Stats()
    _check(backend_lib().ptx_render_counts_multi(_handles(dscenes), len(dscenes), C.byref(camera), C.byref(params),
                                                 out.ctypes.data, C.byref(st)), "ptx_render_counts_multi")
    return out, st
