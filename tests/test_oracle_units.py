"""Unit tests of the oracle's primitives against hand-computed values (the reference has no tests;
these pin the restatement's reading of intersection.hpp, random.hpp, brdf.hpp, main.cpp)."""
import ctypes as C
import math

import numpy as np

import oracle_binding as ob
import ptamd

FLT_MAX = np.finfo(np.float32).max


def f3(*v):
    return np.array(v, dtype=np.float32)


def slab(o, d, t, bmin, bmax):
    o, d, bmin, bmax = f3(*o), f3(*d), f3(*bmin), f3(*bmax)
    return ob.lib().oracle_slab(o.ctypes.data, d.ctypes.data, C.c_float(t), bmin.ctypes.data, bmax.ctypes.data)


def tri(o, d, v0, v1, v2):
    o, d = f3(*o), f3(*d)
    t9 = np.array([*v0, *v1, *v2], dtype=np.float32)
    b = np.zeros(2, dtype=np.float32)
    t = ob.lib().oracle_triangle(o.ctypes.data, d.ctypes.data, t9.ctypes.data, b.ctypes.data)
    return t, b


def test_slab_hit_miss_and_limits():
    # intersection.hpp:48-75
    assert slab((0, 0, -5), (0, 0, 1), FLT_MAX, (-1, -1, -1), (1, 1, 1)) == 4.0
    assert slab((0, 0, -5), (0, 0, 1), 3.0, (-1, -1, -1), (1, 1, 1)) == FLT_MAX      # tmin < ray.t fails
    assert slab((0, 0, 5), (0, 0, 1), FLT_MAX, (-1, -1, -1), (1, 1, 1)) == FLT_MAX    # box behind: tmax > 0 fails
    assert slab((0, 0, 0), (0, 0, 1), FLT_MAX, (-1, -1, -1), (1, 1, 1)) == -1.0       # origin inside: tmin < 0 is returned
    assert slab((3, 0, -5), (0, 0, 1), FLT_MAX, (-1, -1, -1), (1, 1, 1)) == FLT_MAX   # parallel, outside the slab


def test_slab_nan_follows_glm_min_max():
    # direction.x == 0 and origin.x on a box plane: (plane - origin.x) * inv.x = 0 * inf = NaN.
    # glm::min(x, y) = (y < x) ? y : x and glm::max(x, y) = (x < y) ? y : x keep a NaN that sits in x.
    assert slab((-1, 0, -5), (0, 0, 1), FLT_MAX, (-1, -1, -1), (1, 1, 1)) == FLT_MAX  # NaN in t1.x -> tmin NaN -> miss
    assert slab((1, 0, -5), (0, 0, 1), FLT_MAX, (-1, -1, -1), (1, 1, 1)) == FLT_MAX   # t1.x = -inf, t2.x = NaN -> tmax = -inf
    # Flat box in x with the ray inside its plane: both t1.x and t2.x are NaN.  GLM semantics: miss.
    # (IEEE minNum/maxNum -- fminf/fmaxf, v_min_f32 -- would drop both NaNs and report a hit at t = 4:
    # this is the one case where the two semantics differ, and why the HIP kernel keeps an exact path.)
    assert slab((1, 0, -5), (0, 0, 1), FLT_MAX, (1, -1, -1), (1, 1, 1)) == FLT_MAX
    # same flat box, ray not parallel to it: ordinary hit
    assert slab((0, 0, -5), (0.25, 0, 1), FLT_MAX, (1, -1, -1), (1, 1, 1)) == 4.0


def test_triangle_basic_and_rejections():
    # intersection.hpp:9-44
    t, b = tri((0.25, 0.25, -1), (0, 0, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0))
    assert t == 1.0 and b[0] == 0.25 and b[1] == 0.25
    assert tri((2, 2, -1), (0, 0, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0))[0] == FLT_MAX        # outside
    assert tri((0.25, 0.25, 1), (0, 0, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0))[0] == FLT_MAX   # behind: t <= 0
    assert tri((0.25, 0.25, -1), (1, 0, 0), (0, 0, 0), (1, 0, 0), (0, 1, 0))[0] == FLT_MAX  # parallel: |a| < 1e-4
    # epsilon is on the un-normalised determinant: a tiny triangle is rejected (scale dependent)
    assert tri((0.001, 0.001, -1), (0, 0, 1), (0, 0, 0), (0.005, 0, 0), (0, 0.005, 0))[0] == FLT_MAX
    # back faces are hit too (a < 0)
    t, _ = tri((0.25, 0.25, -1), (0, 0, 1), (0, 0, 0), (0, 1, 0), (1, 0, 0))
    assert t == 1.0


def test_random_vec2_draw_order_is_gxx():
    # dvec2(pcg_hash(s), pcg_hash(s)) under g++: arguments evaluated right to left -> first draw in y
    st = C.c_uint32(1023)
    a = ob.lib().oracle_random_float(C.byref(st))
    b = ob.lib().oracle_random_float(C.byref(st))
    st = C.c_uint32(1023)
    v = np.zeros(2, dtype=np.float32)
    ob.lib().oracle_random_vec2(C.byref(st), v.ctypes.data)
    assert v[1] == np.float32(a) and v[0] == np.float32(b)


def test_primary_ray_centre_and_fov():
    cam = ptamd.Camera((C.c_float * 3)(0, 0, 5), (C.c_float * 3)(0, 0, 0), (C.c_float * 3)(0, 1, 0), 90.0)
    d = np.zeros(3, dtype=np.float32)
    ob.lib().oracle_primary_ray(C.byref(cam), 100, 100, C.c_float(50.0), C.c_float(50.0), d.ctypes.data)
    assert np.allclose(d, [0, 0, -1], atol=1e-6)   # pixel centre at integer coordinates (gpuRender.cu:133-137)
    ob.lib().oracle_primary_ray(C.byref(cam), 100, 100, C.c_float(100.0), C.c_float(50.0), d.ctypes.data)
    assert np.allclose(d, [math.sqrt(0.5), 0, -math.sqrt(0.5)], atol=1e-6)  # 45 degrees at the edge for fov 90


def test_sampling_statistics_and_frames():
    rng = np.random.default_rng(3)
    n = f3(0.3, -0.5, 0.8)
    n /= np.linalg.norm(n)
    out = np.zeros(3, dtype=np.float32)
    cos_sum = 0.0
    for _ in range(4000):
        xi = rng.random(2).astype(np.float32)
        ob.lib().oracle_sample_cosine(xi.ctypes.data, n.ctypes.data, out.ctypes.data)
        assert abs(np.linalg.norm(out) - 1) < 1e-5
        c = float(out @ n)
        assert c >= -1e-6
        cos_sum += c
    assert abs(cos_sum / 4000 - 2.0 / 3.0) < 0.02  # E[cos] = 2/3 for a cosine-weighted hemisphere
    # GGX with tiny roughness concentrates the half vector on N (alpha = roughness, brdf.hpp:45-47)
    xi = f3(0.5, 0.25)[:2].copy()
    ob.lib().oracle_sample_ggx(xi.ctypes.data, n.ctypes.data, C.c_float(0.01), out.ctypes.data)
    assert float(out @ n) > 0.999


def test_tonemap_matches_formula():
    # main.cpp:225-233: c /= spp; c *= smoothstep(-0.75, 1.45, c); ACES(0.275 c); pow(1/2.2)
    s = f3(50.0, 10.0, 0.0)
    out = np.zeros(3, dtype=np.float32)
    ob.lib().oracle_tonemap(s.ctypes.data, C.c_float(100.0), out.ctypes.data)
    for ch in range(3):
        c = s[ch] / 100.0
        t = min(max((c + 0.75) / (1.45 + 0.75), 0.0), 1.0)
        c *= t * t * (3 - 2 * t)
        x = 0.275 * c
        c = min(max((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0.0), 1.0)
        assert abs(out[ch] - c ** (1 / 2.2)) < 1e-5


def test_environment_index_directions(scene2):
    # main.cpp:67-75 with the 180-degree rotation about Y: +Y looks at row 0, -Y at the last row
    W, H = scene2.desc.env_width, scene2.desc.env_height
    up = f3(0, 1, 0)
    assert ob.lib().oracle_environment_index(scene2.desc_ptr, up.ctypes.data) < W
    down = f3(0, -1, 0)
    assert ob.lib().oracle_environment_index(scene2.desc_ptr, down.ctypes.data) == W * H - 1  # clamped (gpuRender.cu:19)
    # rotateY(d, -pi) maps +X to -X: atan2(0, -1) = pi -> u = W (spills into the next row, kept as in the reference)
    px = f3(1, 0, 0)
    idx = ob.lib().oracle_environment_index(scene2.desc_ptr, px.ctypes.data)
    assert idx // W in (H // 2, H // 2 + 1)


def test_trace_single_rays_scene2(scene2):
    # a ray from the camera towards the target hits the room; a ray pointing away misses
    o = np.array(list(scene2.camera.position), dtype=np.float32)
    d = -o / np.linalg.norm(o)
    t = C.c_float(); tri_i = C.c_uint32(); cnt = C.c_uint32()
    b = np.zeros(2, dtype=np.float32)
    inst = ob.lib().oracle_trace(scene2.desc_ptr, o.ctypes.data, d.astype(np.float32).ctypes.data, C.byref(t), C.byref(tri_i),
                                 b.ctypes.data, C.byref(cnt))
    assert inst == 0 and 0 < t.value < 5 and tri_i.value < 3828 and cnt.value > 0
    inst = ob.lib().oracle_trace(scene2.desc_ptr, o.ctypes.data, (-d).astype(np.float32).ctypes.data, C.byref(t), C.byref(tri_i),
                                 b.ctypes.data, C.byref(cnt))
    assert inst == 0xFFFFFFFF and t.value == FLT_MAX
