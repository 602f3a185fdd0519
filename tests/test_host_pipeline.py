"""Host scene pipeline (path-tracer_amd/host): RNG, decimal parser, matrices, BVH invariants."""
import ctypes as C

import numpy as np

import ptamd


def pcg_python(state):
    """random.hpp:12-17 restated in Python integers."""
    state = (state * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return state, ((word >> 22) ^ word) & 0xFFFFFFFF


def test_pcg_hash_matches_definition():
    lib = ptamd.host_lib()
    for seed in (0, 1, 1023, 1031, 7142, 0xFFFFFFFF):
        st = C.c_uint32(seed)
        ps = seed
        for _ in range(64):
            ps, expect = pcg_python(ps)
            assert lib.pth_pcg_hash(C.byref(st)) == expect
            assert st.value == ps


def test_random_float_is_exact_scaling():
    lib = ptamd.host_lib()
    st = C.c_uint32(99)
    ps = 99
    for _ in range(1000):
        ps, h = pcg_python(ps)
        got = lib.pth_random_float(C.byref(st))
        # (double)h * 2^-32 narrowed to float (random.hpp:19-21)
        assert np.float32(got) == np.float32(np.float64(h) * (1.0 / 4294967296.0))


def test_decimal_parser_agrees_with_strtod_on_obj_style_numbers():
    lib = ptamd.host_lib()
    rng = np.random.default_rng(5)
    texts = ["0", "1", "-1", "0.5", "-0.573651", "0.001530", "123.456789", "1e3", "-2.5E-3", ".75", "-.125", "+4.0",
             "0.000001", "3.14159265358979", "1e-10", "7E+2"]
    texts += ["%.6f" % v for v in rng.uniform(-2, 2, 500)]
    for t in texts:
        out = C.c_double()
        assert lib.pth_parse_decimal(t.encode(), C.byref(out)) == 0, t
        # the digit-accumulating algorithm is within a few ulp (double) of the correctly rounded value
        assert abs(out.value - float(t)) <= 4e-16 * max(1.0, abs(float(t))), t
    for bad in ["", "abc", "-", "e5"]:
        out = C.c_double()
        assert lib.pth_parse_decimal(bad.encode(), C.byref(out)) != 0


def test_mat4_inverse():
    lib = ptamd.host_lib()
    rng = np.random.default_rng(1)
    for _ in range(50):
        m = rng.normal(size=(4, 4)).astype(np.float32)
        out = np.empty(16, dtype=np.float32)
        mc = np.ascontiguousarray(m.T.reshape(-1))  # column-major
        lib.pth_mat4_inverse(mc.ctypes.data, out.ctypes.data)
        inv = out.reshape(4, 4).T
        assert np.allclose(inv @ m, np.eye(4), atol=2e-3)


def test_mesh_matrix_chain():
    # mesh.cpp:18-41: M = S * Rx * Ry * Rz, world-space translation, then -centroid
    lib = ptamd.host_lib()
    cen = np.array([1.0, 2.0, 3.0], dtype=np.float32)
    rot = np.array([0.3, -1.1, 2.0], dtype=np.float32)
    tr = np.array([10.0, -5.0, 7.0], dtype=np.float32)
    model = np.empty(16, dtype=np.float32)
    inv = np.empty(16, dtype=np.float32)
    lib.pth_mesh_matrices(cen.ctypes.data, rot.ctypes.data, C.c_float(0.2), tr.ctypes.data, model.ctypes.data, inv.ctypes.data)
    M = model.reshape(4, 4).T.astype(np.float64)

    def R(axis, a):
        c, s = np.cos(a), np.sin(a)
        m = np.eye(4)
        i, j = [(1, 2), (2, 0), (0, 1)][axis]
        m[i, i] = c; m[j, j] = c; m[i, j] = -s; m[j, i] = s
        return m
    S = np.diag([0.2, 0.2, 0.2, 1.0])
    T1 = np.eye(4); T1[:3, 3] = tr
    T2 = np.eye(4); T2[:3, 3] = -cen
    expect = T1 @ S @ R(0, rot[0]) @ R(1, rot[1]) @ R(2, rot[2]) @ T2
    assert np.allclose(M, expect, atol=1e-4)
    assert np.allclose(inv.reshape(4, 4).T @ M, np.eye(4), atol=1e-4)


def check_bvh(nodes_u32, n_prims):
    nodes_f = nodes_u32.view(np.float32)
    n = len(nodes_u32)
    assert n % 2 == 1
    seen = np.zeros(n_prims, dtype=np.int32)
    visited = np.zeros(n, dtype=bool)
    stack = [0]
    next_expected = 1  # children are allocated pairwise in depth-first order (bvh.hpp:165-166,184-185)
    while stack:
        i = stack.pop()
        assert not visited[i]
        visited[i] = True
        left_first, count = int(nodes_u32[i, 6]), int(nodes_u32[i, 7])
        if count > 0:
            seen[left_first:left_first + count] += 1
            continue
        l, r = left_first, left_first + 1
        assert l == next_expected and l % 2 == 1 and r < n
        next_expected += 2
        for c in (l, r):  # child boxes inside the parent box
            assert (nodes_f[c, 0:3] >= nodes_f[i, 0:3]).all() and (nodes_f[c, 3:6] <= nodes_f[i, 3:6]).all()
        stack.append(r)
        stack.append(l)
    assert visited.all()
    assert (seen == 1).all()


def test_bvh_invariants_scene2(scene2):
    check_bvh(scene2.blas_nodes, scene2.desc.triangle_total)
    # leaf boxes contain their triangles
    nodes = scene2.blas_nodes
    tris = scene2.triangles.reshape(-1, 3, 3)
    nf = nodes.view(np.float32)
    for i in np.nonzero(nodes[:, 7] > 0)[0][:500]:
        lf, c = int(nodes[i, 6]), int(nodes[i, 7])
        t = tris[lf:lf + c].reshape(-1, 3)
        assert (t.min(0) >= nf[i, 0:3]).all() and (t.max(0) <= nf[i, 3:6]).all()


def test_tlas_invariants_scene1_small(scene1_small):
    check_bvh(scene1_small.tlas_nodes, scene1_small.desc.instance_count)


def test_heatmap_colour_and_bmp(tmp_path):
    counts = np.array([[0, 1], [2, 4]], dtype=np.uint32)
    rgb, mx = ptamd.heatmap_colour(counts)
    assert mx == 4
    assert (rgb[0, 0] == 0).all()            # count 0 stays black (main.cpp:369-373)
    assert rgb.min() >= 0.0 and rgb.max() <= 1.0
    assert rgb[1, 1, 0] > rgb[0, 1, 0]       # afmhot red channel rises with t
    path = str(tmp_path / "o.bmp")
    img = np.zeros((2, 3, 3), dtype=np.float32)
    img[0, 0] = (1.0, 0.0, 0.0)              # row 0 = bottom scan line
    img[1, 2] = (0.0, 0.5, 2.0)              # clamped, truncated (output.cpp:17-21)
    ptamd.write_bmp(path, img)
    raw = open(path, "rb").read()
    assert raw[:2] == b"BM" and len(raw) == 54 + 2 * 12
    row0 = raw[54:54 + 9]
    assert row0[0:3] == bytes([0, 0, 255])   # BGR, bottom row first
    row1 = raw[54 + 12:54 + 12 + 9]
    assert row1[6:9] == bytes([255, 127, 0])


def test_parallel_builder_makes_the_identical_tree(monkeypatch):
    """the multi-threaded SAH builder (sah_builder.hpp:buildBVHParallel) must produce the
    serial builder's tree bit for bit -- node boxes, numbering, leaf ranges and primitive order -- because
    the `-a` heat map counts tests on exactly that tree."""
    import numpy as np
    built = {}
    for threads in ("1", "8", "5"):
        monkeypatch.setenv("PT_BUILD_THREADS", threads)
        s = ptamd.HostScene(1, scatter_instances=50)   # 505,848-triangle BLAS: the parallel path
        built[threads] = (s.blas_nodes.copy(), s.triangles.copy(), s.tlas_nodes.copy(), s.info.blas_build_s)
        s.close()
    for threads in ("8", "5"):
        assert np.array_equal(built["1"][0], built[threads][0])
        assert np.array_equal(built["1"][1], built[threads][1])
        assert np.array_equal(built["1"][2], built[threads][2])
    assert built["1"][0].shape[0] > 500000


def test_flattened_scene_cache(tmp_path):
    """pth_scene_create_cached keys the flattened arrays by the content of the assets;
    a hit hands back byte-identical arrays, camera and info; a damaged file is ignored and rewritten."""
    import glob
    import os
    import numpy as np

    def arrays(s):
        return [s.tlas_nodes.copy(), s.blas_nodes.copy(), s.triangles.copy(), s.instances.copy(),
                np.ctypeslib.as_array(s.desc.environment, shape=(s.desc.env_width * s.desc.env_height * 3,)).copy(),
                np.ctypeslib.as_array(s.desc.texels, shape=(s.desc.texel_total * 3,)).copy(),
                np.ctypeslib.as_array(s.desc.face_normals, shape=(s.desc.face_normal_total * 3,)).copy()]

    plain = ptamd.HostScene(2, sky=(64, 32))
    first = ptamd.HostScene(2, sky=(64, 32), cache_dir=tmp_path)
    assert not first.cache_hit
    files = glob.glob(str(tmp_path / "scene2_*.ptxflat"))
    assert len(files) == 1
    second = ptamd.HostScene(2, sky=(64, 32), cache_dir=tmp_path)
    assert second.cache_hit
    for a, b, c in zip(arrays(plain), arrays(first), arrays(second)):
        assert a.tobytes() == b.tobytes() == c.tobytes()
    assert bytes(plain.camera) == bytes(second.camera)
    for f in ("triangles_per_mesh", "blas_nodes", "tlas_nodes", "instances", "blas_depth", "stand_in_environment"):
        assert getattr(plain.info, f) == getattr(second.info, f)
    assert second.info.blas_nodes == 4899          # the README pin survives the round trip
    # other parameters -> other key
    other = ptamd.HostScene(2, sky=(32, 16), cache_dir=tmp_path)
    assert not other.cache_hit and len(glob.glob(str(tmp_path / "*.ptxflat"))) == 2
    # a truncated file is not trusted
    size = os.path.getsize(files[0])
    with open(files[0], "r+b") as f:
        f.truncate(size // 2)
    again = ptamd.HostScene(2, sky=(64, 32), cache_dir=tmp_path)
    assert not again.cache_hit
    assert os.path.getsize(files[0]) == size
    # a header whose environment size disagrees with the stored array is not trusted either (the backend would
    # read env_width * env_height texels): CacheHeader ends with envWidth, envHeight
    import ctypes as C
    header = 8 + 8 + C.sizeof(ptamd.Camera) + C.sizeof(ptamd.HostInfo)
    header = (header + 3) // 4 * 4
    with open(files[0], "r+b") as f:
        f.seek(header)
        w, h = np.frombuffer(f.read(8), dtype=np.uint32)
        assert (w, h) == (64, 32), "test out of date: CacheHeader layout changed"
        f.seek(header)
        f.write(np.array([4096, 4096], dtype=np.uint32).tobytes())
    lying = ptamd.HostScene(2, sky=(64, 32), cache_dir=tmp_path)
    assert not lying.cache_hit and lying.desc.env_width == 64
    assert (os.stat(tmp_path).st_mode & 0o777) in (0o700, 0o755, 0o775, 0o777)  # pytest made it; ours are created 0700
    for s in (plain, first, second, other, again, lying):
        s.close()


def test_stl_ingest_of_a_scan_sized_file(tmp_path):
    """the Menelaus scan is absent, so the stand-in bust is written as a binary STL of the
    scan's size, with facets that name a vertex twice mixed in; the memory-mapped loader has to drop
    exactly those (stl_reader.h:498-521) and hand the scene pipeline the same triangles as the generator."""
    import shutil
    import time
    models = tmp_path / "models"
    models.mkdir()
    n = ptamd.host_lib().pth_write_standin_stl(str(models / "bust-of-menelaus.stl").encode(), 1000)
    assert n == 505848 + 505
    assert (models / "bust-of-menelaus.stl").stat().st_size == 84 + 50 * n
    t0 = time.time()
    from_file = ptamd.HostScene(0, models_dir=str(models), sky=(64, 32))
    dt = time.time() - t0
    generated = ptamd.HostScene(0, sky=(64, 32))
    assert from_file.info.stand_in_geometry == 0 and generated.info.stand_in_geometry == 1
    assert from_file.info.triangles_per_mesh == 505848
    assert from_file.triangles.tobytes() == generated.triangles.tobytes()
    assert from_file.blas_nodes.tobytes() == generated.blas_nodes.tobytes()
    assert dt < 60
    from_file.close()
    generated.close()
    shutil.rmtree(models)
