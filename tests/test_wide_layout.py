"""The four-wide quantised re-layout the radiance path traverses (path-tracer_amd/csrc/wide_bvh.hpp).

CPU part: ptx_check_wide_layout rebuilds the layout on the host and verifies, with the device's own
arithmetic (fmaf(byte, scale, origin)), that every child's quantised box covers the exact box of the
binary node it stands for and that every leaf is referenced exactly once -- the two properties that make
the wide traversal visit a superset of the binary traversal's leaves.  GPU part: the guarded reciprocal
is bit-identical to IEEE division on its whole range, and the two layouts find the same hits."""
import ctypes as C

import numpy as np
import pytest

import ptamd


def _report(scene):
    rep = (C.c_uint64 * 6)()
    lib = ptamd.backend_lib()
    lib.ptx_check_wide_layout.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    rc = lib.ptx_check_wide_layout(scene.desc_ptr, rep)
    assert rc == 0, lib.ptx_last_error()
    return [int(v) for v in rep]


@pytest.mark.parametrize("which", ["scene2", "scene1_small", "scene0"])
def test_quantised_boxes_cover_and_every_leaf_is_kept(which, request):
    scene = request.getfixturevalue(which)
    nodes, slots, uncovered, bad_leaves, depth, folded = _report(scene)
    assert nodes > 0 and slots >= 2 * nodes
    assert uncovered == 0
    assert bad_leaves == 0
    assert folded > 0          # every other level really is folded away
    assert slots <= 4 * nodes
    assert 0 < depth < 64


def test_malformed_tree_is_rejected_not_looped(scene2):
    """A cycle in the node links must come back as an error (the builder bounds its own work)."""
    d = scene2.desc
    n = d.blas_node_total
    nodes = (ptamd.BvhNode * n).from_address(C.addressof(d.blas_nodes.contents))
    saved = nodes[1].left_first, nodes[1].count
    try:
        nodes[1].left_first, nodes[1].count = 1, 0   # node 1 becomes its own left child
        rep = (C.c_uint64 * 6)()
        lib = ptamd.backend_lib()
        lib.ptx_check_wide_layout.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        assert lib.ptx_check_wide_layout(scene2.desc_ptr, rep) != 0
    finally:
        nodes[1].left_first, nodes[1].count = saved


def test_out_of_range_links_are_rejected_before_the_layout_is_built(scene2):
    """ptx_check_wide_layout takes descriptions that never went through ptx_create: a root whose children lie
    outside the node array must be an error, not a read past the array."""
    d = scene2.desc
    n = d.blas_node_total
    nodes = (ptamd.BvhNode * n).from_address(C.addressof(d.blas_nodes.contents))
    lib = ptamd.backend_lib()
    lib.ptx_check_wide_layout.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    for bad in (n, n - 1, 2, 0x7FFFFFFF):   # past the end; right child past the end; even (not 2k+1); far away
        saved = nodes[0].left_first, nodes[0].count
        try:
            nodes[0].left_first, nodes[0].count = bad, 0
            rep = (C.c_uint64 * 6)()
            assert lib.ptx_check_wide_layout(scene2.desc_ptr, rep) != 0, bad
        finally:
            nodes[0].left_first, nodes[0].count = saved


@pytest.mark.parametrize("which", ["scene2", "scene0"])
def test_tight_tlas_refit_invariants_on_the_host(which, request):
    """csrc/tight_bounds.hpp without a GPU (device -1: bounds from the host loop): no tight box leaves the reference's
    box, the links are untouched, every leaf holds the padded bounds of its instances."""
    scene = request.getfixturevalue(which)
    lib = ptamd.backend_lib()
    lib.ptx_check_tight_tlas.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    rep = (C.c_uint64 * 6)()
    assert lib.ptx_check_tight_tlas(scene.desc_ptr, -1, rep) == 0, lib.ptx_last_error()
    n, checked, differ, outside, leaf_misses, ratio = [int(v) for v in rep]
    assert n == scene.desc.instance_count and checked == n and differ == 0 and outside == 0 and leaf_misses == 0
    assert 0 < ratio <= 1_000_000
    if which == "scene0":
        assert ratio < 900_000   # three rotated busts: the reference's boxes are a sixth larger


@pytest.mark.gpu
def test_reciprocal_is_ieee_on_its_whole_range():
    lib = ptamd.backend_lib()
    lib.ptx_selftest_reciprocal.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
    bad = C.c_uint64(123)
    assert lib.ptx_selftest_reciprocal(0, C.byref(bad)) == 0, lib.ptx_last_error()
    assert bad.value == 0


@pytest.mark.gpu
def test_wide_and_binary_layouts_count_the_same_rays(gpu_scenes):
    """ptx_measure on the radiance path (wide layout): every ray of the binary-layout oracle is there
    (same seeds, same paths) and node steps are roughly halved."""
    import oracle_binding as ob
    hosts, devs = gpu_scenes
    p = ptamd.make_params(200, 120, spp=4, max_bounces=6)
    st = devs[1].measure(hosts[1].camera, p)
    _, st_ref = ob.render(hosts[1], p, seed_mode=ptamd.SEED_REFERENCE_GPU)
    assert abs(int(st.rays) - int(st_ref.rays)) <= max(8, int(0.02 * st_ref.rays))
    assert st.node_steps < 0.7 * st_ref.node_steps
    assert abs(int(st.tri_tests) - int(st_ref.tri_tests)) <= 0.1 * st_ref.tri_tests


@pytest.mark.gpu
def test_wide_and_binary_layouts_render_the_same_image(gpu_scenes, tmp_path):
    """Same seeds, same triangles, same Moeller-Trumbore arithmetic: the two layouts may only differ where a
    tie in t or a one-rounding box miss resolves differently, i.e. in (almost) no pixel."""
    import os
    import subprocess
    import sys
    hosts, devs = gpu_scenes
    out = str(tmp_path / "binary.npy")
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {os.path.dirname(ptamd.__file__)!r})\n"
        "import ptamd\n"
        "for preset in (2, 1):\n"
        "    s = ptamd.HostScene(preset); d = ptamd.DeviceScene(s.desc_ptr)\n"
        "    img, _ = d.render(s.camera, ptamd.make_params(160, 96, spp=8, max_bounces=6))\n"
        f"    np.save({out!r} + str(preset) + '.npy', img)\n"
    )
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PTX_WIDE="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    for preset, min_same in ((2, 0.999), (1, 0.995)):
        binary = np.load(out + str(preset) + ".npy")
        wide, _ = devs[preset].render(hosts[preset].camera, ptamd.make_params(160, 96, spp=8, max_bounces=6))
        same = (np.abs(wide - binary).max(axis=2) == 0).mean()
        rms = float(np.sqrt(np.mean((wide.astype(np.float64) - binary) ** 2)))
        print(f"scene {preset}: {same * 100:.3f} % of pixels bit-identical, RMSE {rms:.2e}")
        assert same >= min_same
        assert rms < 5e-3


def test_tight_pad_covers_a_mesh_modelled_far_from_its_origin():
    """A mesh whose object-space coordinates are large (1e4) and whose instance moves it back to the origin: the fp32 rounding
    error of the transform scales with |M| |v| = 1e4, not with the world-space result (a few units), so a pad relative to the
    box alone (2^-16 of its size: round 2) does not cover what the device reaches through inv_model.  The pad carries an
    allowance of 2^-20 (|M0| |x| + |M1| |y| + |M2| |z| + |M3|) (csrc/tight_bounds.hpp: roundingAllowance); the self-check
    holds the padded box against the bounds taken in double."""
    from test_gpu_analytic import FloorScene
    off = np.array([10000.0, 0.0, 10000.0])
    ca, sa = np.cos(0.7), np.sin(0.7)
    rot_y = np.array([[ca, 0, sa, 0], [0, 1, 0, 0], [-sa, 0, ca, 0], [0, 0, 0, 1.0]])
    back = np.eye(4)
    back[:3, 3] = -off
    model = rot_y @ back                     # move the mesh to the origin, then turn it
    sc = FloorScene((0.5, 0.5, 0.5), (1.0, 1.0, 1.0), half=2.0, model=model)
    for t in sc.tris:                        # the mesh itself sits at (1e4, 0, 1e4)
        for v in (t.v0, t.v1, t.v2):
            v[0] += off[0]
            v[2] += off[2]
    sc.blas[0].aabb_min[:] = (off[0] - 2.0, 0.0, off[2] - 2.0)
    sc.blas[0].aabb_max[:] = (off[0] + 2.0, 0.0, off[2] + 2.0)
    sc.tlas[0].aabb_min[:] = (-3.0, -1e-3, -3.0)
    sc.tlas[0].aabb_max[:] = (3.0, 1e-3, 3.0)
    lib = ptamd.backend_lib()
    lib.ptx_check_tight_tlas.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    rep = (C.c_uint64 * 6)()
    assert lib.ptx_check_tight_tlas(sc.desc_ptr, -1, rep) == 0, lib.ptx_last_error()
    n, checked, differ, outside, misses, ratio = [int(v) for v in rep]
    assert n == 1 and checked == 1 and differ == 0 and outside == 0 and misses == 0

    # what round 2's pad would have done: fp32 bounds as the library computes them, padded by 2^-16 of the box
    inv = np.array(list(sc.inst[0].inv_model), dtype=np.float32).reshape(4, 4).T.astype(np.float64)   # column-major -> rows
    M = np.linalg.inv(inv)
    verts = np.array([list(v) for t in sc.tris for v in (t.v0, t.v1, t.v2)], dtype=np.float32)
    m32 = M[:3].astype(np.float32)
    w32 = (m32[:, 0][None] * verts[:, 0:1] + m32[:, 1][None] * verts[:, 1:2]) + m32[:, 2][None] * verts[:, 2:3]
    w32 = (w32 + m32[:, 3][None]).astype(np.float32)
    exact = verts.astype(np.float64) @ M[:3, :3].T + M[:3, 3]
    lo32, hi32 = w32.min(axis=0), w32.max(axis=0)
    old_pad = np.maximum(hi32 - lo32, np.maximum(np.abs(lo32), np.abs(hi32))) * 2.0 ** -16
    err = max(float((lo32 - exact.min(axis=0)).max()), float((exact.max(axis=0) - hi32).max()))
    allowance = 2.0 ** -20 * (np.abs(M[:3, :3]) @ np.array([off[0] + 2.0, 0.0, off[2] + 2.0]) + np.abs(M[:3, 3]))
    xz = [0, 2]                              # the floor is flat in y: nothing to bound there
    print(f"fp32 bounds miss the exact ones by {err:.2e}; old pad {old_pad[xz].max():.2e}, allowance {allowance[xz].min():.2e}")
    assert err > old_pad[xz].max()           # round 2's pad would have cut into the mesh
    assert allowance[xz].min() > 10 * err    # the allowance covers the fp32 error with room to spare
