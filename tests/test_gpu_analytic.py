"""Known answers that need no oracle: scenes small enough to integrate by hand, handed to the backend through the
C ABI as a caller of the reference would (no host pipeline: the description is built here, field by field).

A square floor under a uniform sky.  Every camera ray hits the floor (Lambert, albedo a); the bounce leaves upwards
from p + 1e-4 N and can only reach the sky (radiance L), so every sample of every pixel returns a * L exactly
(gpuRender.cu:24-116: throughput *= albedo at the hit, *= environment at the miss) and the pixel is
grade(a * L) (gpuRender.cu:158-166) whatever the random numbers were.  With one bounce the loop ends after the hit and
the throughput is returned as it is (gpuRender.cu:27-29): grade(a).  The description also exercises the smallest
legal trees -- a TLAS that is one leaf, a BLAS that is one leaf of two triangles -- a 2x1 environment, no textures,
no vertex attributes."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
import ptamd

pytestmark = pytest.mark.gpu

IDENTITY = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]


def grade(c):
    """gpuRender.cu:158-166 per channel (numpy, float64)."""
    c = np.asarray(c, dtype=np.float64)
    t = np.clip((c + 0.75) / (1.45 + 0.75), 0.0, 1.0)
    c = c * (t * t * (3.0 - 2.0 * t))
    x = 0.275 * c
    c = np.clip((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0.0, 1.0)
    return c ** (1.0 / 2.2)


class FloorScene:
    """ptx_scene_desc of the floor; keeps the arrays alive."""

    def __init__(self, albedo, sky, metalness=0.0, roughness=1.0, half=50.0):
        self.tlas = (ptamd.BvhNode * 1)()
        self.blas = (ptamd.BvhNode * 1)()
        for n, cnt in ((self.tlas[0], 1), (self.blas[0], 2)):
            n.aabb_min[:] = (-half, 0.0, -half)
            n.aabb_max[:] = (half, 0.0, half)
            n.left_first, n.count = 0, cnt
        self.tris = (ptamd.Triangle * 2)()
        self.tris[0].v0[:], self.tris[0].v1[:], self.tris[0].v2[:] = (-half, 0, -half), (-half, 0, half), (half, 0, half)
        self.tris[1].v0[:], self.tris[1].v1[:], self.tris[1].v2[:] = (-half, 0, -half), (half, 0, half), (half, 0, -half)
        self.face_normals = (C.c_float * 6)(0, 1, 0, 0, 1, 0)
        self.geom = (ptamd.Geometry * 1)()
        g = self.geom[0]
        g.node_offset, g.node_count, g.tri_offset, g.tri_count = 0, 1, 0, 2
        g.face_normal_offset, g.vtx_normal_offset, g.vtx_uv_offset = 0, 0xFFFFFFFF, 0xFFFFFFFF
        self.inst = (ptamd.Instance * 1)()
        self.inst[0].inv_model[:] = IDENTITY
        self.inst[0].normal_matrix[:] = IDENTITY
        self.inst[0].geometry_id = self.inst[0].material_id = 0
        self.mat = (ptamd.Material * 1)()
        m = self.mat[0]
        m.albedo[:] = albedo
        m.metalness, m.roughness = metalness, roughness
        m.emissive[:] = (0, 0, 0)
        m.f0[:] = (0.04, 0.04, 0.04)
        m.albedo_texture = m.emissive_texture = 0xFFFFFFFF
        self.env = (C.c_float * 6)(*(list(sky) * 2))
        d = self.desc = ptamd.SceneDesc()
        d.tlas_nodes, d.tlas_node_count = self.tlas, 1
        d.instances, d.instance_count = self.inst, 1
        d.geometries, d.geometry_count = self.geom, 1
        d.blas_nodes, d.blas_node_total = self.blas, 1
        d.triangles, d.triangle_total = self.tris, 2
        d.face_normals, d.face_normal_total = self.face_normals, 2
        d.materials, d.material_count = self.mat, 1
        d.environment, d.env_width, d.env_height = self.env, 2, 1
        self.desc_ptr = C.pointer(d)
        self.camera = ptamd.Camera()
        self.camera.position[:] = (0.3, 5.0, -0.2)
        self.camera.target[:] = (0.0, 0.0, 0.0)
        self.camera.up[:] = (0.0, 0.0, 1.0)
        self.camera.fov_deg = 60.0


@pytest.fixture(scope="module")
def torch_first():
    try:
        import torch
        torch.cuda.is_available()
    except ImportError:
        pass


@pytest.mark.parametrize("albedo,sky", [((0.5, 0.25, 0.75), (0.8, 0.8, 0.8)), ((1.0, 1.0, 1.0), (2.0, 0.5, 0.1)),
                                        ((0.18, 0.18, 0.18), (1.0, 1.0, 1.0))])
def test_floor_under_a_uniform_sky(torch_first, albedo, sky):
    sc = FloorScene(albedo, sky)
    dev = ptamd.DeviceScene(sc.desc_ptr)
    try:
        a, L = np.array(albedo), np.array(sky)
        for spp, bounces, streams, expect in ((1, 6, 0, grade(a * L)), (7, 2, 0, grade(a * L)), (16, 6, 4, grade(a * L)),
                                              (3, 1, 0, grade(a))):
            p = ptamd.make_params(96, 64, spp=spp, max_bounces=bounces, seed_mode=ptamd.SEED_STREAM if streams else ptamd.SEED_REFERENCE_GPU,
                                  sub_streams=streams)
            img, st = dev.render(sc.camera, p)
            err = np.abs(img - expect.astype(np.float32)).max()
            assert err < 2e-6, (spp, bounces, streams, float(err), img[0, 0], expect)
            # every path: one hit; and, if it may bounce, one miss
            m = dev.measure(sc.camera, p)
            assert m.hits == 96 * 64 * spp and m.misses == (0 if bounces == 1 else 96 * 64 * spp)
            ref, _ = ob.render(sc, p, seed_mode=p.seed_mode)
            assert np.abs(ref - img).max() < 2e-6
        # the heat map of one leaf under one leaf: the reference counts no box tests at all for such trees
        # (a root that is a leaf is never slab-tested, intersection.hpp:99-122), and both sides agree on that
        pc = ptamd.make_params(96, 64)
        ref, _ = ob.render_counts(sc, pc)
        got, _ = dev.render_counts(sc.camera, pc)
        assert np.array_equal(ref, got)
    finally:
        dev.close()


def test_mirror_floor_keeps_energy_below_one(torch_first):
    """A metallic floor (F0 = albedo = 1, so Fresnel is 1) under a unit sky: every sample returns the GGX weight
    G * VdotH / (NdotV * NdotH) of one bounce (gpuRender.cu:71-105).  No closed form, but the mean over many samples is
    the directional albedo of the lobe, which is below one and, for a smooth floor, close to it; and the HIP path and
    the oracle, fed the same random numbers, agree to rounding."""
    for roughness, lo in ((0.05, 0.97), (0.5, 0.6)):
        sc = FloorScene((1.0, 1.0, 1.0), (1.0, 1.0, 1.0), metalness=1.0, roughness=roughness)
        dev = ptamd.DeviceScene(sc.desc_ptr)
        try:
            p = ptamd.make_params(64, 48, spp=64, max_bounces=2)
            import torch
            acc = torch.zeros((48, 64, 4), dtype=torch.float32, device="cuda")
            dev.accumulate_device(sc.camera, p, acc.data_ptr(), None)
            torch.cuda.synchronize()
            a = acc.cpu().numpy()
            mean = a[..., :3].sum() / (3.0 * a[..., 3].sum())
            ref, _ = ob.accumulate(sc, p, seed_mode=ptamd.SEED_REFERENCE_GPU)
            print(f"roughness {roughness}: mean throughput {mean:.4f}; oracle {ref[..., :3].sum() / (3.0 * ref[..., 3].sum()):.4f}")
            assert lo < mean <= 1.0 + 1e-3
            # a reflected ray that dips below the floor hits it again from behind the offset origin: rare, and the
            # same on both sides
            assert np.abs(a - ref).max() < 1e-3 * 64
        finally:
            dev.close()
