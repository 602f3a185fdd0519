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

    def __init__(self, albedo, sky, metalness=0.0, roughness=1.0, half=50.0, env=None, textures=None, emissive=(0, 0, 0), model=None, corner_normal=None):
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
        if model is not None:
            # mesh.cpp:6-16: the instance keeps inverse(model) and transpose(inverse(model)); GLM matrices are column-major;
            # the TLAS leaf bounds the eight (here four) transformed corners of the BLAS root box (mesh.cpp:18-41)
            inv = np.linalg.inv(np.asarray(model, dtype=np.float64))
            self.inst[0].inv_model[:] = [float(c) for c in inv.T.reshape(-1)]
            self.inst[0].normal_matrix[:] = [float(c) for c in inv.reshape(-1)]   # transpose(inv), column-major = inv row-major
            corners = np.array([[sx * half, 0.0, sz * half, 1.0] for sx in (-1, 1) for sz in (-1, 1)]) @ np.asarray(model, dtype=np.float64).T
            self.tlas[0].aabb_min[:] = [float(c) for c in corners[:, :3].min(axis=0) - 1e-4]
            self.tlas[0].aabb_max[:] = [float(c) for c in corners[:, :3].max(axis=0) + 1e-4]
            self.inv_model = inv
        self.inst[0].geometry_id = self.inst[0].material_id = 0
        self.mat = (ptamd.Material * 1)()
        m = self.mat[0]
        m.albedo[:] = albedo
        m.metalness, m.roughness = metalness, roughness
        m.emissive[:] = emissive
        m.f0[:] = (0.04, 0.04, 0.04)
        m.albedo_texture = m.emissive_texture = 0xFFFFFFFF
        self.half = half
        if corner_normal is not None:
            # one normal per triangle corner (geometry.cpp:69-87), a function of the corner's position: not normalised
            self.normal_array = np.array([corner_normal(v[0], v[2]) for t in self.tris for v in (t.v0, t.v1, t.v2)], dtype=np.float32)
            g.vtx_normal_offset = 0
        if textures is not None:
            # textures = (albedo texels (h, w, 3), emissive texels (h', w', 3), uv scale, uv offset): the floor's corner
            # (-half, -half) has uv = offset, the opposite one offset + scale, linear in between
            ta, te, self.uv_scale, self.uv_offset = textures
            self.tex = (ptamd.Texture * 2)()
            self.tex[0].width, self.tex[0].height, self.tex[0].texel_offset = ta.shape[1], ta.shape[0], 0
            self.tex[1].width, self.tex[1].height, self.tex[1].texel_offset = te.shape[1], te.shape[0], ta.shape[0] * ta.shape[1]
            self.texel_array = np.ascontiguousarray(np.concatenate([ta.reshape(-1, 3), te.reshape(-1, 3)]), dtype=np.float32)
            uv = []
            for t in self.tris:
                for v in (t.v0, t.v1, t.v2):
                    uv += [self.uv_offset + self.uv_scale * (v[0] + half) / (2 * half), self.uv_offset + self.uv_scale * (v[2] + half) / (2 * half)]
            self.uv_array = np.array(uv, dtype=np.float32)
            g.vtx_uv_offset = 0
            m.albedo_texture, m.emissive_texture = 0, 1
        if env is None:
            env = np.array([[sky, sky]], dtype=np.float32)          # 2 x 1 texels of one colour
        self.env_array = np.ascontiguousarray(env, dtype=np.float32)
        self.env = self.env_array.ctypes.data_as(C.POINTER(C.c_float))
        d = self.desc = ptamd.SceneDesc()
        d.tlas_nodes, d.tlas_node_count = self.tlas, 1
        d.instances, d.instance_count = self.inst, 1
        d.geometries, d.geometry_count = self.geom, 1
        d.blas_nodes, d.blas_node_total = self.blas, 1
        d.triangles, d.triangle_total = self.tris, 2
        d.face_normals, d.face_normal_total = self.face_normals, 2
        d.materials, d.material_count = self.mat, 1
        if textures is not None:
            d.textures, d.texture_count = self.tex, 2
            d.texels, d.texel_total = self.texel_array.ctypes.data_as(C.POINTER(C.c_float)), len(self.texel_array)
            d.vtx_uvs, d.vtx_uv_total = self.uv_array.ctypes.data_as(C.POINTER(C.c_float)), 6
        if corner_normal is not None:
            d.vtx_normals, d.vtx_normal_total = self.normal_array.ctypes.data_as(C.POINTER(C.c_float)), 6
        d.environment, d.env_width, d.env_height = self.env, self.env_array.shape[1], self.env_array.shape[0]
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


def primary_directions(cam, width, height):
    """camera.hpp:8-20 and gpuRender.cu:118-137 in float64, one sample per pixel (no jitter): (height, width, 3)."""
    pos, at, up = (np.array(list(v), dtype=np.float64) for v in (cam.position, cam.target, cam.up))
    z = (at - pos) / np.linalg.norm(at - pos)
    x = np.cross(z, up)
    x /= np.linalg.norm(x)
    y = np.cross(x, z)
    px, py = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    v = np.stack([px - 0.5 * width, py - 0.5 * height,
                  np.full_like(px, -(0.5 * height) / np.tan(0.5 * np.radians(cam.fov_deg)))], axis=-1)
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    return v[..., 0:1] * x + v[..., 1:2] * y + v[..., 2:3] * (-z)


def environment_texels(d, env_w, env_h):
    """main.cpp:54-75 / gpuRender.cu:14-22 in float64: the direction turned by pi about Y, then longitude / latitude."""
    u = (env_w * (np.arctan2(-d[..., 2], -d[..., 0]) / (2.0 * np.pi) + 0.5)).astype(np.int64)
    v = (env_h * np.arccos(np.clip(d[..., 1], -1.0, 1.0)) / np.pi).astype(np.int64)
    return np.minimum(u + v * env_w, env_w * env_h - 1)


SKY_RES = (192, 128)
SKY_VIEWS = [((0.3, 9.0, 0.4), (0, 0, 1), 90.0), ((5.0, 6.5, 1.0), (0, 1, 0), 50.0),
             ((-3.0, 5.5, -4.0), (0, 1, 0), 100.0), ((0.0, 5.2, 8.0), (0.2, 1, 0), 75.0)]


def sky_view(target, up, fov):
    """The floor scene under a 16 x 8 environment of random colours, seen from 5 above the floor; returns the scene,
    the expected frame at one sample per pixel and the mask of the pixels that look above the horizon."""
    env_w, env_h = 16, 8
    env = np.random.default_rng(11).uniform(0.05, 3.0, size=(env_h, env_w, 3)).astype(np.float32)
    sc = FloorScene((0.5, 0.5, 0.5), (1, 1, 1), env=env)
    sc.camera.position[:] = (0.0, 5.0, 0.0)
    sc.camera.target[:] = target
    n = float(np.linalg.norm(up))
    sc.camera.up[:] = [c / n for c in up]
    sc.camera.fov_deg = fov
    d = primary_directions(sc.camera, *SKY_RES)
    expect = grade(env.reshape(-1, 3)[environment_texels(d, env_w, env_h)])
    sky = d[..., 1] > 0.02                                            # the rest may see the floor
    assert sky.mean() > 0.4
    return sc, expect, sky


@pytest.mark.parametrize("target,up,fov", SKY_VIEWS)
def test_sky_only_views_show_the_environment_map(torch_first, target, up, fov):
    """Rays that leave upwards from above the floor hit nothing: the pixel is grade(environment texel) and the texel
    follows from the camera and the environment mapping alone, computed here in numpy from the reference's formulas.
    Pins camera basis, field of view, pixel centres, image orientation and the orientation of the environment map
    (the turn by pi about Y, which row is up, which way the longitude runs) on the GPU without the oracle."""
    sc, expect, sky = sky_view(target, up, fov)
    W, H = SKY_RES
    dev = ptamd.DeviceScene(sc.desc_ptr)
    try:
        p = ptamd.make_params(W, H, spp=1, max_bounces=6)
        img, _ = dev.render(sc.camera, p)
    finally:
        dev.close()
    same = np.abs(img - expect.astype(np.float32)).max(axis=2) < 2e-6
    bad = int((~same & sky).sum())
    print(f"sky pixels {int(sky.sum())}, differing {bad}")
    # a direction within rounding of a texel border may fall either side in fp32: a handful of pixels at most
    assert bad <= sky.sum() // 1000, bad
    # and the oracle agrees with both
    ref, _ = ob.render(sc, p, seed_mode=p.seed_mode)
    assert int(((np.abs(ref - expect.astype(np.float32)).max(axis=2) >= 2e-6) & sky).sum()) <= sky.sum() // 1000


TEX_RES = (160, 120)


def textured_floor():
    """A floor with an albedo and an emissive texture under a uniform sky, seen from above; returns the scene, the
    expected frame at one sample per pixel -- grade((albedo * texel_a(uv) + emissive * texel_e(uv)) * sky), uv from
    the hit point, texel from image.hpp:27-35 -- and the mask of the pixels whose texel is not in doubt."""
    rng = np.random.default_rng(5)
    ta = rng.uniform(0.1, 1.0, size=(4, 8, 3)).astype(np.float32)
    te = rng.uniform(0.0, 2.0, size=(3, 5, 3)).astype(np.float32)
    albedo, emissive, sky = np.array((0.9, 0.7, 0.5)), np.array((0.2, 0.4, 0.6)), np.array((1.5, 1.0, 0.7))
    sc = FloorScene(albedo, sky, textures=(ta, te, 3.0, -0.75), emissive=emissive, half=4.0)
    sc.camera.position[:] = (0.4, 6.0, -0.3)
    sc.camera.target[:] = (0.0, 0.0, 0.2)
    sc.camera.fov_deg = 50.0
    d = primary_directions(sc.camera, *TEX_RES)
    pos = np.array(list(sc.camera.position), dtype=np.float64)
    t = -pos[1] / d[..., 1]
    x, z = pos[0] + t * d[..., 0], pos[2] + t * d[..., 2]
    on_floor = (np.abs(x) < sc.half - 1e-3) & (np.abs(z) < sc.half - 1e-3)
    u = sc.uv_offset + sc.uv_scale * (x + sc.half) / (2 * sc.half)
    v = sc.uv_offset + sc.uv_scale * (z + sc.half) / (2 * sc.half)
    fu, fv = u - np.floor(u), v - np.floor(v)                          # glm::fract
    sure = on_floor
    texel = []
    for tex in (ta, te):
        h, w = tex.shape[:2]
        cx, cy = fu * w, fv * h
        # within 1e-4 of a texel border fp32 may fall either side
        sure = sure & (np.abs(cx - np.round(cx)) > 1e-4) & (np.abs(cy - np.round(cy)) > 1e-4)
        idx = np.clip(cy.astype(np.int64) * w + cx.astype(np.int64), 0, w * h - 1)
        texel.append(tex.reshape(-1, 3)[idx])
    expect = grade((albedo * texel[0] + emissive * texel[1]) * sky)
    assert sure.mean() > 0.5
    return sc, expect, sure


def test_textured_emissive_floor(torch_first):
    """Texture coordinates, their wrap-around (uv runs from -0.75 to 2.25), the texel addressing and the emissive
    term, against numpy: every pixel whose texel is not within rounding of a border must be the closed form."""
    sc, expect, sure = textured_floor()
    dev = ptamd.DeviceScene(sc.desc_ptr)
    try:
        p = ptamd.make_params(*TEX_RES, spp=1, max_bounces=6)
        img, _ = dev.render(sc.camera, p)
    finally:
        dev.close()
    err = np.abs(img - expect.astype(np.float32)).max(axis=2)
    assert float(err[sure].max()) < 2e-6, (int((err[sure] >= 2e-6).sum()), int(sure.sum()))


MODEL_RES = (200, 140)


def placed_floor():
    """The floor as an instance that is scaled, turned about two axes and moved (gpuRender.cu / intersection.hpp:150-157:
    the ray goes into the instance's space through inverse(model), the normal comes back through its transpose).
    Returns the scene, the expected frame -- grade(albedo * sky) where the ray meets the placed square, grade(sky)
    elsewhere -- the mask of the pixels safely inside or outside the silhouette, and the mask of those inside."""
    ca, sa, cb, sb = np.cos(0.6), np.sin(0.6), np.cos(-0.35), np.sin(-0.35)
    rot_z = np.array([[ca, -sa, 0, 0], [sa, ca, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    rot_x = np.array([[1, 0, 0, 0], [0, cb, -sb, 0], [0, sb, cb, 0], [0, 0, 0, 1.0]])
    scale = np.diag([1.5, 1.0, 0.6, 1.0])
    move = np.eye(4)
    move[:3, 3] = (0.7, -0.4, 1.1)
    model = move @ rot_z @ rot_x @ scale
    albedo, sky = np.array((0.6, 0.3, 0.8)), np.array((1.2, 0.9, 0.5))
    sc = FloorScene(albedo, sky, half=2.0, model=model)
    sc.camera.position[:] = (0.5, 7.0, -1.0)
    sc.camera.target[:] = (0.6, 0.0, 0.9)
    sc.camera.fov_deg = 55.0
    d = primary_directions(sc.camera, *MODEL_RES)
    pos = np.array(list(sc.camera.position) + [1.0], dtype=np.float64)
    o = sc.inv_model @ pos
    dl = d @ sc.inv_model[:3, :3].T
    t = -o[1] / dl[..., 1]
    x, z = o[0] + t * dl[..., 0], o[2] + t * dl[..., 2]
    edge = np.maximum(np.abs(x), np.abs(z)) - sc.half                  # < 0 inside the square
    inside = (t > 0) & (edge < 0)
    sure = np.abs(edge) > 1e-3
    expect = np.where(inside[..., None], grade(albedo * sky), grade(sky))
    assert 0.1 < inside.mean() < 0.85
    return sc, expect, sure, inside


def test_placed_floor_instance_transform(torch_first):
    sc, expect, sure, inside = placed_floor()
    dev = ptamd.DeviceScene(sc.desc_ptr)
    try:
        for spp, streams in ((1, 0), (6, 3)):
            p = ptamd.make_params(*MODEL_RES, spp=spp, max_bounces=6, seed_mode=ptamd.SEED_STREAM if streams else ptamd.SEED_REFERENCE_GPU,
                                  sub_streams=streams)
            img, _ = dev.render(sc.camera, p)
            err = np.abs(img - expect.astype(np.float32)).max(axis=2)
            if spp == 1:
                assert float(err[sure].max()) < 2e-6, (int((err[sure] >= 2e-6).sum()), int(sure.sum()))
            else:  # jittered samples: only pixels a whole pixel away from the silhouette are certain; the interior is
                from scipy import ndimage as ndi
                core = ndi.binary_erosion(inside, iterations=2) | ndi.binary_erosion(~inside, iterations=2)
                assert float(err[core].max()) < 2e-6
        # a wrong normal (matrix transposed, say) would send bounces back into the square: a second hit, albedo squared
        m = dev.measure(sc.camera, ptamd.make_params(*MODEL_RES, spp=1, max_bounces=6))
        assert m.hits == int(inside.sum()) or abs(int(m.hits) - int(inside.sum())) <= int((~sure).sum())
        assert m.misses == MODEL_RES[0] * MODEL_RES[1]                   # every path ends in the sky
    finally:
        dev.close()


MIRROR_RES = (180, 120)


def near_mirror_floor():
    """A metallic floor of roughness 1e-4 under a 6 x 3 environment of random colours, seen at angles from steep to
    grazing.  With alpha^2 = 1e-8 below fp32's epsilon the sampled half vector IS the normal (brdf.hpp:45-47:
    cosTheta = sqrt((1 - x) / (1 + (a^2 - 1) x)) = 1), so the bounce is the mirror direction, and
    the weight F * G * VdotH / (NdotV * NdotH) of gpuRender.cu:71-105 becomes F(NdotV) * G(NdotV, NdotV):
    Schlick's F0 + (1 - F0)(1 - c)^5 with F0 = mix(0.04, albedo, metalness) = albedo, and Smith's
    (c / (c (1 - k) + k))^2 with k = roughness / 2 (brdf.hpp:49-74).  Returns the scene, the expected frame at one
    sample per pixel and the mask of the floor pixels."""
    env_w, env_h = 6, 3
    env = np.random.default_rng(23).uniform(0.2, 2.5, size=(env_h, env_w, 3)).astype(np.float32)
    albedo, rough = np.array((0.9, 0.5, 0.2)), 1e-4
    sc = FloorScene(albedo, (1, 1, 1), metalness=1.0, roughness=rough, env=env, half=500.0)
    sc.camera.position[:] = (0.0, 1.0, 0.0)
    sc.camera.target[:] = (3.0, 0.2, 2.0)
    sc.camera.up[:] = (0.0, 1.0, 0.0)
    sc.camera.fov_deg = 70.0
    d = primary_directions(sc.camera, *MIRROR_RES)
    floor = d[..., 1] < -0.02
    c = np.clip(-d[..., 1], 1e-6, 1.0)[..., None]                       # NdotV = NdotL = VdotH
    F = albedo + (1.0 - albedo) * (1.0 - c) ** 5
    k = rough / 2.0
    G = (c / (c * (1.0 - k) + k)) ** 2
    mirrored = d * np.array([1.0, -1.0, 1.0])
    expect = grade(F * G * env.reshape(-1, 3)[environment_texels(mirrored, env_w, env_h)])
    assert 0.3 < floor.mean() < 0.9 and float(c[floor].min()) < 0.1 and float(c[floor].max()) > 0.6
    return sc, expect, floor


def check_near_mirror(img, expect, floor):
    err = np.abs(img - expect.astype(np.float32)).max(axis=2)[floor]
    # only a reflected direction within rounding of a texel border of the environment may differ
    print(f"near-mirror floor: median error {np.median(err):.1e}, 95th percentile {np.percentile(err, 95):.1e}, "
          f"beyond 1e-5: {int((err > 1e-5).sum())} of {err.size}")
    assert np.median(err) < 1e-6 and (err > 1e-5).sum() <= err.size // 1000


def test_near_mirror_floor_reflection_fresnel_and_smith(torch_first):
    sc, expect, floor = near_mirror_floor()
    dev = ptamd.DeviceScene(sc.desc_ptr)
    try:
        img, _ = dev.render(sc.camera, ptamd.make_params(*MIRROR_RES, spp=1, max_bounces=2))
    finally:
        dev.close()
    check_near_mirror(img, expect, floor)


LOBE_RES, LOBE_SPP = (16, 12), 1024
LOBE_CASES = [(0.0, 1.0), (1.0, 0.5), (1.0, 0.2), (0.5, 0.7)]            # (metalness, roughness)


def lobe_scene(metalness, roughness):
    """The floor under a 16 x 8 environment of random colours, seen through a 1-degree field of view (one direction V
    for the whole frame) and the mean radiance the estimator of gpuRender.cu:60-107 must converge to: the integral
    over the upper hemisphere of   albedo * cos / pi * L   for the Lambert branch (metalness 0: cosine-weighted
    sampling, weight albedo) and of   D F G / (4 NdotV) * L   for the GGX branch (half vector sampled from
    D * NdotH with alpha = roughness, reflected; weight F G VdotH / (NdotV NdotH); no diffuse part whatever the
    metalness, F0 = mix(0.04, albedo, metalness)), by quadrature in numpy.  This is the one check of the sampling
    DISTRIBUTIONS and of the tangent frame: the other closed forms hold for any distribution."""
    env_w, env_h = 16, 8
    env = np.random.default_rng(31).uniform(0.1, 3.0, size=(env_h, env_w, 3)).astype(np.float32)
    albedo = np.array((0.9, 0.6, 0.4))
    sc = FloorScene(albedo, (1, 1, 1), metalness=metalness, roughness=roughness, env=env, half=500.0)
    sc.camera.position[:] = (0.0, 4.0, 0.0)
    sc.camera.target[:] = (2.0, 0.0, 1.0)
    sc.camera.up[:] = (0.0, 1.0, 0.0)
    sc.camera.fov_deg = 1.0
    nt, nph = 1024, 4096
    th, ph = np.meshgrid((np.arange(nt) + 0.5) * (np.pi / 2 / nt), (np.arange(nph) + 0.5) * (2 * np.pi / nph), indexing="ij")
    l = np.stack([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)], axis=-1)
    dw = np.sin(th) * (np.pi / 2 / nt) * (2 * np.pi / nph)
    L = env.reshape(-1, 3)[environment_texels(l, env_w, env_h)]
    if metalness == 0.0:
        expect = albedo * (L * (l[..., 1:2] / np.pi) * dw[..., None]).sum(axis=(0, 1))
    else:
        V = -primary_directions(sc.camera, *LOBE_RES)[LOBE_RES[1] // 2, LOBE_RES[0] // 2]
        F0 = 0.04 * (1.0 - metalness) + albedo * metalness
        h = V + l
        h /= np.linalg.norm(h, axis=-1, keepdims=True)
        a, k = roughness, roughness / 2.0
        D = a * a / (np.pi * ((h[..., 1] ** 2) * (a * a - 1.0) + 1.0) ** 2)
        F = F0 + (1.0 - F0) * ((1.0 - (h * V).sum(-1)) ** 5)[..., None]
        G = (V[1] / (V[1] * (1.0 - k) + k)) * (l[..., 1] / (l[..., 1] * (1.0 - k) + k))
        expect = ((D * G / (4.0 * V[1]) * dw)[..., None] * F * L).sum(axis=(0, 1))
    return sc, expect


@pytest.mark.parametrize("metalness,roughness", LOBE_CASES)
def test_sampled_mean_is_the_integral_of_the_lobe(torch_first, metalness, roughness):
    import torch
    sc, expect = lobe_scene(metalness, roughness)
    dev = ptamd.DeviceScene(sc.desc_ptr)
    try:
        for seed_mode, streams in ((ptamd.SEED_REFERENCE_GPU, 0), (ptamd.SEED_STREAM, 8)):
            p = ptamd.make_params(*LOBE_RES, spp=LOBE_SPP, max_bounces=2, seed_mode=seed_mode, sub_streams=streams)
            acc = torch.zeros((LOBE_RES[1], LOBE_RES[0], 4), dtype=torch.float32, device="cuda")
            dev.accumulate_device(sc.camera, p, acc.data_ptr(), None)
            torch.cuda.synchronize()
            a = acc.cpu().numpy().astype(np.float64)
            got = a[..., :3].sum(axis=(0, 1)) / a[..., 3].sum()
            print(f"metalness {metalness} roughness {roughness} streams {streams}: sampled / integral = {got / expect}")
            # 196,608 samples: the statistical error is about 0.2 %
            assert np.abs(got / expect - 1.0).max() < 0.01
    finally:
        dev.close()


def bumpy_mirror_floor():
    """The exact-mirror floor again, with a normal per corner that leans outwards: the shading normal is the normalised
    barycentric mix  b.x * n1 + b.y * n2 + (1 - b.x - b.y) * n0  (geometry.cpp:69-87, b from Moeller-Trumbore,
    intersection.hpp:9-44), bilinear in the hit point on each of the two triangles.  Returns scene, expected frame and
    the mask of the pixels that are certain (on the floor, off the diagonal, reflected ray leaving upwards)."""
    env_w, env_h = 6, 3
    env = np.random.default_rng(29).uniform(0.2, 2.5, size=(env_h, env_w, 3)).astype(np.float32)
    albedo, rough, half = np.array((0.8, 0.6, 0.3)), 1e-4, 3.0
    lean = lambda x, z: (0.5 * x / half, 1.3, -0.35 * z / half)
    sc = FloorScene(albedo, (1, 1, 1), metalness=1.0, roughness=rough, env=env, half=half, corner_normal=lean)
    sc.camera.position[:] = (0.5, 4.0, -3.5)
    sc.camera.target[:] = (0.0, 0.0, 0.3)
    sc.camera.up[:] = (0.0, 1.0, 0.0)
    sc.camera.fov_deg = 60.0
    d = primary_directions(sc.camera, *MIRROR_RES)
    pos = np.array(list(sc.camera.position), dtype=np.float64)
    t = -pos[1] / d[..., 1]
    x, z = pos[0] + t * d[..., 0], pos[2] + t * d[..., 2]
    corner = lambda cx, cz: np.array(lean(cx, cz), dtype=np.float64)
    n00, n01, n11, n10 = corner(-half, -half), corner(-half, half), corner(half, half), corner(half, -half)
    upper = z >= x                                                      # triangle 0: (-h,-h) (-h,h) (h,h); 1: (-h,-h) (h,h) (h,-h)
    bx = np.where(upper, (z - x) / (2 * half), (z + half) / (2 * half))[..., None]
    by = np.where(upper, (x + half) / (2 * half), (x - z) / (2 * half))[..., None]
    n1 = np.where(upper[..., None], n01, n11)
    n2 = np.where(upper[..., None], n11, n10)
    N = bx * n1 + by * n2 + (1.0 - bx - by) * n00
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    c = -(d * N).sum(-1, keepdims=True)
    r = d + 2.0 * c * N
    sure = (d[..., 1] < 0) & (np.maximum(np.abs(x), np.abs(z)) < half - 1e-3) & (np.abs(z - x) > 1e-3) & (c[..., 0] > 0.05) & (r[..., 1] > 0.05)
    c = np.clip(c, 1e-6, 1.0)
    k = rough / 2.0
    expect = grade((albedo + (1.0 - albedo) * (1.0 - c) ** 5) * (c / (c * (1.0 - k) + k)) ** 2
                   * env.reshape(-1, 3)[environment_texels(r, env_w, env_h)])
    assert sure.mean() > 0.2
    return sc, expect, sure


def test_interpolated_normals_on_a_mirror_floor(torch_first):
    sc, expect, sure = bumpy_mirror_floor()
    dev = ptamd.DeviceScene(sc.desc_ptr)
    try:
        img, _ = dev.render(sc.camera, ptamd.make_params(*MIRROR_RES, spp=1, max_bounces=2))
    finally:
        dev.close()
    check_near_mirror(img, expect, sure)


class BoxScene(FloorScene):
    """A closed cube [-1, 1]^3 around the camera: twelve triangles in one BLAS leaf under one TLAS leaf."""

    def __init__(self, albedo, emissive, sky):
        super().__init__(albedo, sky, emissive=emissive)
        corners = lambda axis, side: [tuple(side if k == axis else (a if k == (axis + 1) % 3 else b) for k in range(3))
                                      for a, b in ((-1.0, -1.0), (1.0, -1.0), (1.0, 1.0), (-1.0, 1.0))]
        self.tris = (ptamd.Triangle * 12)()
        normals = []
        i = 0
        for axis in range(3):
            for side in (-1.0, 1.0):
                q = corners(axis, side)
                for tri in ((q[0], q[1], q[2]), (q[0], q[2], q[3])):
                    self.tris[i].v0[:], self.tris[i].v1[:], self.tris[i].v2[:] = tri
                    normals += [side if k == axis else 0.0 for k in range(3)]
                    i += 1
        self.face_normals = (C.c_float * 36)(*normals)
        for n, cnt in ((self.tlas[0], 1), (self.blas[0], 12)):
            n.aabb_min[:] = (-1.0, -1.0, -1.0)
            n.aabb_max[:] = (1.0, 1.0, 1.0)
            n.left_first, n.count = 0, cnt
        self.geom[0].tri_count = 12
        d = self.desc
        d.triangles, d.triangle_total = self.tris, 12
        d.face_normals, d.face_normal_total = self.face_normals, 12
        self.camera.position[:] = (0.2, -0.1, 0.3)
        self.camera.target[:] = (0.9, 0.55, -0.7)
        self.camera.up[:] = (0.0, 1.0, 0.0)
        self.camera.fov_deg = 100.0


FURNACE = ((0.8, 0.6, 0.9), (0.1, 0.3, 0.05), (7.0, 7.0, 7.0))          # albedo, emissive, a sky nobody may see


def check_furnace(render, measure=None):
    """Inside a closed box no path ever misses: after B bounces the loop ends and returns the throughput
    (gpuRender.cu:27-29), the product of (albedo + emissive) over B hits, whatever was sampled: grade((a + e)^B).
    Checks the bounce loop, its count, and that a bounce leaves from the near side of the wall (p + 1e-4 N with N
    turned against the ray, gpuRender.cu:52-55 and :109)."""
    a, e = np.array(FURNACE[0]), np.array(FURNACE[1])
    for bounces, spp in ((1, 1), (2, 3), (4, 2), (6, 4)):
        p = ptamd.make_params(120, 80, spp=spp, max_bounces=bounces)
        img = render(p)
        err = np.abs(img - grade((a + e) ** bounces).astype(np.float32)).max(axis=2)
        # Moeller-Trumbore is not watertight: a ray through an edge of the cube may slip out and see the sky
        assert (err > 2e-6).sum() <= 2, (bounces, int((err > 2e-6).sum()), float(err.max()))
        if measure is not None:
            m = measure(p)
            assert abs(int(m.hits) - 120 * 80 * spp * bounces) <= 2 * bounces and m.misses <= 2


def test_furnace_box(torch_first):
    sc = BoxScene(*FURNACE)
    dev = ptamd.DeviceScene(sc.desc_ptr)
    try:
        check_furnace(lambda p: dev.render(sc.camera, p)[0], lambda p: dev.measure(sc.camera, p))
    finally:
        dev.close()
