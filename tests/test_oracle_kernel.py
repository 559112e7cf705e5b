"""The oracle against first principles and against the reference's documented quirks
(SURVEY.md Appendix A).  CPU only.  The reference has no tests or golden vectors for this path
(parity unpinned), so these are known-answer checks derived from the kernel sources."""
import ctypes

import numpy as np
import pytest

from conftest import same_values

INF = np.float32(np.inf)


def brute_force_hits(arr, rays):
    """Möller-Trumbore distance of every ray to EVERY face (float64, no BVH): (n_rays, n_faces), inf = miss."""
    fv = arr["facesV"]
    v = arr["vertices"][:, :3].astype(np.float64)
    a, b, c = v[fv[:, 0]], v[fv[:, 1]], v[fv[:, 2]]
    e1, e2 = b - a, c - a
    out = np.full((len(rays), len(fv)), np.inf)
    for i, r in enumerate(rays.astype(np.float64)):
        o, d = r[:3], r[3:]
        p = np.cross(d, e2)
        det = np.einsum("ij,ij->i", e1, p)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            tv = o - a
            u = np.einsum("ij,ij->i", tv, p) * inv
            q = np.cross(tv, e1)
            w = (q @ d) * inv
            t = np.einsum("ij,ij->i", e2, q) * inv
        ok = (u >= 0) & (w >= 0) & (u + w <= 1) & (t >= 1e-5) & np.isfinite(t)
        out[i, ok] = t[ok]
    return out


def random_rays(rng, n, lo, hi):
    o = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


@pytest.mark.parametrize("kind,triangles,skip", [("cornell", 0, True), ("cornell", 0, False), ("sponza", 6000, True), ("hairball", 3000, True)])
def test_traversal_agrees_with_brute_force(cfg_defaults, oracle, kind, triangles, skip):
    """The stackless walk over the flattened, skip-ahead-compacted BVH (pt_bvh.cl:82-123 over
    PathTracer.cpp:238-347) finds the same closest face as testing every triangle."""
    pbr = cfg_defaults
    pbr.cfg_set(**{"bvh.skip_ahead": skip})
    sc = pbr.HostScene.generate(kind, 2, triangles)
    cfg = sc.config(64, 64)
    arr = sc.arrays()
    lo = arr["vertices"][:, :3].min(0) - 0.2
    hi = arr["vertices"][:, :3].max(0) + 0.2
    rays = random_rays(np.random.default_rng(5), 300, lo, hi)
    t, face, normal, counts = oracle.trace_rays(sc.desc, cfg, rays)
    all_t = brute_force_hits(arr, rays)
    bt = all_t.min(axis=1)

    hit = np.isfinite(bt)
    # rays that graze an edge may be accepted by one formulation and not the other
    assert np.mean(np.isfinite(t[hit])) > 0.995
    assert np.mean(np.isfinite(t[~hit])) < 0.005 if (~hit).any() else True
    both = hit & np.isfinite(t)
    assert np.allclose(t[both], bt[both], rtol=2e-4, atol=2e-5)
    # the reported face is A closest face (coplanar / shared-edge faces tie at equal t)
    t_of_reported = all_t[np.nonzero(both)[0], face[both]]
    assert np.allclose(t_of_reported, bt[both], rtol=2e-4, atol=2e-5)
    # the unit geometric normal of the hit face
    n = normal[both]
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-5)
    assert (counts[:, 0] >= 1).all()


def _single_leaf_scene(pbr):
    """Hand-made flat arrays: root + one leaf holding one triangle in the z = 0 plane."""
    bvh = np.array([
        [-1, -1, -1, -1, 1, 1, 1, -1],       # root (never tested)
        [-1, -1, -0.1, 0, 1, 1, 0.1, -1],    # leaf: face 0
    ], np.float32)
    facesV = np.array([[0, 1, 2, 0]], np.uint32)
    verts = np.array([[-1, -1, 0, 0], [1, -1, 0, 0], [0, 1, 0, 0]], np.float32)
    mats = np.zeros((1, 16), np.float32)
    mats[0, :6] = [1, 1, 0, 0, 0, 1]
    mats[0, 8:11] = 0.5
    mats[0, 12:15] = 1.0
    lights = np.zeros((1, 12), np.float32)
    keep = (bvh, facesV, verts, mats, lights)
    desc = pbr.SceneDesc()
    desc.bvh, desc.num_nodes = bvh.ctypes.data, 2
    desc.facesV, desc.num_faces = facesV.ctypes.data, 1
    desc.vertices, desc.num_vertices = verts.ctypes.data, 3
    desc.materials, desc.num_materials, desc.brdf = mats.ctypes.data, 1, 1
    desc.lights, desc.num_lights = lights.ctypes.data, 0
    cfg = pbr.Config()
    cfg.width = cfg.height = 8
    cfg.brdf, cfg.max_depth, cfg.max_added_depth, cfg.samples = 1, 3, 5, 1
    cfg.anti_aliasing = 0.7
    cfg.sky_light[0], cfg.sky_light[1], cfg.sky_light[2] = 0.25, 0.5, 0.75
    cfg.tile_world, cfg.tile_rank = 1, 0
    return desc, cfg, keep


def test_triangle_hit_distance_known_answer(pbr, oracle):
    desc, cfg, keep = _single_leaf_scene(pbr)
    rays = np.array([[0, 0, 2, 0, 0, -1], [0, 0, -3, 0, 0, 1], [5, 5, 2, 0, 0, -1], [0, 0, 2, 0, 0, 1]], np.float32)
    t, face, normal, counts = oracle.trace_rays(desc, cfg, rays)
    assert t[0] == pytest.approx(2.0, abs=1e-6) and face[0] == 0
    assert t[1] == pytest.approx(3.0, abs=1e-6)
    assert t[2] == INF and t[3] == INF
    assert np.allclose(np.abs(normal[0]), [0, 0, 1])
    assert counts[0, 0] == 1 and counts[0, 1] == 1      # one node visit, one face test
    assert counts[2, 1] == 0                             # box missed: no face test


def test_root_leaf_scene_renders_nothing(pbr, oracle):
    """Appendix A.1: traversal starts at node 1; with a single (root) node the walk never tests
    a face.  The C ABI rejects such scenes; the oracle shows the reference behaviour."""
    desc, cfg, keep = _single_leaf_scene(pbr)
    bvh = keep[0].copy()
    bvh[1] = [9, 9, 9, -1, 10, 10, 10, -1]   # node 1: empty container far away, miss link -1
    desc.bvh = bvh.ctypes.data
    t, face, _, counts = oracle.trace_rays(desc, cfg, np.array([[0, 0, 2, 0, 0, -1]], np.float32))
    assert t[0] == INF and counts[0, 0] == 1 and counts[0, 1] == 0


def test_escaping_paths_pick_up_sky_light_exactly(pbr, oracle):
    """Every camera ray of this setup misses: pixel = SKY_LIGHT, .w = INF (first-hit distance)."""
    desc, cfg, keep = _single_leaf_scene(pbr)
    cam = pbr.Camera()
    cam.eye.x, cam.eye.y, cam.eye.z = 0, 0, 50
    cam.w.x, cam.w.y, cam.w.z = 0, 0, 1          # looking AWAY from the triangle
    cam.u.x, cam.v.y = 1, 1
    cam.focusPoint[0] = cam.focusPoint[1] = -1
    r = oracle.Renderer(desc, cfg)
    img = r.render_frame(0.0333, 0.0, 0.01, cam)
    assert np.all(img[..., 0] == np.float32(0.25)) and np.all(img[..., 1] == np.float32(0.5)) and np.all(img[..., 2] == np.float32(0.75))
    assert np.all(np.isinf(img[..., 3]))
    assert r.counter_dict()["paths"] == 64 and r.counter_dict()["hits"] == 0


def test_running_mean_matches_setcolors(pbr, oracle):
    """setColors (pt_rgb.cl:9-21): out = mix( new, previous, pixelWeight ); .w is replaced."""
    desc, cfg, keep = _single_leaf_scene(pbr)
    cam = pbr.Camera()
    cam.eye.z = 50
    cam.w.z, cam.u.x, cam.v.y = 1, 1, 1
    cam.focusPoint[0] = cam.focusPoint[1] = -1
    r = oracle.Renderer(desc, cfg)
    r.image[:] = np.float32(2.0)
    img = r.render_frame(1.0, 0.75, 0.01, cam)
    new = np.array([0.25, 0.5, 0.75], np.float32)
    expect = new + (np.float32(2.0) - new) * np.float32(0.75)
    assert np.all(img[..., :3] == expect)


def test_depth_exhausted_paths_contribute_zero(cfg_defaults, oracle):
    """Appendix A.9: a path that runs out of depth inside the scene adds nothing; with
    max_depth = 1 a diffuse first hit ends the path black (pathtracing.cl:274-276)."""
    pbr = cfg_defaults
    pbr.cfg_set(**{"render.max_depth": 1, "render.max_added_depth": 0})
    sc = pbr.HostScene.generate("cornell")
    cfg, cam = sc.config(32, 32), sc.camera()
    r = oracle.Renderer(sc.desc, cfg)
    img = r.render_frame(0.0333, 0.0, pbr.pixel_dimension(32, 32), cam)
    hit = np.isfinite(img[..., 3])
    diffuse_black = (img[..., :3][hit] == 0).all(axis=1)
    assert hit.mean() > 0.5
    # everything that hit an opaque, non-extending material is exactly black; glossy/glass may extend
    assert diffuse_black.mean() > 0.6
    sky = ~hit
    assert np.allclose(img[..., :3][sky], [cfg.sky_light[0], cfg.sky_light[1], cfg.sky_light[2]])


def _new_ray_seed_advance(oracle, brdf, mat_floats, n=64):
    rng = np.random.default_rng(3)
    inp = np.zeros((n, 12), np.float32)
    d = rng.normal(size=(n, 3))
    d[:, 2] = -np.abs(d[:, 2]) - 0.2
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    inp[:, 0:3] = rng.uniform(-1, 1, (n, 3))
    inp[:, 3:6] = d
    inp[:, 6:9] = [0, 0, 1]
    inp[:, 9] = 2.0
    inp[:, 10] = rng.integers(0, 50, n) + 0.5      # exactly representable: seed + 1 stays exact
    mtl = np.asarray(mat_floats, np.float32)
    out = np.empty((n, 8), np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    oracle.lib().orc_new_ray(brdf, mtl.ctypes.data, inp.ctypes.data_as(fp), n, out.ctypes.data_as(fp))
    return inp, out


def test_rng_stream_order_shirley_ashikhmin(oracle):
    """Appendix A.8: opaque S-A material draws exactly 3 numbers in getNewRay (a, b, and the
    always-evaluated diffuse fallback); d < 1 draws one more first."""
    opaque = [1, 1, 10, 10, 0.3, 1, 0, 0, .5, .5, .5, 0, 1, 1, 1, 0]
    inp, out = _new_ray_seed_advance(oracle, 1, opaque)
    assert np.all(out[:, 6] - inp[:, 10] == 3.0)
    assert np.allclose(out[:, 0:3], inp[:, 0:3] + 2.0 * inp[:, 3:6], atol=1e-6)      # origin = o + t*dir
    assert np.allclose(np.linalg.norm(out[:, 3:6], axis=1), 1.0, atol=1e-5)
    assert (out[:, 5] >= -1e-6).all()                                                  # stays in the upper hemisphere
    assert (out[:, 7] == 0).all()

    # d = 0.5: one draw decides transmission; refract draws 1 more (2 total, or 1 on total
    # internal reflection); otherwise the 3 S-A draws follow (4 total)
    glassy = [0.5, 1.5, 10, 10, 0.3, 1, 0, 0, .5, .5, .5, 0, 1, 1, 1, 0]
    inp, out = _new_ray_seed_advance(oracle, 1, glassy, 256)
    adv = out[:, 6] - inp[:, 10]
    assert set(np.unique(adv)).issubset({1.0, 2.0, 4.0})
    assert ((adv <= 2.0) == (out[:, 7] == 1.0)).all()                                  # refraction path sets addDepth


def test_rng_stream_order_schlick(oracle):
    mirror = [1, 1, 1, 0, .5, .5, .5, 0, 1, 1, 1, 0]         # rough = 0: perfect reflection, no draws
    inp, out = _new_ray_seed_advance(oracle, 0, mirror)
    assert np.all(out[:, 6] == inp[:, 10])
    refl = inp[:, 3:6] - 2 * inp[:, 5:6] * np.array([0, 0, 1], np.float32)
    assert np.allclose(out[:, 3:6], refl, atol=1e-6)
    rough = [1, 1, 1, 0.6, .5, .5, .5, 0, 1, 1, 1, 0]        # 2 draws (+1 if the lobe sample dips below)
    inp, out = _new_ray_seed_advance(oracle, 0, rough, 256)
    assert set(np.unique(out[:, 6] - inp[:, 10])).issubset({2.0, 3.0})


def test_orb_light_quirk_radius_not_squared(cfg_defaults, oracle, pbr):
    """Appendix A.11: intersectSphere compares d2 with r, not r*r (pt_intersect.cl:53-57): a ray
    passing an orb of radius 0.25 at distance 0.4 (0.16 <= 0.25) still 'hits' it."""
    desc, cfg, keep = _single_leaf_scene(pbr)
    lights = np.zeros((1, 12), np.float32)
    lights[0, 0:3] = [0.4, 0, 10]
    lights[0, 4:7] = [3, 2, 1]
    lights[0, 8:10] = [2, 0.25]
    desc.lights, desc.num_lights = lights.ctypes.data, 1
    cam = pbr.Camera()
    cam.eye.z = 5
    cam.w.z, cam.u.x, cam.v.y = 1, 1, 1     # looking +z, away from the triangle, past the orb
    cam.focusPoint[0] = cam.focusPoint[1] = -1
    cfg.anti_aliasing = 0.0
    r = oracle.Renderer(desc, cfg)
    img = r.render_frame(0.5, 0.0, 1e-4, cam)
    assert np.all(img[..., :3] == np.array([3, 2, 1], np.float32))


def test_shadow_rays_change_only_lit_scenes(cfg_defaults, oracle):
    """SHADOW_RAYS needs lights at load time (ObjParser.cpp:133); the generated scenes have none,
    so switching it on must not change a single bit."""
    pbr = cfg_defaults
    sc = pbr.HostScene.generate("cornell")
    cam, px = sc.camera(), pbr.pixel_dimension(32, 32)
    imgs = []
    for shadow in (0, 1):
        cfg = sc.config(32, 32)
        cfg.shadow_rays = shadow
        imgs.append(oracle.Renderer(sc.desc, cfg).render(0, pbr.frame_seeds(0, 2), px, cam))
    assert same_values(imgs[0], imgs[1])


def test_threads_do_not_change_results(cfg_defaults, oracle):
    pbr = cfg_defaults
    sc = pbr.HostScene.generate("cornell")
    cfg, cam, px = sc.config(48, 40), sc.camera(), pbr.pixel_dimension(48, 40)
    a = oracle.Renderer(sc.desc, cfg, threads=1).render(0, pbr.frame_seeds(0, 2), px, cam)
    b = oracle.Renderer(sc.desc, cfg, threads=4).render(0, pbr.frame_seeds(0, 2), px, cam)
    assert same_values(a, b)


@pytest.mark.parametrize("brdf", [1, 0])
def test_the_baseline_legs_native_build_gives_the_same_bits(cfg_defaults, oracle, brdf):
    """bench.py times the oracle built -O3 -march=native on the machine it runs on (oracle.build_native);
    the tests check everything against the portable -O2 -mavx2 build.  -ffp-contract=off in both: identical images,
    debug images and counters — the flags buy time, never bits."""
    pbr = cfg_defaults
    pbr.cfg_set(**{"render.brdf": brdf, "render.max_depth": 4})
    sc = pbr.HostScene.generate("sponza", 2, 3000)
    cfg, cam, px = sc.config(64, 48), sc.camera(), pbr.pixel_dimension(64, 48)
    portable = oracle.Renderer(sc.desc, cfg, threads=2)
    native = oracle.Renderer(sc.desc, cfg, threads=2, native=True)
    assert native.native and not portable.native and "native" in oracle.build_native()
    a = portable.render(0, pbr.frame_seeds(0, 3), px, cam)
    b = native.render(0, pbr.frame_seeds(0, 3), px, cam)
    assert same_values(a, b) and same_values(portable.debug, native.debug)
    assert portable.counter_dict() == native.counter_dict()


# ----------------------------------------------------------------------------------------------
# fixtures of the reference's own scenes (tests/golden/ref_*.npz, make_reference_scenes.py)
# ----------------------------------------------------------------------------------------------

import os  # noqa: E402
import sys  # noqa: E402

from conftest import REFERENCE_MODELS, ROOT  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_reference_scenes  # noqa: E402


@pytest.mark.parametrize("name", sorted(make_reference_scenes.CASES))
def test_oracle_reproduces_the_reference_scene_fixtures(pbr, oracle, name):
    """The committed fixtures are self-contained (arrays + constants + camera + seeds): the oracle, fed from them
    alone, gives the stored image / debug image / counters / ray batch — on this container and on the GPU box."""
    data = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    desc, cfg, cam, keep = make_reference_scenes.scene_from_fixture(pbr, data)      # `keep` owns the arrays desc points at
    assert pbr.validate_scene(desc) == ""
    out = make_reference_scenes.outputs_of(pbr, oracle, data)
    for key in ("image", "debug", "counters", "rays", "ray_t", "ray_counts"):
        assert np.array_equal(out[key], data[key], equal_nan=True), key
    # what makes these scenes worth having: glass, extreme lobes, lights
    mats = data["materials"]
    if "pillars" in name or "spheres" in name:
        assert (mats[:, 0] < 1.0).any()                                          # d < 1: refraction on whole images
    if "suzanne" in name and mats.shape[1] == 16:
        assert (mats[:, 2] == 100000.0).any()                                    # nu = nv = 1e5
    assert (data["lights"].shape[0] == 1) == ("shadow" in name)


@pytest.mark.skipif(not os.path.isdir(REFERENCE_MODELS), reason="reference assets only exist in the build container")
@pytest.mark.parametrize("name", sorted(make_reference_scenes.CASES))
def test_loader_and_builder_reproduce_the_fixture_inputs(pbr, name):
    """Container only: the reference's .obj / .mtl / .lights through host/model_io.cpp + host/bvh_builder.cpp still
    give the arrays the fixtures hold."""
    data = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    sc, cfg, cam, px, frames = make_reference_scenes.load_case(pbr, name)
    fresh = make_reference_scenes.inputs_of(pbr, sc, cfg, cam, px, frames)
    for key, value in fresh.items():
        assert np.array_equal(np.asarray(value), data[key]), key
    pbr.cfg_reset()
