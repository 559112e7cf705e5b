"""Known answers ON THE HIP PATH that do not route through oracle/pt_oracle.c: what the kernel source of the
reference implies for inputs simple enough to work out by hand.  (The oracle is a restatement of the same source;
these tests would still hold the HIP kernels to the reference if the restatement were wrong.)"""
import numpy as np
import pytest

from conftest import same_values

pytestmark = pytest.mark.gpu


@pytest.fixture()
def device(pbr, gpu_device):
    dev = pbr.Device(gpu_device)
    yield dev
    dev.close()


def one_triangle_scene(pbr, brdf=1):
    """Root + one leaf, one triangle (0,0,0) (1,0,0) (0,1,0) in the plane z = 0, one matte material."""
    verts = np.array([[0, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0]], np.float32)
    normals = np.array([[0, 0, 1, 0]], np.float32)
    facesV = np.array([[0, 1, 2, 0]], np.uint32)
    facesN = np.array([[0, 0, 0, 0]], np.uint32)
    bvh = np.array([[0, 0, 0, -1, 1, 1, 0, -1],                  # root: container, never tested (pt_bvh.cl:84)
                    [0, 0, 0, 0, 1, 1, 0, -1]], np.float32)      # node 1: leaf, face 0, no second face
    if brdf == 1:
        mats = np.array([[1, 1, 0, 0, 0, 1, 0, 0, .5, .25, .125, 0, 1, 1, 1, 0]], np.float32)
    else:
        mats = np.array([[1, 1, 1, 1, .5, .25, .125, 0, 1, 1, 1, 0]], np.float32)
    d = pbr.SceneDesc()
    d.bvh, d.num_nodes = bvh.ctypes.data, 2
    d.facesV, d.facesN, d.num_faces = facesV.ctypes.data, facesN.ctypes.data, 1
    d.vertices, d.num_vertices = verts.ctypes.data, 3
    d.normals, d.num_normals = normals.ctypes.data, 1
    d.materials, d.num_materials, d.brdf = mats.ctypes.data, 1, brdf
    d.lights, d.num_lights = None, 0
    return d, (verts, normals, facesV, facesN, bvh, mats)


def plain_config(pbr, w, h, brdf=1, depth=4, added=0, sky=(0.75, 0.5, 0.25)):
    c = pbr.Config()
    c.width, c.height, c.brdf, c.shadow_rays = w, h, brdf, 0
    c.max_depth, c.max_added_depth, c.samples = depth, added, 1
    c.anti_aliasing, c.phong_tessellation = 0.0, 0.0
    c.sky_light[0], c.sky_light[1], c.sky_light[2], c.sky_light[3] = sky[0], sky[1], sky[2], 0.0
    c.tile_world, c.tile_rank = 1, 0
    return c


def look(pbr, eye, center):
    cam = pbr.Camera()
    e = np.asarray(eye, np.float32)
    c = np.asarray(center, np.float32)
    pbr.host.pbrh_camera_lookat(e.ctypes.data_as(pbr._fp), c.ctypes.data_as(pbr._fp), cam)
    return cam


@pytest.mark.parametrize("traversal", [0, 1, 2])
def test_single_triangle_distances_by_hand(pbr, device, traversal):
    """flatTriAndRayIntersect (pt_intersect.cl:92-129) through the stackless walk: rays straight down on the triangle
    hit at their height, rays beside it or from below its plane's far side in the wrong direction miss; one node visit
    and one face test each (node 1 is always visited, pt_bvh.cl:84-88)."""
    desc, keep = one_triangle_scene(pbr)
    device.upload_scene(desc)
    cfg = plain_config(pbr, 8, 8)
    cfg.traversal = traversal                        # the ray-ordered walks (round 5) over a tree of one leaf: the same known answers
    device.configure(cfg)
    heights = np.array([0.5, 1.0, 2.0, 3.0, 8.0, 64.0], np.float32)
    rays = []
    for z in heights:
        rays.append([0.25, 0.25, z, 0, 0, -1])       # inside the triangle
        rays.append([0.75, 0.75, z, 0, 0, -1])       # inside the leaf's box, outside the triangle (u + v > 1)
        rays.append([0.25, 0.25, z, 0, 0, 1])        # pointing away
        rays.append([2.5, 0.25, z, 0, 0, -1])        # beside the box
    rays.append([0.25, 0.25, -2.0, 0, 0, 1])         # from behind: the test is two-sided (no culling in the reference)
    rays = np.array(rays, np.float32)
    t, face, normal, counts = device.diag_trace(rays)
    for k, z in enumerate(heights):
        hit, edge, away, beside = t[4 * k: 4 * k + 4]
        assert abs(float(hit) - float(z)) <= 4e-7 * float(z), (z, hit)    # t = ( t - f ) + f with f = tNear - 0.001: a few ulp
        assert face[4 * k] == 0 and same_values(normal[4 * k], [0, 0, 1])
        assert np.isinf(edge) and np.isinf(away) and np.isinf(beside)
        assert counts[4 * k].tolist() == [1, 1]                            # node 1, face 0
        assert counts[4 * k + 1].tolist() == [1, 1]                        # box hit, face tested and rejected
        assert counts[4 * k + 2].tolist() == [1, 0] and counts[4 * k + 3].tolist() == [1, 0]
    assert abs(float(t[-1]) - 2.0) <= 1e-6 and face[-1] == 0


@pytest.mark.parametrize("traversal,arith", [(0, 0), (2, 0), (0, 1), (2, 1)])
@pytest.mark.parametrize("brdf", [1, 0])
@pytest.mark.parametrize("frames", [1, 5])
def test_sky_only_frames_are_the_sky_colour_exactly(pbr, device, brdf, frames, traversal, arith):
    """A camera that sees no geometry: every path leaves at depth 0 with color = 1, so finalColor = SKY_LIGHT
    (pathtracing.cl:263-266,320-323), secondaryPaths = 1, and the running mean of equal frames is that value
    (pt_rgb.cl:9-21: c + ( prev - c ) * w with prev == c); .w = the first-hit distance = INFINITY."""
    desc, keep = one_triangle_scene(pbr, brdf)
    sky = (0.75, 0.5, 0.25)
    cfg = plain_config(pbr, 64, 40, brdf=brdf, sky=sky)
    cfg.anti_aliasing = 1.0
    cfg.traversal, cfg.arith = traversal, arith      # exact in every mode: no arithmetic touches the colour ( x * rcp( 1 ) = x )
    cam = look(pbr, (0.3, 0.3, 5.0), (0.3, 0.3, 0.0))                        # the view direction is eye - center (Camera.cpp): away from the triangle
    device.upload_scene(desc)
    device.configure(cfg)
    device.render(0, pbr.frame_seeds(0, frames), pbr.pixel_dimension(64, 40), cam)
    img = device.read_output()
    assert same_values(img[..., :3], np.broadcast_to(np.asarray(sky, np.float32), img[..., :3].shape))
    assert np.isinf(img[..., 3]).all()
    c = device.counters()
    assert c == {"nodes": 64 * 40 * frames, "tris": 0, "hits": 0, "paths": 64 * 40 * frames}


@pytest.mark.parametrize("traversal,arith", [(0, 0), (2, 0), (0, 1), (2, 1)])
@pytest.mark.parametrize("brdf", [1, 0])
def test_depth_exhausted_paths_contribute_nothing(pbr, device, brdf, traversal, arith):
    """MAX_DEPTH = 1, no added depth: a path whose first hit is an opaque surface ends there with no contribution
    (pathtracing.cl:274-276) — the pixel is exactly 0 with a finite first-hit distance — and a path that misses is the
    sky.  (BRDF 0 draws extendDepth's random number first, pt_utils.cl:89-96: rough = 1 never extends.)"""
    desc, keep = one_triangle_scene(pbr, brdf)
    sky = (0.75, 0.5, 0.25)
    w, h = 96, 64
    cfg = plain_config(pbr, w, h, brdf=brdf, depth=1, added=0, sky=sky)
    cfg.traversal, cfg.arith = traversal, arith
    cam = look(pbr, (0.3, 0.3, 2.0), (0.3, 0.3, 4.0))                        # straight down on the triangle (view direction = eye - center)
    device.upload_scene(desc)
    device.configure(cfg)
    device.render(0, pbr.frame_seeds(0, 3), pbr.pixel_dimension(w, h), cam)
    img = device.read_output()
    hit = np.isfinite(img[..., 3])
    assert 0.02 < hit.mean() < 0.9
    assert not img[hit][:, :3].any()                                          # exactly 0
    assert same_values(img[~hit][:, :3], np.broadcast_to(np.asarray(sky, np.float32), img[~hit][:, :3].shape))
    # the first-hit distance of the centre pixel's neighbourhood is ~2 (the camera is 2 above the plane)
    centre = img[h // 2 - 2: h // 2 + 2, w // 2 - 2: w // 2 + 2, 3]
    assert np.all(np.abs(centre[np.isfinite(centre)] - 2.0) < 0.05)
    c = device.counters()
    assert c["paths"] == w * h * 3 and c["nodes"] == c["paths"] and c["hits"] == int(hit.sum()) * 3


@pytest.mark.parametrize("traversal", [0, 1, 2])
def test_closest_hits_against_brute_force_in_float64(pbr, device, traversal):
    """The walk over the host-built BVH (20 k triangles) returns the geometric closest hit: Moeller-Trumbore over ALL
    triangles in float64 numpy, no tree, no oracle."""
    pbr.cfg_reset()
    sc = pbr.HostScene.generate("dragon", 9, 20000)
    arr = sc.arrays()
    device.upload_scene(sc.desc)
    cfg = sc.config(8, 8)
    cfg.traversal = traversal
    device.configure(cfg)
    rng = np.random.default_rng(3)
    v = arr["vertices"][:, :3]
    n = 400
    rays = np.zeros((n, 6), np.float32)
    rays[:, 0:3] = rng.uniform(v.min(0) - 0.2, v.max(0) + 0.2, (n, 3))
    d = rng.normal(size=(n, 3))
    rays[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    t, face, _, _ = device.diag_trace(rays)
    tri = v[arr["facesV"][:, :3].astype(np.int64)].astype(np.float64)
    a, e1, e2 = tri[:, 0], tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    hits = 0
    for k in range(n):
        o, dd = rays[k, 0:3].astype(np.float64), rays[k, 3:6].astype(np.float64)
        p = np.cross(dd, e2)
        det = (e1 * p).sum(1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            tv = o - a
            u = (tv * p).sum(1) * inv
            q = np.cross(tv, e1)
            vv = (q * dd).sum(1) * inv
            tt = (e2 * q).sum(1) * inv
        ok = (u >= -1e-6) & (vv >= -1e-6) & (u + vv <= 1 + 1e-6) & (tt > 1e-4) & np.isfinite(tt)
        best = tt[ok].min() if ok.any() else np.inf
        if np.isfinite(best) or np.isfinite(t[k]):
            hits += 1
            assert abs(float(t[k]) - best) <= 1e-3 * max(1.0, best), (k, float(t[k]), best)
    assert hits > 50
