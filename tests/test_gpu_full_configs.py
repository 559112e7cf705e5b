"""BASELINE.json's three large configurations at THEIR OWN SIZE on the GPU (the stage and whole-image tests of
test_gpu_parity.py use <= 20 000-triangle stand-ins the oracle can render in seconds):

  configs[2]  Dragon-class, ~870 k triangles, 1920 x 1080
  configs[3]  Sponza-class, ~260 k triangles, 1920 x 1080
  configs[4]  hairball, 2 M triangles, 3840 x 2160

i.e. million-node streams with the full LDS prefix and cold DFS tails, 8.3 M-pixel queues, renders that cross the
16 GiB frame-buffer cap.  Per configuration: (a) the WHOLE image and the whole debug image (per-pixel node /
face-test counters, pathtracing.cl:73-78) and the launch's counters against the oracle — 2.07 M / 8.3 M pixels, every one
of them — for the state machine the tuner keeps on these scenes and for the tuner's own run; the five other schedules
are held to the same frame bit for bit (so every schedule equals the oracle everywhere); (b) size-independent
properties: run-to-run identical, 1 + 2 == 3 frames, paths == W * H * frames, the debug image sums to the launch's
counters; (c) 8-way tile shards re-assemble the unsharded frame; (d) a render long enough for two launch pairs equals
the same frames rendered in one-frame launches; (e) 2^21 random rays through the walk over each full-size tree.
Walk: pt_bvh.cl:82-123; kernel: pathtracing.cl:207-334."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import same_values, describe_mismatch

pytestmark = pytest.mark.gpu

# name: (generator kind, seed, triangle budget, width, height, max_depth) — bench.py's WORKLOADS at BASELINE.json's sizes
FULL = {
    "dragon": ("dragon", 1, 870000, 1920, 1080, 3),
    "sponza": ("sponza", 2, 260000, 1920, 1080, 3),
    "hairball": ("hairball", 3, 2000000, 3840, 2160, 3),
}
PLANS = ["refill-lean", "refill-wide", "phased-lean", "phased-wide", "phased-mid", "refill-mid", "phased-dual"]

_scenes = {}


def full_scene(pbr, name):
    """The scene (host BVH build: seconds at 2 M triangles) once per session."""
    kind, seed, tris, w, h, depth = FULL[name]
    pbr.cfg_reset()
    pbr.cfg_set(**{"render.max_depth": depth})
    if name not in _scenes:
        _scenes[name] = pbr.HostScene.generate(kind, seed, tris)
    sc = _scenes[name]
    return sc, sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h), w, h


@pytest.fixture()
def device(pbr, gpu_device):
    dev = pbr.Device(gpu_device)
    yield dev
    dev.close()


_whole = {}


def oracle_whole_frames(pbr, oracle, name):
    """The oracle's complete image, debug image and counters for the first frames of a full-size configuration (3 frames
    at 1920 x 1080, 2 at 3840 x 2160: seconds on the GPU box's 256 host threads), once per session."""
    if name not in _whole:
        sc, cfg, cam, px, w, h = full_scene(pbr, name)
        frames = 2 if w * h > 4000000 else 3
        ref = oracle.Renderer(sc.desc, cfg, threads=os.cpu_count() or 8)
        image = ref.render(0, pbr.frame_seeds(0, frames), px, cam)
        _whole[name] = (frames, image.copy(), ref.debug.copy(), ref.counter_dict())
    return _whole[name]


@pytest.mark.parametrize("name", sorted(FULL))
def test_whole_frame_against_the_oracle_in_every_schedule(pbr, oracle, device, name):
    """The parity unit at full size is the whole frame: every pixel of the image and of the debug image and the launch's
    counters equal the oracle's for the state machine (phased-mid); each of the six other plans then equals that frame
    bit for bit — and so the oracle's, everywhere."""
    sc, cfg, cam, px, w, h = full_scene(pbr, name)
    assert sc.info["faces"] > 0.95 * FULL[name][2]
    frames, image, debug, counters = oracle_whole_frames(pbr, oracle, name)
    seeds = pbr.frame_seeds(0, frames)
    assert np.isfinite(image[..., 3]).mean() > 0.05, "the camera sees geometry"

    device.upload_scene(sc.desc)
    device.configure(cfg)
    first = None
    for plan_name in ["phased-mid"] + [p for p in PLANS if p != "phased-mid"]:
        device.pin_plan(PLANS.index(plan_name))
        device.reset_accum()
        before = device.counters()
        device.render(0, seeds, px, cam)
        assert device.last_plan()[0] == plan_name
        got, dbg = device.read_output(), device.read_debug()
        spent = {k: v - before[k] for k, v in device.counters().items()}
        if first is None:
            assert same_values(got, image), plan_name + ": " + describe_mismatch(got, image)
            assert same_values(dbg, debug), plan_name + " debug: " + describe_mismatch(dbg, debug)
            assert spent == counters, (plan_name, spent, counters)
            first = (got, dbg, spent)
        else:
            assert same_values(got, first[0]) and same_values(dbg, first[1]), plan_name
            assert spent == first[2], plan_name
    assert first[2]["paths"] == w * h * frames
    assert np.isfinite(first[0][..., :3]).all() and first[0][..., :3].min() >= 0.0


@pytest.mark.parametrize("name", sorted(FULL))
def test_properties_at_full_size(pbr, oracle, device, name):
    """No schedule forced: the tuner screens its plans on these very frames — whichever renders which frame, the bits
    are the same, and they are the oracle's for the whole frame."""
    sc, cfg, cam, px, w, h = full_scene(pbr, name)
    frames, image, debug, counters = oracle_whole_frames(pbr, oracle, name)
    device.upload_scene(sc.desc)
    device.configure(cfg)
    device.render(0, pbr.frame_seeds(0, frames), px, cam)
    tuned = device.read_output()
    assert same_values(tuned, image), describe_mismatch(tuned, image)
    assert same_values(device.read_debug(), debug) and device.counters() == counters
    device.reset_accum()
    device.render(0, pbr.frame_seeds(0, 3), px, cam)
    a = device.read_output()
    device.reset_accum()
    device.render(0, pbr.frame_seeds(0, 3), px, cam)
    assert same_values(device.read_output(), a)                        # run-to-run identical
    device.reset_accum()
    device.render(0, pbr.frame_seeds(0, 1), px, cam)
    before = device.counters()
    device.render(1, pbr.frame_seeds(1, 2), px, cam)
    assert same_values(device.read_output(), a)                        # 1 + 2 frames == 3 frames
    after = device.counters()
    # the debug image holds the LAST frame's per-pixel counters (x = face tests / 1082, y = node visits / 1265,
    # pathtracing.cl:73-78): rendered alone, its sums are that launch's counters
    device.render(3, pbr.frame_seeds(3, 1), px, cam)
    last = device.counters()
    dbg = device.read_debug().astype(np.float64)
    nodes = np.rint(dbg[..., 1] * 1265.0).sum()
    tris = np.rint(dbg[..., 0] * 1082.0).sum()
    assert int(nodes) == last["nodes"] - after["nodes"] and int(tris) == last["tris"] - after["tris"]
    assert after["paths"] - before["paths"] == w * h * 2


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("name", sorted(FULL))
def test_ray_ordered_walk_at_full_size(pbr, oracle, device, name, mode):
    """pbr_config.traversal = six / eight orders (3: eight orders over compact 64-byte records, the same visits) at the configurations' own size (not a reference mode: the reference walks
    one fixed order, pt_bvh.cl:102,112).  Two claims, whole frame each:
      (1) HIP(ordered) == oracle(ordered) bit for bit — image, debug image (this walk's own counters), launch counters —
          in the state machine, in the two-paths-per-lane plan and in whatever the tuner runs, and 8-way tile shards of it
          re-assemble that frame;
      (2) oracle(ordered) against oracle(reference order): SURVEY.md section 8(c)'s image tolerance, |d| <= 1e-4 per
          channel on >= 99.5 % of the pixels and mean |d| <= 1e-5; the share of bit-identical pixels is printed (only
          equal-distance hits — exact ties, coplanar faces — can differ), with fewer node visits and face tests."""
    import torch
    sc, cfg, cam, px, w, h = full_scene(pbr, name)
    frames, ref_image, ref_debug, ref_counters = oracle_whole_frames(pbr, oracle, name)
    seeds = pbr.frame_seeds(0, frames)
    cfg.traversal = mode
    walk = oracle.Renderer(sc.desc, cfg, threads=os.cpu_count() or 8)
    image = walk.render(0, seeds, px, cam)
    counters = walk.counter_dict()

    with np.errstate(invalid="ignore"):
        d = np.abs(image.astype(np.float64) - ref_image)[..., :3]
    d[np.isnan(image[..., :3]) & np.isnan(ref_image[..., :3])] = 0.0
    identical = np.all((image == ref_image) | (np.isnan(image) & np.isnan(ref_image)), axis=2).mean()
    print("%s traversal %d: %.5f %% of %d pixels bit-identical to the reference order's, max |d| %.3g; node visits %.3fx, face tests %.3fx" % (
        name, mode, 100 * identical, w * h, d.max(), counters["nodes"] / ref_counters["nodes"], counters["tris"] / ref_counters["tris"]))
    assert (d.max(axis=2) <= 1e-4).mean() >= 0.995 and d.mean() <= 1e-5
    assert identical >= 0.999
    assert counters["paths"] == ref_counters["paths"]
    assert counters["nodes"] < ref_counters["nodes"] and counters["tris"] < ref_counters["tris"]

    device.upload_scene(sc.desc)
    device.configure(cfg)
    for plan_name in ("phased-mid", "phased-dual", None):
        device.pin_plan(PLANS.index(plan_name) if plan_name else -1)
        device.reset_accum()
        before = device.counters()
        device.render(0, seeds, px, cam)
        got, dbg = device.read_output(), device.read_debug()
        spent = {k: v - before[k] for k, v in device.counters().items()}
        assert same_values(got, image), "%s: %s" % (plan_name, describe_mismatch(got, image))
        assert same_values(dbg, walk.debug), "%s debug: %s" % (plan_name, describe_mismatch(dbg, walk.debug))
        assert spent == counters, (plan_name, spent, counters)

    world, gathered = 8, None
    for rank in range(world):
        c = pbr.Config.from_buffer_copy(cfg)
        c.tile_world, c.tile_rank = world, rank
        device.configure(c)
        device.render(0, seeds, px, cam)
        if gathered is None:
            gathered = torch.zeros(world * device.tile_bytes() // 4, dtype=torch.float32, device="cuda")
        device.export_tiles(gathered.data_ptr() + rank * device.tile_bytes())
    torch.cuda.synchronize()
    device.import_tiles(gathered.data_ptr())
    assert same_values(device.read_full(), image)


@pytest.mark.parametrize("name", sorted(FULL))
def test_eight_way_shards_equal_the_unsharded_frame(pbr, device, name):
    import torch
    sc, cfg, cam, px, w, h = full_scene(pbr, name)
    world, seeds = 8, pbr.frame_seeds(0, 2)
    device.upload_scene(sc.desc)
    device.configure(cfg)
    device.render(0, seeds, px, cam)
    want, total = device.read_output(), device.counters()
    d = pbr.Device(0)
    d.upload_scene(sc.desc)
    gathered = None
    summed = {k: 0 for k in total}
    for rank in range(world):
        c = pbr.Config.from_buffer_copy(cfg)
        c.tile_world, c.tile_rank = world, rank
        d.configure(c)
        d.render(0, seeds, px, cam)
        if gathered is None:
            gathered = torch.zeros(world * d.tile_bytes() // 4, dtype=torch.float32, device="cuda")
        d.export_tiles(gathered.data_ptr() + rank * d.tile_bytes())
        for k, v in d.counters().items():
            summed[k] += v
    torch.cuda.synchronize()
    d.import_tiles(gathered.data_ptr())
    got = d.read_full()
    assert same_values(got, want), describe_mismatch(got, want)
    assert summed == total                                             # every path walked exactly once, same visits
    d.close()


@pytest.mark.parametrize("name", ["sponza", "hairball"])
def test_render_across_the_frame_buffer_cap(pbr, device, name):
    """A multi-frame render keeps {finalColor, focus} of every (pixel, frame) in a buffer capped at 16 GiB; beyond it
    the render runs as several launch pairs (1080p: 517 frames, 3840 x 2160: 129).  Two pairs == the frames one by one,
    as the reference's viewer renders them (PathTracer.cpp:59-71)."""
    sc, cfg, cam, px, w, h = full_scene(pbr, name)
    cap = (16 << 30) // (16 * w * h)
    frames = cap + 3
    device.upload_scene(sc.desc)
    device.configure(cfg)
    device.pin_plan(4)                                                 # one plan, no tuning chunks: the cap alone splits the render
    device.render(0, pbr.frame_seeds(0, frames), px, cam)
    assert device.last_trace()[1] == 2                                 # two path-tracing launches
    fused = device.read_output()
    assert device.counters()["paths"] == w * h * frames
    device.pin_plan(-1)                                                # frame by frame, whichever plans the tuner tries
    device.reset_accum()
    for k in range(frames):
        device.render(k, pbr.frame_seeds(k, 1), px, cam)
    single = device.read_output()
    assert same_values(fused, single), describe_mismatch(fused, single)


@pytest.mark.parametrize("brdf,shadow", [(0, 1), (1, 1), (0, 0)])
def test_lights_shadow_rays_and_schlick_at_full_size(pbr, oracle, device, brdf, shadow):
    """The kernel variants the three BASELINE configurations do not reach (BRDF 0, lights, shadow rays — K9, K10, K16) on
    the Sponza-class scene at its own size: 260 k triangles, 1920 x 1080, an orb light and a point light, the whole
    image + debug image against the oracle in the state machine and in the lock-step kernel."""
    pbr.cfg_reset()
    pbr.cfg_set(**{"render.max_depth": 3, "render.brdf": brdf})
    sc = pbr.HostScene.generate("sponza", 2, 260000)
    w, h, frames = 1920, 1080, 2
    cfg, cam, px = sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h)
    cfg.shadow_rays = shadow
    v = sc.arrays()["vertices"][:, :3]
    centre, size = (v.min(0) + v.max(0)) / 2, (v.max(0) - v.min(0))
    lights = np.zeros((2, 12), np.float32)
    lights[0] = [centre[0], centre[1] + 0.2 * size[1], centre[2], 0, 4.0, 3.5, 3.0, 0, 2, 0.05 * float(size.max()), 0, 0]   # orb
    lights[1] = [centre[0] - 0.2 * size[0], centre[1], centre[2] + 0.1 * size[2], 0, 1, 1, 1, 0, 1, 0, 0, 0]               # point
    desc = pbr.SceneDesc.from_buffer_copy(sc.desc)
    desc.lights, desc.num_lights = lights.ctypes.data, 2
    seeds = pbr.frame_seeds(0, frames)
    ref = oracle.Renderer(desc, cfg, threads=os.cpu_count() or 8)
    want = ref.render(0, seeds, px, cam)
    device.upload_scene(desc)
    device.configure(cfg)
    for plan in (4, 5):
        device.pin_plan(plan)
        device.reset_accum()
        before = device.counters()
        device.render(0, seeds, px, cam)
        got, dbg = device.read_output(), device.read_debug()
        assert same_values(got, want), describe_mismatch(got, want)
        assert same_values(dbg, ref.debug)
        assert {k: v - before[k] for k, v in device.counters().items()} == ref.counter_dict()
    assert ref.counter_dict()["paths"] == w * h * frames


def test_depth_of_field_at_full_size(pbr, oracle, device):
    """K2's depth of field (pathtracing.cl:58-65, pt_utils.cl:344-373) at 1920 x 1080 on the Dragon-class scene: frame by
    frame (every pixel reads the focus pixel's previous-frame distance), a 16-row band against the oracle."""
    sc, cfg, cam, px, w, h = full_scene(pbr, "dragon")
    cam.focusPoint[0], cam.focusPoint[1] = w // 2, h // 2
    rows = (h // 2 - 8, h // 2 + 8)
    band = slice(rows[0], rows[1])
    ref = oracle.Renderer(sc.desc, cfg, threads=os.cpu_count() or 8)
    device.upload_scene(sc.desc)
    device.configure(cfg)
    for k, seed in enumerate(pbr.frame_seeds(0, 3)):
        weight = float(np.float32(k) / np.float32(k + 1))
        # the oracle needs the whole previous frame only through two of its pixels' .w: this pixel's (in the band) and the
        # focus pixel's (in the band too) — rendering the band is enough as long as both lie inside it
        out = ref.render_frame(float(seed), weight, px, cam, rows=rows)
        ref.image[band] = out[band]
        device.render_frame(float(seed), weight, px, cam)
        got = device.read_output()
        assert same_values(got[band], ref.image[band]), "frame %d: %s" % (k, describe_mismatch(got[band], ref.image[band]))
        device.accumulate()
    assert np.isfinite(ref.image[band][..., 3]).any()


@pytest.mark.parametrize("kind,seed,triangles", [("sponza", 2, 260000), ("dragon", 1, 870000), ("hairball", 3, 2000000)])
def test_traversal_soak_on_the_large_scenes(pbr, oracle, gpu_device, kind, seed, triangles):
    """2^21 random rays (origins in and around the scene, directions uniform, some axis-parallel and some with one
    exactly-zero component: 1 / 0 in the slab test) through the walk over the full-size trees: distance, face, normal and
    the node / face-test counts of every ray against the oracle's walk."""
    from conftest import same_values, describe_mismatch
    pbr.cfg_reset()
    sc = pbr.HostScene.generate(kind, seed, triangles)
    cfg = sc.config(64, 64)
    v = sc.arrays()["vertices"][:, :3]
    rng = np.random.default_rng(5)
    n = 1 << 21
    rays = np.concatenate([rng.uniform(v.min(0) - 0.3, v.max(0) + 0.3, (n, 3)), rng.normal(size=(n, 3))], axis=1).astype(np.float32)
    k = np.arange(n // 16)
    rays[k, 3 + k % 3] = 0.0                                                  # one exactly-zero component
    rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    rays[7:16, 3:] = np.nan                                                   # and a few rays that are not rays at all
    rays[:7, 3:] = [[1, 0, 0], [0, 1, 0], [0, 0, -1], [0, -1, 0], [1, 1, 0], [0, 0, 1], [-1, 0, 0]]
    dev = pbr.Device(gpu_device)
    dev.upload_scene(sc.desc)
    t, face, normal, counts = dev.diag_trace(rays)
    pieces = [(a, min(a + (1 << 15), n)) for a in range(0, n, 1 << 15)]
    with ThreadPoolExecutor(min(64, os.cpu_count() or 8)) as pool:
        parts = list(pool.map(lambda s: oracle.trace_rays(sc.desc, cfg, rays[s[0]:s[1]]), pieces))
    ot, oface, onormal, ocounts = (np.concatenate([p[k] for p in parts]) for k in range(4))
    assert same_values(t, ot), describe_mismatch(t, ot)
    hit = np.isfinite(ot)
    assert hit.sum() > n // 20
    assert np.array_equal(face[hit], oface[hit]) and same_values(normal[hit], onormal[hit])
    assert np.array_equal(counts, ocounts)
    assert dev.guard_trips() == [0, 0, 0]
    dev.close()
