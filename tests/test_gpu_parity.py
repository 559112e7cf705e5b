"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs — bit for bit (compared numerically: -0 == +0, NaN == NaN) — stage by stage and
on whole images, plus size-independent properties at BASELINE.json's full resolution.
Every test here needs a real MI355X."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, same_values, describe_mismatch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden  # noqa: E402


@pytest.fixture()
def device(pbr, gpu_device):
    dev = pbr.Device(gpu_device)
    yield dev
    dev.close()


# the seven plans launch() chooses from, as pbr_diag_pin_plan numbers them
PLANS = {"refill-lean": 0, "refill-wide": 1, "phased-lean": 2, "phased-wide": 3, "phased-mid": 4, "refill-mid": 5, "phased-dual": 6}


def force_schedule(device, plan):
    """Pin one of the seven plans (by name) for this test's device; None leaves the choice to the tuner.  The library reads
    no environment variable: tests steer it through pbr_diag_pin_plan / pbr_diag_set_knob."""
    if plan is not None:
        device.pin_plan(PLANS[plan])


def make_scene(pbr, kind="cornell", seed=1, triangles=0, **cfg):
    pbr.cfg_reset()
    pbr.cfg_set(**cfg)
    return pbr.HostScene.generate(kind, seed, triangles)


def both_render(pbr, oracle, dev, sc, w, h, frames, first=0, cam=None, cfg=None, desc=None):
    cfg = cfg or sc.config(w, h)
    cam = cam or sc.camera()
    desc = desc or sc.desc
    px = pbr.pixel_dimension(w, h)
    seeds = pbr.frame_seeds(first, frames)
    ref = oracle.Renderer(desc, cfg, threads=8)
    want = ref.render(first, seeds, px, cam)
    dev.upload_scene(desc)
    dev.configure(cfg)
    dev.render(first, seeds, px, cam)
    return dev.read_output(), want, ref


# ----------------------------------------------------------------------------------------------
# stages
# ----------------------------------------------------------------------------------------------

def test_math_layer_bit_exact(pbr, oracle, device):
    rng = np.random.default_rng(0)
    wide = np.concatenate([rng.uniform(-60, 60, 200000), rng.uniform(-2e4, 2e4, 50000), [0, -0.0, np.inf, -np.inf, np.nan, 3e8, 1e-40]]).astype(np.float32)
    unit = np.concatenate([rng.uniform(-1, 1, 200000), [-1, 1, 0.5, -0.5, 1.5, np.nan]]).astype(np.float32)
    for op, x in (("sin", wide), ("cos", wide), ("tan", wide), ("randhash", wide), ("acos", unit), ("atan", wide)):
        got, want = device.diag_math(op, x), oracle.math(op, x)
        assert same_values(got, want), op + ": " + describe_mismatch(got, want)
    base = np.concatenate([rng.uniform(0, 1, 200000), rng.uniform(0, 30, 50000)]).astype(np.float32)
    expo = np.concatenate([10 ** rng.uniform(-3, 5.5, 200000), rng.uniform(-20, 20, 50000)]).astype(np.float32)
    assert same_values(device.diag_math("pow", base, expo), oracle.math("pow", base, expo))
    sp = np.array([0, -0.0, 1, -1, np.inf, -np.inf, np.nan, 2, -2, 0.5, -0.5, 3, -3, 1e-40, 16777216, 16777217, -7], np.float32)
    X, Y = [a.ravel().copy() for a in np.meshgrid(sp, sp)]
    got, want = device.diag_math("pow", X, Y), oracle.math("pow", X, Y)
    assert same_values(got, want), describe_mismatch(got, want)


@pytest.mark.parametrize("kind,triangles,skip", [("cornell", 0, True), ("cornell", 0, False), ("sponza", 8000, True), ("dragon", 8000, True), ("hairball", 6000, True)])
def test_traversal_bit_exact(pbr, oracle, device, kind, triangles, skip):
    sc = make_scene(pbr, kind, 2, triangles, **{"bvh.skip_ahead": skip})
    cfg = sc.config(64, 64)
    v = sc.arrays()["vertices"][:, :3]
    rng = np.random.default_rng(3)
    rays = np.concatenate([rng.uniform(v.min(0) - 0.3, v.max(0) + 0.3, (5000, 3)), rng.normal(size=(5000, 3))], axis=1).astype(np.float32)
    rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    rays[:7, 3:] = [[1, 0, 0], [0, 1, 0], [0, 0, -1], [0, -1, 0], [1, 1, 0], [0, 0, 1], [-1, 0, 0]]   # axis-parallel: 1/0 = inf in the slab test
    device.upload_scene(sc.desc)
    t, face, normal, counts = device.diag_trace(rays)
    ot, oface, onormal, ocounts = oracle.trace_rays(sc.desc, cfg, rays)
    assert same_values(t, ot), describe_mismatch(t, ot)
    hit = np.isfinite(ot)
    assert hit.sum() > 500
    assert np.array_equal(face[hit], oface[hit]) and same_values(normal[hit], onormal[hit])
    assert np.array_equal(counts, ocounts)
    assert device.guard_trips() == [0, 0, 0]


MATERIALS_SA = [
    [1, 1, 0, 0, 0, 1, 0, 0, .7, .7, .7, 0, 1, 1, 1, 0],
    [1, 1, 200, 200, .7, .6, 0, 0, .8, .8, .85, 0, .9, .9, .9, 0],
    [1, 1, 100000, 100000, 1, 1, 0, 0, 1, 1, 1, 0, 1, 1, 1, 0],       # suzanne.mtl "Suzanne": pow( x, 1e5 )
    [1, 1, 10, 1000, .3, .8, 0, 0, .5, .6, .7, 0, .9, .8, .7, 0],     # anisotropic
    [0.1, 1.5, 0, 0, .2, 1, 0, 0, .95, .95, 1, 0, 1, 1, 1, 0],        # glass
    [0.6, 1.33, 50, 50, .5, .5, 0, 0, .2, .4, .9, 0, 1, 1, 1, 0],
]
MATERIALS_SCHLICK = [
    [1, 1, 1, 1, .7, .7, .7, 0, 1, 1, 1, 0],
    [1, 1, 1, 0.15, .8, .8, .85, 0, .9, .9, .9, 0],
    [1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 0],                             # mirror
    [1, 1, 0.3, 0.6, .5, .6, .7, 0, .9, .8, .7, 0],                   # anisotropic, p < 1
    [0.1, 1.5, 1, 0.05, .95, .95, 1, 0, 1, 1, 1, 0],
]


def single_material_desc(pbr, sc, brdf, mtl):
    """The scene's geometry with ONE material (index 0 is what the diag kernels read)."""
    d = pbr.SceneDesc.from_buffer_copy(sc.desc)
    m = np.tile(np.asarray(mtl, np.float32), (sc.desc.num_materials, 1)).copy()
    d.materials, d.brdf = m.ctypes.data, brdf
    return d, m


@pytest.mark.parametrize("brdf,materials", [(1, MATERIALS_SA), (0, MATERIALS_SCHLICK)])
def test_brdf_and_new_ray_bit_exact(pbr, oracle, device, brdf, materials):
    sc = make_scene(pbr, **{"render.brdf": brdf})
    rng = np.random.default_rng(9)
    n = 4000

    def unit(a):
        return a / np.linalg.norm(a, axis=1, keepdims=True)

    normal = unit(rng.normal(size=(n, 3)))
    out_dir = unit(rng.normal(size=(n, 3)))
    out_dir -= 2 * np.maximum(0, (out_dir * normal).sum(1, keepdims=True)) * normal     # arriving: out_dir . n <= 0
    in_dir = unit(rng.normal(size=(n, 3)))
    in_dir += 2 * np.maximum(0, -(in_dir * normal).sum(1, keepdims=True)) * normal      # leaving: in_dir . n >= 0
    ev = np.zeros((n, 16), np.float32)
    ev[:, 0:3], ev[:, 3:6], ev[:, 6:9] = out_dir, in_dir, normal
    nr = np.zeros((n, 12), np.float32)
    nr[:, 0:3] = rng.uniform(-1, 1, (n, 3))
    nr[:, 3:6], nr[:, 6:9] = out_dir, normal
    nr[: n // 4, 6:9] *= -1                                                               # back-facing normals too
    nr[:, 9] = rng.uniform(0.01, 5, n)
    nr[:, 10] = rng.uniform(0, 300, n)
    fp = ctypes.POINTER(ctypes.c_float)

    for mtl in materials:
        desc, keep = single_material_desc(pbr, sc, brdf, mtl)
        device.upload_scene(desc)
        m0 = np.asarray(mtl, np.float32)
        want = np.empty((n, 4), np.float32)
        oracle.lib().orc_brdf_eval(brdf, m0.ctypes.data, ev.ctypes.data_as(fp), n, want.ctypes.data_as(fp))
        got = device.diag_brdf(ev)
        assert same_values(got, want), "brdf %r: %s" % (mtl[:6], describe_mismatch(got, want))
        want = np.empty((n, 8), np.float32)
        oracle.lib().orc_new_ray(brdf, m0.ctypes.data, nr.ctypes.data_as(fp), n, want.ctypes.data_as(fp))
        got = device.diag_new_ray(nr)
        assert same_values(got, want), "new ray %r: %s" % (mtl[:6], describe_mismatch(got, want))


# ----------------------------------------------------------------------------------------------
# whole images
# ----------------------------------------------------------------------------------------------

@pytest.mark.parametrize("schedule", ["refill-lean", "refill-mid", "phased-lean", "phased-mid", "phased-dual", None])
@pytest.mark.parametrize("cfg", [
    {"render.max_depth": 4},
    {"render.max_depth": 4, "render.brdf": 0},
    {"render.max_depth": 8, "render.max_added_depth": 2},
    {"render.samples": 3},
    {"render.antialiasing": 0.0, "render.max_depth": 1, "render.max_added_depth": 0},
])
def test_cornell_image_bit_exact(pbr, oracle, device, schedule, cfg):
    force_schedule(device, schedule)
    sc = make_scene(pbr, **cfg)
    got, want, ref = both_render(pbr, oracle, device, sc, 64, 48, 5)
    assert same_values(got, want), describe_mismatch(got, want)
    assert same_values(device.read_debug(), ref.debug)
    assert device.counters() == ref.counter_dict()
    assert device.guard_trips() == [0, 0, 0]


@pytest.mark.parametrize("kind,triangles,w,h", [("sponza", 20000, 96, 56), ("dragon", 20000, 64, 64), ("hairball", 20000, 64, 64)])
@pytest.mark.parametrize("schedule", ["refill-wide", "refill-lean", "refill-mid", "phased-wide", "phased-lean", "phased-mid", "phased-dual"])
def test_larger_scenes_bit_exact(pbr, oracle, device, kind, triangles, w, h, schedule):
    """Both schedules (pt_kernel.hpp: lock-step / lane state machine) and the three register budgets, with the tree top
    staged in LDS, against the oracle."""
    force_schedule(device, schedule)
    sc = make_scene(pbr, kind, 4, triangles)
    got, want, ref = both_render(pbr, oracle, device, sc, w, h, 4)
    assert same_values(got, want), describe_mismatch(got, want)
    assert same_values(device.read_debug(), ref.debug)
    assert device.counters() == ref.counter_dict()


@pytest.mark.parametrize("plan,name", [(0, "refill-lean"), (1, "refill-wide"), (2, "phased-lean"), (3, "phased-wide"), (4, "phased-mid"), (5, "refill-mid"), (6, "phased-dual")])
@pytest.mark.parametrize("kind,triangles,cfg", [
    ("cornell", 0, {"render.max_depth": 5, "render.max_added_depth": 2}),
    ("dragon", 12000, {"render.max_depth": 3, "render.brdf": 0}),
    ("hairball", 9000, {"render.max_depth": 3, "render.samples": 2}),
])
def test_every_tuner_candidate_bit_exact(pbr, oracle, device, plan, name, kind, triangles, cfg):
    """The seven plans launch() chooses from (pbr_diag_pin_plan pins one): 4, 6 and 8 waves per SIMD of the lock-step kernel and of
    the lane state machine, 768- and 1024-thread blocks, and the state machine with two paths per lane — each against the oracle, images, debug image and counters."""
    device.pin_plan(plan)
    sc = make_scene(pbr, kind, 7, triangles, **cfg)
    got, want, ref = both_render(pbr, oracle, device, sc, 88, 56, 6)
    assert device.last_plan()[0] == name
    assert same_values(got, want), describe_mismatch(got, want)
    assert same_values(device.read_debug(), ref.debug)
    assert device.counters() == ref.counter_dict()


def test_schedule_tuner_through_a_viewer_then_a_batch(pbr, oracle, device):
    """No schedule forced: launch() screens its seven plans, times the finalists and keeps one (pbr_hip.hip) — on
    frame-by-frame calls first, as the reference's viewer renders (PathTracer.cpp:60-68), which cannot separate a
    launch's fixed cost from its per-frame cost; the first long render then times the finalists again.  Whatever
    it picks, the accumulated image is the oracle's."""
    sc = make_scene(pbr, **{"render.max_depth": 4})
    w, h = 40, 32
    cfg, cam, px = sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h)
    ref = oracle.Renderer(sc.desc, cfg, threads=8)
    device.upload_scene(sc.desc)
    device.configure(cfg)
    first = 0
    # screening 7 plans x 2 launches, refinement <= 3 finalists x 4 launches — and once more when the best two end within
    # 5 % of each other (a close call, decided by this box's timing): 26 or 38 launches.  (Round 5: the assertion stood at 30
    # and failed once in a dozen full-suite runs.)
    for n in [1] * 44 + [60, 40]:
        seeds = pbr.frame_seeds(first, n)
        want = ref.render(first, seeds, px, cam)
        device.render(first, seeds, px, cam)
        first += n
        if n == 1 and first == 44:
            assert device.last_plan()[1] >= 0, "44 single-frame launches settle the tuner"
    assert device.last_plan()[1] >= 0
    got = device.read_output()
    assert same_values(got, want), describe_mismatch(got, want)
    assert device.counters() == ref.counter_dict()


@pytest.mark.parametrize("chunk", [1, 2, 3])
@pytest.mark.parametrize("schedule", ["refill-lean", "phased-lean"])
def test_frame_parallel_chunks_fold_in_frame_order(pbr, oracle, device, schedule, chunk):
    """Multi-frame renders hand out (pixel, frame) units and fold the frames afterwards (foldFrames);
    a render split into several launch pairs (knob chunk_frames) must give the same bits, debug
    image (last frame's counters) and totals as the oracle's frame-by-frame sequence."""
    force_schedule(device, schedule)
    device.set_knob("chunk_frames", chunk)
    sc = make_scene(pbr, **{"render.max_depth": 3, "render.samples": 2})
    got, want, ref = both_render(pbr, oracle, device, sc, 56, 40, 7)
    assert same_values(got, want), describe_mismatch(got, want)
    assert same_values(device.read_debug(), ref.debug)
    assert device.counters() == ref.counter_dict()


@pytest.mark.parametrize("w,h", [(8, 8), (24, 136), (200, 8), (72, 72)])
@pytest.mark.parametrize("schedule", ["refill-lean", "phased-mid", "refill-wide"])
def test_banded_queue_covers_every_pixel_once(pbr, oracle, device, schedule, w, h):
    """The pixel-slot queue is cut into 8 bands (one head per XCD, tiles column by column inside a
    band); image shapes with fewer tile rows than bands, one row, one column."""
    force_schedule(device, schedule)
    sc = make_scene(pbr, **{"render.max_depth": 2})
    got, want, ref = both_render(pbr, oracle, device, sc, w, h, 4)
    assert same_values(got, want), describe_mismatch(got, want)
    assert device.counters() == ref.counter_dict()


@pytest.mark.parametrize("batch", [1, 3, 16, 64])
@pytest.mark.parametrize("schedule", ["refill-lean", "refill-mid", "refill-wide"])
def test_batched_refill_does_not_change_results(pbr, oracle, device, schedule, batch):
    """Lock-step kernels (round 3): a lane whose unit is finished waits until `refill_batch` lanes of its wave wait with
    it before they take their next units together (1 = at once; 64 = only when the whole wave has finished).  When a lane
    takes its next unit changes nothing about the unit."""
    force_schedule(device, schedule)
    device.set_knob("refill_batch", batch)
    sc = make_scene(pbr, "sponza", 5, 9000, **{"render.max_depth": 4, "render.samples": 2})
    got, want, ref = both_render(pbr, oracle, device, sc, 104, 72, 5)
    assert same_values(got, want), describe_mismatch(got, want)
    assert same_values(device.read_debug(), ref.debug)
    assert device.counters() == ref.counter_dict()
    assert device.guard_trips() == [0, 0, 0]


@pytest.mark.parametrize("slots", [0, 7, 300])
def test_lds_staging_size_does_not_change_results(pbr, oracle, device, slots):
    """Any prefix of the hot-node ranking may be staged (knob lds_slots caps it; 0 = none)."""
    device.set_knob("lds_slots", slots)
    sc = make_scene(pbr, "sponza", 4, 12000)
    got, want, ref = both_render(pbr, oracle, device, sc, 64, 40, 3)
    assert same_values(got, want), describe_mismatch(got, want)
    assert device.counters() == ref.counter_dict()


def lit_scene(pbr, brdf):
    """Cornell + one orb light above the hole + one point light (a scene file with lights needs
    render.shadow_rays = 1 at load time, ObjParser.cpp:133; here the arrays are patched)."""
    sc = make_scene(pbr, **{"render.brdf": brdf, "render.max_depth": 4})
    lights = np.zeros((2, 12), np.float32)
    lights[0] = [0.1, 1.6, 0.2, 0, 4.0, 3.5, 3.0, 0, 2, 0.12, 0, 0]
    lights[1] = [-0.5, 0.4, 0.6, 0, 1, 1, 1, 0, 1, 0, 0, 0]
    desc = pbr.SceneDesc.from_buffer_copy(sc.desc)
    desc.lights, desc.num_lights = lights.ctypes.data, 2
    return sc, desc, lights


@pytest.mark.parametrize("schedule", [None, "phased-dual"])
@pytest.mark.parametrize("brdf", [1, 0])
@pytest.mark.parametrize("shadow", [0, 1])
def test_lights_and_shadow_rays_bit_exact(pbr, oracle, device, brdf, shadow, schedule):
    force_schedule(device, schedule)
    sc, desc, keep = lit_scene(pbr, brdf)
    cfg = sc.config(64, 64)
    cfg.shadow_rays = shadow
    got, want, ref = both_render(pbr, oracle, device, sc, 64, 64, 4, cfg=cfg, desc=desc)
    assert same_values(got, want), describe_mismatch(got, want)
    assert device.counters() == ref.counter_dict()
    # the orb is visible through the hole: some pixels carry its colour, not the sky's
    assert (want[..., 0] > 1.5).any()


def test_depth_of_field_frame_by_frame(pbr, oracle, device):
    """setFocus: every pixel reads the focus pixel's previous-frame distance (pathtracing.cl:58-65),
    so frames cannot be fused; frame 0 has no previous distances (w = 0: no lens offset)."""
    sc = make_scene(pbr, **{"render.max_depth": 4})
    w, h = 64, 64
    cfg, cam, px = sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h)
    cam.focusPoint[0], cam.focusPoint[1] = 20, 30
    device.upload_scene(sc.desc)
    device.configure(cfg)
    with pytest.raises(pbr.PbrError, match="focusPoint"):
        device.render(0, pbr.frame_seeds(0, 2), px, cam)
    ref = oracle.Renderer(sc.desc, cfg, threads=8)
    for k, seed in enumerate(pbr.frame_seeds(0, 4)):
        weight = float(np.float32(k) / np.float32(k + 1))
        ref.image = ref.render_frame(float(seed), weight, px, cam)
        device.render_frame(float(seed), weight, px, cam)
        got = device.read_output()
        assert same_values(got, ref.image), "frame %d: %s" % (k, describe_mismatch(got, ref.image))
        device.accumulate()
    nofocus = pbr.Camera.from_buffer_copy(cam)
    nofocus.focusPoint[0] = nofocus.focusPoint[1] = -1
    plain = oracle.Renderer(sc.desc, cfg, threads=8).render(0, pbr.frame_seeds(0, 4), px, nofocus)
    assert not same_values(plain, ref.image)                       # the lens did something


def test_depth_of_field_across_tile_shards(pbr, oracle, gpu_device):
    """DOF with tile sharding: the focus pixel's tile lives on one rank; its previous-frame distance is handed
    round per frame (pbr_get_focus_depth on the owner -> broadcast -> pbr_set_focus_depth everywhere)."""
    sc = make_scene(pbr, **{"render.max_depth": 3})
    w, h, world = 64, 48, 3
    cfg, cam, px = sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h)
    cam.focusPoint[0], cam.focusPoint[1] = 37, 22
    ref = oracle.Renderer(sc.desc, cfg, threads=8)
    ranks = []
    for r in range(world):
        dev = pbr.Device(gpu_device)
        c = pbr.Config.from_buffer_copy(cfg)
        c.tile_world, c.tile_rank = world, r
        dev.upload_scene(sc.desc)
        dev.configure(c)
        ranks.append(dev)
    with pytest.raises(pbr.PbrError, match="pbr_set_focus_depth"):
        ranks[0].render_frame(0.0333, 0.0, px, cam)
    for k, seed in enumerate(pbr.frame_seeds(0, 4)):
        weight = float(np.float32(k) / np.float32(k + 1))
        ref.image = ref.render_frame(float(seed), weight, px, cam)
        owners = [dev.get_focus_depth(37, 22) for dev in ranks]
        assert sum(owned for _, owned in owners) == 1
        depth = [t for t, owned in owners if owned][0]              # the "broadcast"
        got = np.zeros((h, w, 4), np.float32)
        for dev in ranks:
            dev.set_focus_depth(depth)
            dev.render_frame(float(seed), weight, px, cam)
            got += dev.read_output()                                  # other ranks' tiles read 0
            dev.accumulate()
        assert same_values(got, ref.image), "frame %d: %s" % (k, describe_mismatch(got, ref.image))
    for dev in ranks:
        dev.close()


def test_write_input_and_explicit_weights(pbr, oracle, device):
    """CL::updateImageReadOnly + arbitrary pixelWeight: the reference's host ping-pong."""
    sc = make_scene(pbr)
    w, h = 40, 24
    cfg, cam, px = sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h)
    prev = np.random.default_rng(1).random((h, w, 4), dtype=np.float32)
    device.upload_scene(sc.desc)
    device.configure(cfg)
    device.write_input(prev)
    device.render_frame(1.25, 0.3, px, cam)
    ref = oracle.Renderer(sc.desc, cfg)
    ref.image = prev.copy()
    want = ref.render_frame(1.25, 0.3, px, cam)
    assert same_values(device.read_output(), want)


@pytest.mark.parametrize("world", [2, 3])
def test_tile_sharding_is_bit_identical(pbr, oracle, device, world):
    """SURVEY.md §8e invariant on one GPU: `world` contexts each render their tiles; the
    exported compact buffers, concatenated as an all-gather would, re-assemble the 1-rank frame."""
    import torch
    sc = make_scene(pbr, **{"render.max_depth": 4})
    w, h = 72, 40                                                   # 45 tiles: uneven shares
    cam, px, seeds = sc.camera(), pbr.pixel_dimension(w, h), pbr.frame_seeds(0, 5)
    full, want, _ = both_render(pbr, oracle, device, sc, w, h, 5)
    assert same_values(full, want)

    gathered, devs = None, []
    for rank in range(world):
        d = pbr.Device(0)
        devs.append(d)
        cfg = sc.config(w, h)
        cfg.tile_world, cfg.tile_rank = world, rank
        d.upload_scene(sc.desc)
        d.configure(cfg)
        d.render(0, seeds, px, cam)
        if gathered is None:
            gathered = torch.zeros(world * d.tile_bytes() // 4, dtype=torch.float32, device="cuda")
        d.export_tiles(gathered.data_ptr() + rank * d.tile_bytes())
        part = d.read_output()
        mask = pbr.tiles.rows_of_rank(w, h, world, rank)
        assert same_values(part[mask], want[mask]) and not part[~mask].any()
        # the exported buffer is the layout tiles.py describes
        host = gathered.cpu().numpy().reshape(world, -1)[rank]
        assert same_values(host.reshape(-1, 64, 4), pbr.tiles.pack_rank_tiles(np.where(mask[..., None], want, 0), world, rank))
    torch.cuda.synchronize()
    devs[0].import_tiles(gathered.data_ptr())
    assert same_values(devs[0].read_full(), want)
    for d in devs:
        d.close()


@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_hip_reproduces_golden_fixtures(pbr, device, name):
    kind, seed, tris, overrides, w, h, frames = make_golden.CASES[name]
    want = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    sc = make_scene(pbr, kind, seed, tris, **overrides)
    assert np.array_equal(sc.arrays()["bvh"], want["bvh"])
    device.upload_scene(sc.desc)
    device.configure(sc.config(w, h))
    device.render(0, pbr.frame_seeds(0, frames), float(want["px_dim"]), sc.camera())
    got = device.read_output()
    assert same_values(got, want["image"]), describe_mismatch(got, want["image"])
    assert same_values(device.read_debug(), want["debug"])
    c = device.counters()
    assert [c["nodes"], c["tris"], c["hits"], c["paths"]] == want["counters"].tolist()
    t, face, _, counts = device.diag_trace(want["rays"])
    assert same_values(t, want["ray_t"]) and np.array_equal(counts, want["ray_counts"])


def test_path_tracer_driver_matches_device_calls(pbr, oracle, gpu_device):
    """The C++ PathTracer (host/path_tracer.h) driven like GLWidget drives the reference's:
    initOpenCLBuffers, then generateImage per frame; and generateImages (fused)."""
    sc = make_scene(pbr, **{"render.max_depth": 4})
    w, h = 64, 40
    cfg, cam, px = sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h)
    want = oracle.Renderer(sc.desc, cfg, threads=8).render(0, pbr.frame_seeds(0, 4), px, cam)
    pt = pbr.PathTracer(gpu_device, w, h)
    pt.initOpenCLBuffers(sc)
    for _ in range(4):
        img = pt.generateImage()
    assert pt.sampleCount() == 4 and same_values(img, want)
    pt2 = pbr.PathTracer(gpu_device, w, h)
    pt2.initOpenCLBuffers(sc)
    assert same_values(pt2.generateImages(4), want)
    more = pt2.generateImages(3)                                    # continues the running mean at n = 4
    want7 = oracle.Renderer(sc.desc, cfg, threads=8).render(0, pbr.frame_seeds(0, 7), px, cam)
    assert same_values(more, want7)
    pt.close()
    pt2.close()


@pytest.mark.parametrize("cfg", [{"render.max_depth": 4}, {"render.max_depth": 3, "render.brdf": 0, "render.samples": 2},
                                 {"render.max_depth": 4, "hip.traversal": 2}])      # the ray-ordered walk, switched in the caller's configuration
def test_cl_adaptor_driven_like_the_reference(pbr, oracle, gpu_device, cfg):
    """host/cl_adaptor.h: class CL with the reference's public methods over the C ABI, driven by the exact
    call sequence of the reference's PathTracer (createBuffer x 7, setReplacement, images, loadProgram,
    createKernel, setKernelArg, then per frame updateImageReadOnly / execute / readImageOutput x 2)."""
    w, h = 64, 40
    sc = make_scene(pbr, **dict(cfg, **{"window.width": w, "window.height": h}))
    config, cam, px = sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h)
    # #SKY_LIGHT# travels as text printed with %f (PathTracer.cpp:466-472): the adaptor reads what the OpenCL compiler would
    for k in range(3):
        config.sky_light[k] = float(np.float32(float("%f" % config.sky_light[k])))
    ref = oracle.Renderer(sc.desc, config, threads=8)
    want = ref.render(0, pbr.frame_seeds(0, 5), px, cam)
    image, debug = sc.render_through_cl_adaptor(5)
    assert same_values(image, want), describe_mismatch(image, want)
    assert same_values(debug, ref.debug)
    # feeding the read-back buffer twice before a frame (the adaptor's on-device swap must happen once, the second feed is
    # an upload like the reference's): same frames
    image, debug = sc.render_through_cl_adaptor(5, refeed_every=2)
    assert same_values(image, want), describe_mismatch(image, want)


def check_flat_tree(nodes, facesV_out, facesV_in, vertices):
    """Structural invariants of the reference's flat format for a tree with <= 2 faces per leaf."""
    n = nodes.shape[0]
    leaf = nodes[:, 3] >= 0
    assert not leaf[0] and nodes[0, 3] == -1
    # every face exactly once, in leaf order
    first = nodes[leaf, 3].astype(np.int64)
    second = nodes[leaf, 7].astype(np.int64)
    assert ((second == -1) | (second == first + 1)).all()
    covered = np.sort(np.concatenate([first, second[second >= 0]]))
    assert np.array_equal(covered, np.arange(facesV_in.shape[0]))
    key = lambda f: np.sort(f.view([("", f.dtype)] * 4).ravel())
    assert np.array_equal(key(np.ascontiguousarray(facesV_out)), key(np.ascontiguousarray(facesV_in)))      # a permutation of the input
    # depth-first order: a container's subtree is [i + 1, end) with end = its miss link, or the enclosing end
    end = np.empty(n, np.int64)
    stack = [n]
    for i in range(n):
        while stack and i >= stack[-1]:
            stack.pop()
        enclosing = stack[-1] if stack else n
        if leaf[i]:
            end[i] = i + 1
        else:
            link = int(nodes[i, 7])
            assert link == -1 or i + 1 < link <= n
            end[i] = link if link != -1 else enclosing
            assert end[i] <= enclosing
            stack.append(end[i])
    # boxes: a leaf's box is the exact bound of its faces, a container's the bound of its subtree's leaves
    tri = vertices[facesV_out[:, :3].astype(np.int64), :3]                    # (m, 3, 3)
    flo, fhi = tri.min(1), tri.max(1)
    for i in np.nonzero(leaf)[0][:4000]:
        f0, f1 = int(nodes[i, 3]), int(nodes[i, 7])
        lo, hi = flo[f0], fhi[f0]
        if f1 >= 0:
            lo, hi = np.minimum(lo, flo[f1]), np.maximum(hi, fhi[f1])
        assert np.array_equal(nodes[i, 0:3], lo) and np.array_equal(nodes[i, 4:7], hi)
    for i in np.nonzero(~leaf)[0][1:2000]:
        sub = np.arange(i + 1, end[i])
        sub = sub[leaf[sub]]
        assert np.array_equal(nodes[i, 0:3], nodes[sub, 0:3].min(0)) and np.array_equal(nodes[i, 4:7], nodes[sub, 4:7].max(0))


@pytest.mark.parametrize("traversal", [0, 2])
@pytest.mark.parametrize("builder", ["ploc", "lbvh"])
@pytest.mark.parametrize("kind,triangles", [("cornell", 0), ("sponza", 6000), ("hairball", 30001)])
def test_device_bvh_build_emits_the_reference_format(pbr, oracle, device, kind, triangles, builder, traversal):
    """pbr_build_bvh: the tree built on the device (locally-ordered clustering, or round 1's radix tree with
    knob bvh_builder = 1) is a valid tree in the reference's flat format (structure, exact boxes, every face once);
    HIP and oracle agree bit for bit when both walk it — in the reference's order and in eight ray-ordered ones (round 5:
    the ordered walk is what makes such a tree, whose stored child order is arbitrary, as good as the host's in the reference's
    walk) —; and the hits are the geometric closest hits (brute force)."""
    device.set_knob("bvh_builder", {"ploc": 0, "lbvh": 1}[builder])
    sc = make_scene(pbr, kind, 5, triangles, **{"render.max_depth": 3, "hip.traversal": traversal})
    arr = sc.arrays()
    nodes, fv, fn = device.build_bvh(arr["vertices"], arr["facesV"], arr["facesN"])
    check_flat_tree(nodes, fv, arr["facesV"], arr["vertices"])
    again, fv2, _ = device.build_bvh(arr["vertices"], arr["facesV"], arr["facesN"])
    assert np.array_equal(again, nodes) and np.array_equal(fv2, fv)            # deterministic
    desc = pbr.SceneDesc.from_buffer_copy(sc.desc)
    desc.bvh, desc.num_nodes = nodes.ctypes.data, nodes.shape[0]
    desc.facesV, desc.facesN = fv.ctypes.data, fn.ctypes.data
    got, want, ref = both_render(pbr, oracle, device, sc, 64, 40, 3, desc=desc)
    assert same_values(got, want), describe_mismatch(got, want)
    assert device.counters() == ref.counter_dict()
    # closest hits against brute force over all triangles
    rng = np.random.default_rng(3)
    v = arr["vertices"][:, :3]
    rays = np.zeros((300, 6), np.float32)
    rays[:, 0:3] = rng.uniform(v.min(0), v.max(0), (300, 3))
    d = rng.normal(size=(300, 3)); rays[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    t, face, _, _ = device.diag_trace(rays)
    tri = v[fv[:, :3].astype(np.int64)].astype(np.float64)
    a, e1, e2 = tri[:, 0], tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    for k in range(rays.shape[0]):
        o, dd = rays[k, 0:3].astype(np.float64), rays[k, 3:6].astype(np.float64)
        p = np.cross(dd, e2); det = (e1 * p).sum(1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            tv = o - a; u = (tv * p).sum(1) * inv
            q = np.cross(tv, e1); vv = (q * dd).sum(1) * inv
            tt = (e2 * q).sum(1) * inv
        ok = (u >= -1e-6) & (vv >= -1e-6) & (u + vv <= 1 + 1e-6) & (tt > 1e-4) & np.isfinite(tt)
        best = tt[ok].min() if ok.any() else np.inf
        if np.isfinite(best) or np.isfinite(t[k]):
            assert abs(float(t[k]) - best) <= 1e-3 * max(1.0, best), (k, float(t[k]), best)


@pytest.mark.parametrize("faces", [1, 2, 3, 5, 64, 257])
def test_device_bvh_build_small_and_regular_inputs(pbr, device, faces):
    """1 and 2 faces (a single leaf under an added container), odd counts, and a regular strip of identical
    triangles — all pair distances tie, the rounds must still pair clusters up instead of chaining."""
    verts = np.zeros((faces + 2, 4), np.float32)
    verts[:, 0] = np.arange(faces + 2) // 2
    verts[:, 1] = np.arange(faces + 2) % 2
    fv = np.zeros((faces, 4), np.uint32)
    fv[:, 0], fv[:, 1], fv[:, 2] = np.arange(faces), np.arange(faces) + 1, np.arange(faces) + 2
    fn = np.zeros_like(fv)
    nodes, outV, outN = device.build_bvh(verts, fv, fn)
    assert nodes.shape[0] <= 2 * faces - 1 or faces == 1
    check_flat_tree(nodes, outV, fv, verts)
    if faces <= 2:
        assert nodes.shape[0] == 2 and nodes[1, 3] == 0 and nodes[1, 7] == (1 if faces == 2 else -1)


def test_device_bvh_build_rejects_what_it_cannot_order(pbr, device):
    verts = np.zeros((6, 4), np.float32)
    verts[:, 0] = np.nan
    fv = np.array([[0, 1, 2, 0], [3, 4, 5, 0], [0, 2, 4, 0]], np.uint32)
    with pytest.raises(pbr.PbrError):
        device.build_bvh(verts, fv, np.zeros_like(fv))


def smooth_scene(pbr, tmp_path, **cfg):
    """A UV sphere with per-vertex normals (curved: Phong tessellation applies) on a flat floor (three equal
    normals per face: the flat test applies), written as OBJ / MTL and loaded through the host loader + BVH."""
    rings, segs = 10, 16
    verts, norms, faces = [], [], []
    for i in range(rings + 1):
        th = np.pi * i / rings
        for j in range(segs):
            ph = 2 * np.pi * j / segs
            n = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
            verts.append(0.6 * n + [0.0, 0.75, 0.0]); norms.append(n)
    for i in range(rings):
        for j in range(segs):
            a, b = i * segs + j, i * segs + (j + 1) % segs
            c, d = a + segs, b + segs
            if i > 0:
                faces.append((a, b, c))
            if i < rings - 1:
                faces.append((b, d, c))
    lines = ["mtllib s.mtl", "o Ball"]
    lines += ["v %.9g %.9g %.9g" % tuple(v) for v in verts]
    lines += ["vn %.9g %.9g %.9g" % tuple(n) for n in norms]
    lines += ["usemtl Shiny"] + ["f %d//%d %d//%d %d//%d" % (a + 1, a + 1, b + 1, b + 1, c + 1, c + 1) for a, b, c in faces]
    k, kn = len(verts), len(norms)
    lines += ["o Floor", "v -2 0 -2", "v 2 0 -2", "v 2 0 2", "v -2 0 2", "vn 0 1 0", "usemtl Matte",
              "f %d//%d %d//%d %d//%d" % (k + 1, kn + 1, k + 3, kn + 1, k + 2, kn + 1),
              "f %d//%d %d//%d %d//%d" % (k + 1, kn + 1, k + 4, kn + 1, k + 3, kn + 1)]
    (tmp_path / "s.obj").write_text("\n".join(lines) + "\n")
    (tmp_path / "s.mtl").write_text("newmtl Shiny\nKd 0.8 0.3 0.2\nKs 0.9 0.9 0.9\nnu 200\nnv 200\nRs 0.4\nRd 0.6\n\nnewmtl Matte\nKd 0.6 0.6 0.7\n\nnewmtl sky_light\nKd 0.9 0.95 1.0\n")
    pbr.cfg_reset()
    pbr.cfg_set(**cfg)
    return pbr.HostScene.load_obj(str(tmp_path) + "/", "s.obj")


@pytest.mark.parametrize("traversal", [0, 1, 2, 3])
@pytest.mark.parametrize("brdf", [1, 0])
def test_phong_tessellation_bit_exact(pbr, oracle, device, tmp_path, brdf, traversal):
    """K19 (pt_phongtess.cl, PHONGTESS = 1): curved faces are intersected as Phong-tessellated patches (cubic +
    quadratics), flat ones as triangles; hits carry the patch normal into the shading.  Round 6: in the ray-ordered walks
    too (six orders, eight orders, eight orders over compact records) — against the oracle in the same walk."""
    sc = smooth_scene(pbr, tmp_path, **{"render.max_depth": 4, "render.brdf": brdf, "render.phong_tessellation": 0.6})
    arr = sc.arrays()
    tri_n = arr["normals"][arr["facesN"][:, :3].astype(np.int64), :3]
    curved = ~(np.all(tri_n[:, 0] == tri_n[:, 1], axis=1) & np.all(tri_n[:, 1] == tri_n[:, 2], axis=1))
    assert curved.sum() > 200 and (~curved).sum() >= 2
    w, h = 64, 48
    cfg = sc.config(w, h)
    cfg.traversal = traversal
    assert cfg.phong_tessellation == np.float32(0.6)
    got, want, ref = both_render(pbr, oracle, device, sc, w, h, 4, cfg=cfg)
    assert same_values(got, want), describe_mismatch(got, want)
    assert same_values(device.read_debug(), ref.debug)
    assert device.counters() == ref.counter_dict()
    assert device.last_plan()[0] == "refill-lean-phong"
    assert device.last_kernel().startswith("ptk_f%d::pathTracing<%d, false, false, 4, true>" % ({0: 0, 1: 1, 2: 1, 3: 5}[traversal], brdf))
    # the tessellation changes the picture: the same scene with PHONGTESS off renders differently
    flat = sc.config(w, h)
    flat.traversal = traversal
    flat.phong_tessellation = 0.0
    plain = oracle.Renderer(sc.desc, flat, threads=8).render(0, pbr.frame_seeds(0, 4), pbr.pixel_dimension(w, h), sc.camera())
    assert not same_values(plain, want)
    # Phong tessellation pins its own plan (include/pbr_hip.h): a pinned state-machine plan does not apply to it
    device.pin_plan(PLANS["phased-mid"])
    device.render(4, pbr.frame_seeds(4, 1), pbr.pixel_dimension(w, h), sc.camera())
    assert device.last_plan()[0] == "refill-lean-phong"


@pytest.mark.parametrize("seed", range(int(os.environ.get("PBR_SOAK_SEEDS", "256"))))
def test_random_configurations_bit_exact(pbr, oracle, device, seed):
    """Seeded differential sweep over the configuration space: scene kind / size, image shape, depths, samples,
    BRDF, anti-aliasing, lights + shadow rays, schedule, register budget, LDS share, frames per launch pair.
    256 seeds in the suite; PBR_SOAK_SEEDS=n runs the first n (a soak of 6000 is logged in profiles/r02/soak.txt)."""
    rng = np.random.default_rng(1000 + seed)
    kind = ["cornell", "sponza", "dragon", "hairball"][rng.integers(4)]
    tris = 0 if kind == "cornell" else int(rng.integers(300, 6000))
    brdf = int(rng.integers(2))
    cfg_keys = {
        "render.max_depth": int(rng.integers(1, 6)), "render.max_added_depth": int(rng.integers(0, 4)),
        "render.samples": int(rng.integers(1, 4)), "render.brdf": brdf,
        "render.antialiasing": float(rng.choice([0.0, 0.7, 1.5])),
    }
    schedule = [None, "refill-lean", "refill-wide", "phased-lean", "phased-wide", "phased-mid", "refill-mid"][rng.integers(7)]
    if seed % 5 == 4:       # (added in round 4 without moving the other draws of a seed: the soak logs stay comparable)
        schedule = "phased-dual"
    force_schedule(device, schedule)
    if rng.integers(3) == 0:
        device.set_knob("lds_slots", int(rng.integers(0, 200)))
    if rng.integers(2):
        device.set_knob("chunk_frames", int(rng.integers(1, 4)))
    if rng.integers(3) == 0:
        device.set_knob("drain_mode", int(rng.integers(0, 4)))
    if rng.integers(3) == 0:
        device.set_knob("refill_batch", int(rng.choice([1, 5, 40, 64])))
    w, h = 8 * int(rng.integers(1, 12)), 8 * int(rng.integers(1, 9))
    frames = int(rng.integers(1, 6))
    first = int(rng.integers(0, 3))
    sc = make_scene(pbr, kind, int(rng.integers(1, 100)), tris, **cfg_keys)
    cfg, desc, keep = sc.config(w, h), sc.desc, None
    if kind == "cornell" and rng.integers(2):
        lights = np.zeros((2, 12), np.float32)
        lights[0] = [0.1, 1.6, 0.2, 0, 4.0, 3.5, 3.0, 0, 2, 0.12, 0, 0]
        lights[1] = [-0.5, 0.4, 0.6, 0, 1, 1, 1, 0, 1, 0, 0, 0]
        desc = pbr.SceneDesc.from_buffer_copy(sc.desc)
        desc.lights, desc.num_lights = lights.ctypes.data, 2
        keep = lights
        cfg.shadow_rays = int(rng.integers(2))
    what = "%s tris=%d %dx%d frames=%d first=%d %s schedule=%s" % (kind, tris, w, h, frames, first, cfg_keys, schedule)
    got, want, ref = both_render(pbr, oracle, device, sc, w, h, frames, first=first, cfg=cfg, desc=desc)
    assert same_values(got, want), what + ": " + describe_mismatch(got, want)
    assert same_values(device.read_debug(), ref.debug), what
    assert device.counters() == ref.counter_dict(), what


@pytest.mark.parametrize("traversal", [0, 2])
def test_guard_build_with_the_cxx_node_phase_gives_the_same_bits(pbr, device, tmp_path, traversal):
    """libpbrhip_guard.so (-DPBR_GUARD) bounds every device loop and compiles traverse()'s node phase from C++ instead
    of the hand-scheduled block: same images, no guard trips — in the reference's walk and in the ray-ordered one (whose
    forward-only links the bounded traversal loop also holds to "every node at most once").  Run in a child process (the
    library is chosen at import)."""
    sc = make_scene(pbr, "sponza", 4, 9000, **{"render.max_depth": 3, "hip.traversal": traversal})
    w, h = 64, 40
    assert sc.config(w, h).traversal == traversal
    device.upload_scene(sc.desc)
    device.configure(sc.config(w, h))
    device.render(0, pbr.frame_seeds(0, 3), pbr.pixel_dimension(w, h), sc.camera())
    want = device.read_output()
    script = tmp_path / "guarded.py"
    script.write_text(
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import pbr_loader\n"
        "pbr = pbr_loader.load()\n"
        "pbr.cfg_reset(); pbr.cfg_set(**{'render.max_depth': 3, 'hip.traversal': %d})\n"
        "sc = pbr.HostScene.generate('sponza', 4, 9000)\n"
        "dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(sc.config(%d, %d))\n"
        "dev.render(0, pbr.frame_seeds(0, 3), pbr.pixel_dimension(%d, %d), sc.camera())\n"
        "assert dev.guard_trips() == [0, 0, 0], dev.guard_trips()\n"
        "first = dev.read_output()\n"
        "np.save(%r, first)\n"
        "for plan, mode in ((4, 3), (2, 0), (5, 1), (3, 2), (6, 1)):\n"           # the other schedules and how their launches end, every loop bounded
        "    dev.pin_plan(plan); dev.set_knob('drain_mode', mode); dev.reset_accum()\n"
        "    dev.render(0, pbr.frame_seeds(0, 3), pbr.pixel_dimension(%d, %d), sc.camera())\n"
        "    assert dev.guard_trips() == [0, 0, 0], (plan, dev.guard_trips())\n"
        "    assert np.array_equal(dev.read_output(), first, equal_nan=True), plan\n" % (ROOT, traversal, w, h, w, h, str(tmp_path / "guarded.npy"), w, h))
    env = dict(os.environ, PBR_GUARD="1", PBR_LAB_ENV="1")
    done = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    got = np.load(tmp_path / "guarded.npy")
    assert same_values(got, want), describe_mismatch(got, want)


def test_display_step_is_the_clamped_linear_image(pbr, device):
    """pbr_read_display: what shader/pathtracing.frag puts on an 8-bit framebuffer, converted on the device."""
    sc = make_scene(pbr, **{"render.max_depth": 3})
    w, h = 72, 40
    device.upload_scene(sc.desc)
    device.configure(sc.config(w, h))
    rng = np.random.default_rng(5)
    img = rng.uniform(-0.5, 1.5, (h, w, 4)).astype(np.float32)
    img[3, 5, 0], img[4, 6, 1], img[0, 0, 2] = np.nan, np.inf, -np.inf
    device.write_input(img)
    device.render_frame(0.0333, 1.0, pbr.pixel_dimension(w, h), sc.camera())     # weight 1: imageOut ~ the injected image
    lin = device.read_output()
    assert np.isnan(lin[3, 5, 0]) and lin[4, 6, 1] == np.inf and (lin[..., :3] > 1.0).any() and (lin[..., :3] < 0.0).any()
    want = np.floor(np.clip(np.nan_to_num(lin[..., :3], nan=0.0, posinf=1.0, neginf=0.0), 0.0, 1.0).astype(np.float32) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
    got = device.read_display()
    assert np.array_equal(got[..., :3], want) and (got[..., 3] == 255).all()
    assert np.array_equal(device.read_display(top_row_first=True), got[::-1])


# ----------------------------------------------------------------------------------------------
# error behaviour of the boundary
# ----------------------------------------------------------------------------------------------

def test_call_sequence_and_validation_errors(pbr, device):
    sc = make_scene(pbr)
    cam, px = sc.camera(), pbr.pixel_dimension(64, 64)
    with pytest.raises(pbr.PbrError, match="before"):
        device.render(0, pbr.frame_seeds(0, 1), px, cam)
    device.upload_scene(sc.desc)
    with pytest.raises(pbr.PbrError, match="before"):
        device.render(0, pbr.frame_seeds(0, 1), px, cam)

    cfg = sc.config(60, 64)
    with pytest.raises(pbr.PbrError, match="multiples of 8"):
        device.configure(cfg)
    cfg = sc.config(64, 64)
    cfg.phong_tessellation = 0.5
    device.configure(cfg)                                    # Phong tessellation is built (K19); flat scenes just never take its path
    cfg = sc.config(64, 64)
    cfg.brdf = 0
    device.configure(cfg)
    with pytest.raises(pbr.PbrError, match="BRDF"):
        device.render(0, pbr.frame_seeds(0, 1), px, cam)

    # a face without `usemtl` carries material -1 (ObjParser.cpp:192): out of bounds in the reference
    arr = sc.arrays()
    bad = arr["facesV"].copy()
    bad[3, 3] = 0xFFFFFFFF
    d = pbr.SceneDesc.from_buffer_copy(sc.desc)
    d.facesV = bad.ctypes.data
    with pytest.raises(pbr.PbrError, match="material index"):
        device.upload_scene(d)
    links = arr["bvh"].copy()
    links[2, 7] = 1e6
    d = pbr.SceneDesc.from_buffer_copy(sc.desc)
    d.bvh = links.ctypes.data
    with pytest.raises(pbr.PbrError, match="node 2"):
        device.upload_scene(d)
    leaves = np.nonzero(arr["bvh"][:, 7] > arr["bvh"][:, 3])[0]                # leaves with two faces
    pair = arr["bvh"].copy()
    pair[leaves[0], 7] += 1                                                      # second face must be first + 1
    d = pbr.SceneDesc.from_buffer_copy(sc.desc)
    d.bvh = pair.ctypes.data
    with pytest.raises(pbr.PbrError, match="first face \\+ 1"):
        device.upload_scene(d)
    d = pbr.SceneDesc.from_buffer_copy(sc.desc)
    d.num_nodes = 1
    with pytest.raises(pbr.PbrError, match="root is never tested"):
        device.upload_scene(d)
    # an array that ends in a container (children outside the array) cannot be walked
    last = arr["bvh"].copy()
    last[-1, 3], last[-1, 7] = -1.0, -1.0
    d = pbr.SceneDesc.from_buffer_copy(sc.desc)
    d.bvh = last.ctypes.data
    with pytest.raises(pbr.PbrError, match="last node is a container"):
        device.upload_scene(d)


# ----------------------------------------------------------------------------------------------
# BASELINE.json sizes: size-independent properties at 1920 x 1080
# ----------------------------------------------------------------------------------------------

def test_full_hd_properties(pbr, oracle, device):
    """Config 2 (Cornell, 1920x1080, depth 8) at a few frames: deterministic; continuing a render
    equals one longer render; paths = W*H*frames; and the whole frame equals the oracle's."""
    sc = make_scene(pbr, **{"render.max_depth": 8})
    w, h = 1920, 1080
    cfg, cam, px = sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h)
    device.upload_scene(sc.desc)
    device.configure(cfg)
    device.render(0, pbr.frame_seeds(0, 6), px, cam)
    a = device.read_output()
    assert device.counters()["paths"] == w * h * 6
    device.reset_accum()
    device.render(0, pbr.frame_seeds(0, 6), px, cam)
    assert same_values(device.read_output(), a)                     # run-to-run identical
    device.reset_accum()
    device.render(0, pbr.frame_seeds(0, 2), px, cam)
    device.render(2, pbr.frame_seeds(2, 4), px, cam)
    assert same_values(device.read_output(), a)                     # 2 + 4 frames == 6 frames
    # the whole 1920 x 1080 frame, six frames deep, against the oracle (Cornell at depth 8 is the oracle's slowest
    # configuration per sample and still seconds on the GPU box's host threads)
    ref = oracle.Renderer(sc.desc, cfg, threads=os.cpu_count() or 8)
    want = ref.render(0, pbr.frame_seeds(0, 6), px, cam)
    assert same_values(a, want), describe_mismatch(a, want)
    assert same_values(device.read_debug(), ref.debug)
    assert np.isfinite(a[..., :3]).all() and a[..., :3].min() >= 0.0


def test_full_hd_sharded_equals_unsharded(pbr, device):
    """Config 4's invariant at full size: 8-way tile shards, gathered, are the 1-GPU frame."""
    import torch
    sc = make_scene(pbr, "sponza", 2, 30000)
    w, h, world = 1920, 1080, 8
    cam, px, seeds = sc.camera(), pbr.pixel_dimension(w, h), pbr.frame_seeds(0, 2)
    device.upload_scene(sc.desc)
    device.configure(sc.config(w, h))
    device.render(0, seeds, px, cam)
    want = device.read_output()
    gathered = torch.zeros(world * (w * h * 16 // world) // 4, dtype=torch.float32, device="cuda")
    d = pbr.Device(0)
    d.upload_scene(sc.desc)
    for rank in range(world):
        cfg = sc.config(w, h)
        cfg.tile_world, cfg.tile_rank = world, rank
        d.configure(cfg)
        d.render(0, seeds, px, cam)
        assert d.tile_bytes() == w * h * 16 // world
        d.export_tiles(gathered.data_ptr() + rank * d.tile_bytes())
    torch.cuda.synchronize()
    d.import_tiles(gathered.data_ptr())
    assert same_values(d.read_full(), want)
    d.close()


@pytest.mark.parametrize("mode", [[], ["--traversal", "eight-order", "--arith", "native"]])
def test_bench_two_ranks_rehearsal_matches_one_rank(tmp_path, mode):
    """bench.py's N > 1 path (tile sharding + all-gather + scatter) rehearsed with two processes on
    ONE GPU (gloo): the gathered frame is the single-rank frame, bit for bit, and every path is counted — in the default
    mode and with both opt-in modes on (every plan of a mode renders the same bits, so the vote's outcome does not show)."""
    import json
    import subprocess
    common = ["--steps", "3", "--warmup", "1", "--width", "256", "--height", "144", "--cpu-seconds", "0", "--repeats", "2", "--triangles", "20000",
              "--hold-seconds", "0"] + mode
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dump", str(tmp_path / "one.npy")] + common,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--backend", "gloo", "--one-device", "--dump", str(tmp_path / "two.npy")] + common,
                         capture_output=True, text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    j2 = json.loads(two.stdout.strip().splitlines()[-1])
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2
    assert j1["per_sample"] == j2["per_sample"]                     # same work, counted once
    assert j2["roofline"]["algorithmic_GBs"] > 0 and j2["value"] > 0
    assert j1["repeats"] == j2["repeats"] == 2 and len(j2["per_rank_ms"]["render"]) == 2
    assert j1["config"]["scene"] == "sponza" and j2["schedule_tuned"]     # the elected plan, pinned on both ranks
    assert j2["config"]["traversal"] == ("eight-order" if mode else "reference") and j2["config"]["arith"] == ("native" if mode else "exact")
    assert j2["roofline"]["kernel"].startswith("ptk_f3::" if mode else "ptk_f0::")
    assert len(j2["plan_votes"]) == 2 and j2["schedule"] in [pbr_plan_name(v) for v in j2["plan_votes"]]
    assert same_values(np.load(tmp_path / "one.npy"), np.load(tmp_path / "two.npy"))


def pbr_plan_name(index):
    return ("refill-lean", "refill-wide", "phased-lean", "phased-wide", "phased-mid", "refill-mid", "phased-dual")[index]


def test_bench_starts_its_own_ranks(tmp_path):
    """The driver's command shape: `python bench.py --gpus N ...` with no launcher around it.  bench.py must start its N
    ranks itself (before anything touches the GPU in the parent) and print ONE line: here N = 2 on one GPU over gloo."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device",
                          "--steps", "3", "--warmup", "1", "--width", "256", "--height", "144", "--cpu-seconds", "0", "--triangles", "20000",
                          "--dump", str(tmp_path / "two.npy")],
                         capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1
    assert len(j["per_rank_ms"]["render"]) == 2 and len(j["per_rank_ms"]["gather"]) == 2
    assert j["value"] > 0 and j["scaling"] == "strong"


def test_environment_steers_lab_runs_only(pbr, gpu_device, cfg_defaults, monkeypatch):
    """the C library reads no environment variable, and the Python harness maps PBR_* variables onto knobs only
    for a process that says it is a lab run (PBR_LAB_ENV=1): a stray PBR_PLAN in a user's environment changes nothing."""
    sc = pbr.HostScene.generate("cornell", 1, 0)
    cfg, cam, px = sc.config(64, 48), sc.camera(), pbr.pixel_dimension(64, 48)

    def first_plan():
        dev = pbr.Device(gpu_device)
        dev.upload_scene(sc.desc)
        dev.configure(cfg)
        dev.render(0, pbr.frame_seeds(0, 1), px, cam)
        plan = dev.last_plan()
        dev.close()
        return plan

    monkeypatch.delenv("PBR_LAB_ENV", raising=False)
    monkeypatch.setenv("PBR_PLAN", "3")
    monkeypatch.setenv("PBR_BVH_BUILDER", "no-such-builder")         # would raise if it were looked at
    assert first_plan()[0] == pbr.Device.PLAN_NAMES[0]               # the tuner's first candidate: nothing was pinned
    monkeypatch.setenv("PBR_LAB_ENV", "1")
    with pytest.raises(pbr.PbrError):                                # a lab run: the variables count, and a bad value is an error, not a KeyError
        first_plan()
    monkeypatch.setenv("PBR_BVH_BUILDER", "lbvh")
    assert first_plan() == (pbr.Device.PLAN_NAMES[3], -1)            # pinned by PBR_PLAN


def test_bench_full_scale_shape_rehearsed_on_one_device(tmp_path):
    """the SHAPE of the driver's 8-GPU run on the hardware there is — `python3 bench.py --gpus 8`
    with no launcher, the driver's --steps 20 --warmup 5, the Sponza-class scene at 1920x1080, eight ranks on ONE device
    over gloo.  Eight concurrent imports behind the build lock, eight contexts on one device, eight tuners and the
    vote, the all-gather and the scatter: one line, n_gpus 8, eight per-rank timings, eight votes, and the gathered frame
    is the one-rank frame bit for bit.  No 8-GPU number is claimed from this: the ranks share one GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    common = ["--steps", "20", "--warmup", "5", "--cpu-seconds", "0", "--repeats", "1", "--hold-seconds", "0"]
    eight = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--one-device",
                            "--dump", str(tmp_path / "eight.npy")] + common, capture_output=True, text=True, timeout=1100, env=env)
    assert eight.returncode == 0, eight.stderr[-3000:]
    lines = [ln for ln in eight.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, eight.stdout[-2000:]
    j8 = json.loads(lines[0])
    assert j8["n_gpus"] == 8 and j8["steps"] == 20 and j8["warmup"] == 5 and j8["config"]["scene"] == "sponza"
    assert (j8["config"]["width"], j8["config"]["height"]) == (1920, 1080)
    assert len(j8["per_rank_ms"]["render"]) == 8 and len(j8["per_rank_ms"]["gather"]) == 8 and all(v > 0 for v in j8["per_rank_ms"]["render"])
    assert len(j8["plan_votes"]) == 8 and j8["schedule"] in [pbr_plan_name(v) for v in j8["plan_votes"] if v >= 0]
    assert j8["value"] > 0 and j8["scaling"] == "strong"
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dump", str(tmp_path / "one.npy")] + common,
                         capture_output=True, text=True, timeout=600, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert j1["per_sample"] == j8["per_sample"]                     # the same paths, each counted once, on whichever rank
    assert same_values(np.load(tmp_path / "one.npy"), np.load(tmp_path / "eight.npy"))


def test_bench_runs_the_rccl_leg_with_one_rank(tmp_path):
    """the `nccl` leg had never executed on an MI355X in any form (bench.py skipped torch.distributed
    when world == 1, every rehearsal went through gloo and host memory).  `--force-dist`: init_process_group("nccl",
    world_size=1) — a real RCCL communicator —, all_gather_into_tensor on the device buffers of pbr_export_tiles /
    pbr_import_tiles, scatterGathered, the plan vote and the control reductions through RCCL; the gathered frame must be
    the rendered frame at 1920 x 1080 (bench.py asserts it and says so in the line) and the run's frame the plain run's."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    common = ["--steps", "8", "--warmup", "4", "--cpu-seconds", "0", "--repeats", "2", "--hold-seconds", "0"]
    forced = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--backend", "nccl",
                             "--dump", str(tmp_path / "dist.npy")] + common, capture_output=True, text=True, timeout=900, env=env)
    assert forced.returncode == 0, forced.stderr[-3000:]
    # ONE line on stdout, although RCCL prints its version banner to the C-level stdout when the communicator is created
    assert len(forced.stdout.strip().splitlines()) == 1, forced.stdout[-2000:]
    line = json.loads(forced.stdout.strip())
    assert line["n_gpus"] == 1 and (line["config"]["width"], line["config"]["height"]) == (1920, 1080)
    assert line["force_dist"] == {"backend": "nccl", "world_size": 1, "gathered_frame_equals_rendered": True, "gather_ms": line["force_dist"]["gather_ms"]}
    assert line["force_dist"]["gather_ms"] > 0 and len(line["per_rank_ms"]["render"]) == 1 and len(line["plan_votes"]) == 1
    assert line["roofline"]["kernel"].startswith("ptk_f0::pathTracing")
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dump", str(tmp_path / "plain.npy")] + common,
                           capture_output=True, text=True, timeout=900, env=env)
    assert plain.returncode == 0, plain.stderr[-3000:]
    assert same_values(np.load(tmp_path / "dist.npy"), np.load(tmp_path / "plain.npy"))
    # round 6: a plain one-GPU run of the default mode also times the two opt-in modes on the same workload and steps — extra
    # keys of the SAME single line; `value` stays the default mode's
    assert len(plain.stdout.strip().splitlines()) == 1, plain.stdout[-2000:]
    plain_line = json.loads(plain.stdout.strip())
    assert (plain_line["config"]["traversal"], plain_line["config"]["arith"]) == ("reference", "exact")
    assert plain_line["roofline"]["kernel"].startswith("ptk_f0::") and plain_line["deal"] in ("spatial", "cost-classes", "expensive-last")
    assert sorted(plain_line["modes"]) == ["eight-order", "eight-order+native"]
    for key, flavour, parity in (("eight-order", "ptk_f1::", "tolerance"), ("eight-order+native", "ptk_f3::", "statistical")):
        leg = plain_line["modes"][key]
        assert leg["parity"] == parity and leg["repeats"] >= 5 and leg["value"] > 0 and leg["unit"] == "Msamples/s"
        assert leg["roofline"]["kernel"].startswith(flavour), leg["roofline"]["kernel"]
        assert leg["per_sample"]["node_visits"] < plain_line["per_sample"]["node_visits"]       # the ordered walk visits fewer nodes
        assert leg["scene_device_bytes"]["walk_streams"] > 0
    assert "modes" not in line                                     # the collective leg's line is the default mode alone


def test_configs_0_as_baseline_json_writes_it(pbr, oracle, device):
    """BASELINE.json configs[0]: "Cornell box 256 x 256, 4 spp, depth 4" — the reference's own CPU-runnable case, at its own
    size (it had only been run at 64 x 48 and 64 x 64): the whole frame, the debug image and the counters."""
    sc = make_scene(pbr, **{"render.max_depth": 4})
    got, want, ref = both_render(pbr, oracle, device, sc, 256, 256, 4)
    assert same_values(got, want), describe_mismatch(got, want)
    assert same_values(device.read_debug(), ref.debug)
    assert device.counters() == ref.counter_dict() and device.counters()["paths"] == 256 * 256 * 4
    assert np.isfinite(want[..., 3]).mean() > 0.5


# ----------------------------------------------------------------------------------------------
# the reference's own scenes (resources/models/testing/*.obj|.mtl|.lights) as committed fixtures
# ----------------------------------------------------------------------------------------------

import make_reference_scenes  # noqa: E402


@pytest.mark.parametrize("plan", [None, 0, 3, 4, 6])
@pytest.mark.parametrize("name", sorted(make_reference_scenes.CASES))
def test_hip_renders_the_reference_scenes(pbr, device, name, plan):
    """reference-authored geometry, material sets (glass d = 0 in pillars / spheres — K13 on whole
    images —, the nu = nv = 100000 lobes of suzanne.mtl, `light` flags) and suzanne.lights with shadow rays, BRDF 0 and
    1.  Inputs = the seven wire-format arrays + kernel constants + camera stored in tests/golden/ref_*.npz (made from
    the reference's files by make_reference_scenes.py; nothing is read from /root/reference here); expected = the
    oracle's image, debug image, counters and a 4096-ray closest-hit batch.  Tuner (None) and four forced plans."""
    if plan is not None:
        device.pin_plan(plan)
    data = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    desc, cfg, cam, keep = make_reference_scenes.scene_from_fixture(pbr, data)
    device.upload_scene(desc)
    device.configure(cfg)
    device.render(0, data["seeds"], float(data["px_dim"]), cam)
    got = device.read_output()
    assert same_values(got, data["image"]), describe_mismatch(got, data["image"])
    assert same_values(device.read_debug(), data["debug"])
    c = device.counters()
    assert [c["nodes"], c["tris"], c["hits"], c["paths"]] == data["counters"].tolist()
    t, face, _, counts = device.diag_trace(data["rays"])
    assert same_values(t, data["ray_t"]) and np.array_equal(counts, data["ray_counts"])
    hit = np.isfinite(data["ray_t"])
    assert hit.sum() > 1000 and np.array_equal(face[hit], data["ray_face"][hit])
    assert device.guard_trips() == [0, 0, 0]


def test_pinned_plan_is_the_plan_that_renders(pbr, oracle, device):
    """pbr_diag_pin_plan (what the ranks of a multi-GPU run do with rank 0's choice): the pinned schedule renders every
    launch, no tuning launches; -1 hands the choice back to the tuner; the pin survives upload / configure; the bits
    do not depend on it."""
    sc = make_scene(pbr, "sponza", 3, 9000, **{"render.max_depth": 3})
    w, h = 72, 48
    cfg, cam, px = sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h)
    want = oracle.Renderer(sc.desc, cfg, threads=8).render(0, pbr.frame_seeds(0, 5), px, cam)
    for plan in (2, 5, 6):
        device.pin_plan(plan)
        device.upload_scene(sc.desc)
        device.configure(cfg)
        device.render(0, pbr.frame_seeds(0, 2), px, cam)
        assert device.last_plan() == (pbr.Device.PLAN_NAMES[plan], -1)       # pinned, and the tuner has not run
        assert device.last_trace()[1] == 1                                    # one launch: no tuning chunks
        device.render(2, pbr.frame_seeds(2, 3), px, cam)
        assert device.last_plan()[0] == pbr.Device.PLAN_NAMES[plan]
        assert same_values(device.read_output(), want)
    device.pin_plan(-1)
    device.reset_accum()
    device.render(0, pbr.frame_seeds(0, 5), px, cam)
    assert device.last_plan()[0] == pbr.Device.PLAN_NAMES[0]                  # the tuner is back: it screens its first candidate on these frames
    assert same_values(device.read_output(), want)
    with pytest.raises(pbr.PbrError):
        device.pin_plan(9)
