"""The ray-ordered walk on the GPU (pbr_config.traversal = PBR_WALK_SIX_ORDERS / PBR_WALK_EIGHT_ORDERS): the HIP path
against the oracle IN THE SAME MODE, bit for bit — per ray (distance, face, normal, node and face-test counts), whole
images with the debug image and the counters in every plan, lights and shadow rays, both BRDFs, depth of field, tile
shards, a seeded sweep of random configurations.  What the mode owes the reference's own order (the same image within
SURVEY.md section 8(c)'s tolerance) is held on the CPU by tests/test_walk_order_cpu.py and at full size by
tests/test_gpu_full_configs.py.  The reference walks one fixed order: pt_bvh.cl:82-123, :102,112."""
import os

import numpy as np
import pytest

from conftest import same_values, describe_mismatch
from test_gpu_parity import PLANS, both_render, device, force_schedule, make_scene  # noqa: F401  (device: fixture)

pytestmark = pytest.mark.gpu

MODES = {1: "six orders", 2: "eight orders", 3: "eight orders, compact records"}     # 3: the same visits as 2 in one 64-byte record per node (round 6)


def cornell_lights(pbr, sc):
    lights = np.zeros((2, 12), np.float32)
    lights[0] = [0.1, 1.6, 0.2, 0, 4.0, 3.5, 3.0, 0, 2, 0.12, 0, 0]
    lights[1] = [-0.5, 0.4, 0.6, 0, 1, 1, 1, 0, 1, 0, 0, 0]
    desc = pbr.SceneDesc.from_buffer_copy(sc.desc)
    desc.lights, desc.num_lights = lights.ctypes.data, 2
    return desc, lights


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("kind,triangles,skip", [("cornell", 0, True), ("cornell", 0, False), ("sponza", 8000, True), ("dragon", 8000, True), ("hairball", 6000, True)])
def test_ordered_traversal_bit_exact(pbr, oracle, device, mode, kind, triangles, skip):
    """Per ray: the walk of the configured mode through pbr_diag_trace against orc_trace_rays in that mode."""
    sc = make_scene(pbr, kind, 2, triangles, **{"bvh.skip_ahead": skip})
    cfg = sc.config(64, 64)
    cfg.traversal = mode
    v = sc.arrays()["vertices"][:, :3]
    rng = np.random.default_rng(3)
    rays = np.concatenate([rng.uniform(v.min(0) - 0.3, v.max(0) + 0.3, (5000, 3)), rng.normal(size=(5000, 3))], axis=1).astype(np.float32)
    rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    rays[:9, 3:] = [[1, 0, 0], [0, 1, 0], [0, 0, -1], [0, -1, 0], [1, 1, 0], [0, 0, 1], [-1, 0, 0], [1, -1, 0], [-1, 1, 1]]   # ties of the dominant axis, zero and -0 components
    rays[5, 3] = -0.0
    device.upload_scene(sc.desc)
    device.configure(cfg)
    t, face, normal, counts = device.diag_trace(rays)
    ot, oface, onormal, ocounts = oracle.trace_rays(sc.desc, cfg, rays)
    assert same_values(t, ot), describe_mismatch(t, ot)
    hit = np.isfinite(ot)
    assert hit.sum() > 500
    assert np.array_equal(face[hit], oface[hit]) and same_values(normal[hit], onormal[hit])
    assert np.array_equal(counts, ocounts)
    # ... and it is another walk than the reference's: the per-ray node counts differ
    cfg.traversal = 0
    assert not np.array_equal(oracle.trace_rays(sc.desc, cfg, rays)[3], ocounts)


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("plan", sorted(PLANS))
@pytest.mark.parametrize("kind,triangles,keys,lit", [
    ("cornell", 0, {"render.max_depth": 5, "render.max_added_depth": 2}, True),
    ("sponza", 14000, {"render.max_depth": 3}, False),
    ("dragon", 12000, {"render.max_depth": 3, "render.brdf": 0}, False),
    ("hairball", 9000, {"render.max_depth": 3, "render.samples": 2}, False),
])
def test_ordered_walk_image_bit_exact_in_every_plan(pbr, oracle, device, mode, plan, kind, triangles, keys, lit):
    """Image, debug image (this walk's own per-pixel counters) and launch counters, every plan; Cornell with an orb light,
    a point light and shadow rays (the any-hit walk takes the ray's order too)."""
    device.pin_plan(PLANS[plan])
    sc = make_scene(pbr, kind, 7, triangles, **keys)
    cfg, desc, keep = sc.config(88, 56), sc.desc, None
    cfg.traversal = mode
    if lit:
        desc, keep = cornell_lights(pbr, sc)
        cfg.shadow_rays = 1
    got, want, ref = both_render(pbr, oracle, device, sc, 88, 56, 5, cfg=cfg, desc=desc)
    assert device.last_plan()[0] == plan
    assert same_values(got, want), describe_mismatch(got, want)
    assert same_values(device.read_debug(), ref.debug)
    assert device.counters() == ref.counter_dict()
    assert device.guard_trips() == [0, 0, 0]


@pytest.mark.parametrize("mode", sorted(MODES))
def test_the_tuner_and_a_mode_switch(pbr, oracle, device, mode):
    """No plan pinned: the tuner screens its plans in the mode's kernels.  Reconfiguring the same context with another
    traversal renders that mode (its streams are built on first use and kept)."""
    sc = make_scene(pbr, "sponza", 5, 12000, **{"render.max_depth": 3})
    w, h = 64, 48
    cam, px, seeds = sc.camera(), pbr.pixel_dimension(w, h), pbr.frame_seeds(0, 30)
    device.upload_scene(sc.desc)
    images = {}
    for m in (mode, 0, mode):
        cfg = sc.config(w, h)
        cfg.traversal = m
        device.configure(cfg)
        before = device.counters()
        device.render(0, seeds, px, cam)
        ref = oracle.Renderer(sc.desc, cfg, threads=8)
        want = ref.render(0, seeds, px, cam)
        got = device.read_output()
        assert same_values(got, want), "traversal %d: %s" % (m, describe_mismatch(got, want))
        spent = {k: v - before[k] for k, v in device.counters().items()}
        assert spent == ref.counter_dict()
        images[m] = spent
    assert images[mode]["nodes"] < images[0]["nodes"]
    assert device.last_kernel().startswith("ptk_f%d::" % {1: 1, 2: 1, 3: 5}[mode])
    # what the mode costs in node memory: six / eight streams of the same records + the 32-byte table of first references —
    # or, compact, ONE 64-byte record per node (twice the reference stream) + the table
    mem = device.scene_bytes()
    assert mem["nodes"] == 32 * sc.desc.num_nodes
    if mode == 3:
        assert mem["walk_streams"] == 2 * mem["nodes"] + 32 and mem["walk_streams"] <= 2.1 * mem["nodes"], mem
    else:
        orders = {1: 6, 2: 8}[mode]
        assert mem["walk_streams"] == orders * (mem["nodes"] - 32) + 32 + 32, mem


@pytest.mark.parametrize("mode", sorted(MODES))
def test_ordered_walk_tile_shards_reassemble(pbr, oracle, gpu_device, mode):
    sc = make_scene(pbr, "dragon", 3, 10000, **{"render.max_depth": 3})
    w, h, world = 72, 48, 3
    cfg, cam, px, seeds = sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h), pbr.frame_seeds(0, 4)
    cfg.traversal = mode
    ref = oracle.Renderer(sc.desc, cfg, threads=8)
    want = ref.render(0, seeds, px, cam)
    got = np.zeros((h, w, 4), np.float32)
    total = {"nodes": 0, "tris": 0, "hits": 0, "paths": 0}
    for r in range(world):
        dev = pbr.Device(gpu_device)
        c = pbr.Config.from_buffer_copy(cfg)
        c.tile_world, c.tile_rank = world, r
        dev.upload_scene(sc.desc)
        dev.configure(c)
        dev.render(0, seeds, px, cam)
        got += dev.read_output()
        for k, v in dev.counters().items():
            total[k] += v
        dev.close()
    assert same_values(got, want), describe_mismatch(got, want)
    assert total == ref.counter_dict()


@pytest.mark.parametrize("mode", sorted(MODES))
def test_ordered_walk_depth_of_field_and_frame_by_frame(pbr, oracle, device, mode):
    """The reference's per-frame call sequence (render_frame + accumulate) with a focus point, in an ordered mode."""
    sc = make_scene(pbr, **{"render.max_depth": 3})
    w, h = 48, 40
    cfg, cam, px = sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h)
    cfg.traversal = mode
    cam.focusPoint[0], cam.focusPoint[1] = 20, 17
    ref = oracle.Renderer(sc.desc, cfg, threads=8)
    device.upload_scene(sc.desc)
    device.configure(cfg)
    for k, seed in enumerate(pbr.frame_seeds(0, 4)):
        weight = float(np.float32(k) / np.float32(k + 1))
        ref.image = ref.render_frame(float(seed), weight, px, cam)
        device.render_frame(float(seed), weight, px, cam)
        got = device.read_output()
        assert same_values(got, ref.image), "frame %d: %s" % (k, describe_mismatch(got, ref.image))
        device.accumulate()


def test_mode_validation(pbr, device, tmp_path):
    sc = make_scene(pbr)
    device.upload_scene(sc.desc)
    cfg = sc.config(16, 16)
    cfg.traversal = 4
    with pytest.raises(pbr.PbrError, match="traversal"):
        device.configure(cfg)
    cfg.traversal, cfg.arith = 0, 2
    with pytest.raises(pbr.PbrError, match="arith"):
        device.configure(cfg)
    cfg.traversal, cfg.arith, cfg.phong_tessellation = 1, 0, 0.5
    device.configure(cfg)                          # Phong tessellation in a ray-ordered walk: refused up to round 5, a mode like any other since


@pytest.mark.parametrize("seed", range(int(os.environ.get("PBR_WALK_SOAK_SEEDS", "96"))))
def test_random_configurations_in_an_ordered_mode(pbr, oracle, device, seed):
    """The seeded differential sweep of test_gpu_parity.py in the two ordered modes: scene kind / size, image shape,
    depths, samples, BRDF, anti-aliasing, lights + shadow rays, plan, LDS share, frames per launch pair.  96 seeds in the
    suite; PBR_WALK_SOAK_SEEDS=n runs the first n (a soak of 30 000 is logged in profiles/r05/soak_walk.txt)."""
    rng = np.random.default_rng(77000 + seed)
    mode = 1 + seed % 3
    kind = ["cornell", "sponza", "dragon", "hairball"][rng.integers(4)]
    tris = 0 if kind == "cornell" else int(rng.integers(300, 6000))
    brdf = int(rng.integers(2))
    keys = {
        "render.max_depth": int(rng.integers(1, 6)), "render.max_added_depth": int(rng.integers(0, 4)),
        "render.samples": int(rng.integers(1, 4)), "render.brdf": brdf,
        "render.antialiasing": float(rng.choice([0.0, 0.7, 1.5])), "bvh.skip_ahead": bool(rng.integers(2)),
    }
    schedule = [None, "refill-lean", "refill-wide", "phased-lean", "phased-wide", "phased-mid", "refill-mid", "phased-dual"][rng.integers(8)]
    force_schedule(device, schedule)
    if rng.integers(3) == 0:
        device.set_knob("lds_slots", int(rng.integers(0, 200)))
    if rng.integers(2):
        device.set_knob("chunk_frames", int(rng.integers(1, 4)))
    w, h = 8 * int(rng.integers(1, 12)), 8 * int(rng.integers(1, 9))
    frames, first = int(rng.integers(1, 6)), int(rng.integers(0, 3))
    sc = make_scene(pbr, kind, int(rng.integers(1, 100)), tris, **keys)
    cfg, desc, keep = sc.config(w, h), sc.desc, None
    cfg.traversal = mode
    if kind == "cornell" and rng.integers(2):
        desc, keep = cornell_lights(pbr, sc)
        cfg.shadow_rays = int(rng.integers(2))
    what = "%s tris=%d %dx%d frames=%d first=%d %s schedule=%s traversal=%d" % (kind, tris, w, h, frames, first, keys, schedule, mode)
    got, want, ref = both_render(pbr, oracle, device, sc, w, h, frames, first=first, cfg=cfg, desc=desc)
    assert same_values(got, want), what + ": " + describe_mismatch(got, want)
    assert same_values(device.read_debug(), ref.debug), what
    assert device.counters() == ref.counter_dict(), what


def test_a_modes_streams_live_and_die_with_its_configuration(pbr, device):
    """what a mode needs is prepared by the call that completes "scene + mode" — pbr_configure with a scene
    present, or pbr_upload_scene into a configured context — so that this call is the one that can fail, not a render in the
    middle of a viewer's loop; and the streams of a traversal that is no longer configured are freed."""
    sc = make_scene(pbr, "sponza", 5, 9000)
    cfg = sc.config(64, 48)
    device.upload_scene(sc.desc)
    nodes = 32 * sc.desc.num_nodes
    for mode, expect in ((2, 8 * (nodes - 32) + 64), (3, 2 * nodes + 32), (0, 0), (1, 6 * (nodes - 32) + 64), (0, 0)):
        cfg.traversal = mode
        device.configure(cfg)                                  # no render yet
        assert device.scene_bytes()["walk_streams"] == expect, (mode, device.scene_bytes())
    cfg.traversal = 3                                            # configured first, scene second: the upload builds them
    device.configure(cfg)
    device.upload_scene(sc.desc)
    assert device.scene_bytes()["walk_streams"] == 2 * nodes + 32


def test_the_device_builder_says_which_radius_it_used(pbr, device):
    """pbr_build_bvh clusters with radius 3 for a tree that will be walked in ray order and 32 for the reference's
    walk — by the traversal the context is configured with AT THE TIME OF THE CALL; pbr_diag_bvh_build_info shows which."""
    sc = make_scene(pbr, "sponza", 5, 6000)
    a = sc.arrays()
    assert device.bvh_build_radius() == 0
    device.build_bvh(a["vertices"], a["facesV"], a["facesN"])    # not configured yet: the reference walk's radius
    assert device.bvh_build_radius() == 32
    cfg = sc.config(32, 32)
    cfg.traversal = 2
    device.upload_scene(sc.desc)
    device.configure(cfg)
    device.build_bvh(a["vertices"], a["facesV"], a["facesN"])
    assert device.bvh_build_radius() == 3
    device.set_knob("ploc_radius", 7)
    device.build_bvh(a["vertices"], a["facesV"], a["facesN"])
    assert device.bvh_build_radius() == 7
