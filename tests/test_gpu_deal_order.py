"""The ORDER in which the work queue deals its tiles (round 6: cost-ordered dealing, expensive tiles first) is placement
only: whatever the order — the spatial one, the library's own two cost orders (falling classes for short render calls, the
expensive quarter last for long ones), a reversed or a shuffled one handed in through
pbr_diag_set_tile_order — every (pixel, frame) unit is rendered exactly once, and image, debug image and counters are the
oracle's bit for bit.  Every plan, sharded and unsharded, single-frame and multi-frame launches."""
import numpy as np
import pytest

from conftest import same_values, describe_mismatch

pytestmark = pytest.mark.gpu

PLANS = {"refill-lean": 0, "refill-wide": 1, "phased-lean": 2, "phased-wide": 3, "phased-mid": 4, "refill-mid": 5, "phased-dual": 6}


def _scene(pbr, kind, triangles, depth=3):
    pbr.cfg_reset()
    pbr.cfg_set(**{"render.max_depth": depth})
    return pbr.HostScene.generate(kind, 2, triangles)


def _orders(order, first, rng):
    """Adversarial permutations of every band's stretch: reversed, shuffled, rotated by one."""
    rev, shuf, rot = order.copy(), order.copy(), order.copy()
    for b in range(8):
        seg = order[first[b]:first[b + 1]]
        rev[first[b]:first[b + 1]] = seg[::-1]
        shuf[first[b]:first[b + 1]] = rng.permutation(seg)
        rot[first[b]:first[b + 1]] = np.roll(seg, 1)
    return {"reversed": rev, "shuffled": shuf, "rotated": rot}


@pytest.mark.parametrize("plan", sorted(PLANS))
@pytest.mark.parametrize("world,rank", [(1, 0), (3, 1)])
def test_any_dealing_order_renders_the_oracles_bits(pbr, oracle, gpu_device, plan, world, rank):
    w, h, frames = 200, 120, 5                     # 25 x 15 tiles: ragged against 8 bands and against 3 ranks
    sc = _scene(pbr, "sponza", 9000)
    cfg, cam, px = sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h)
    seeds = pbr.frame_seeds(0, frames)
    ref = oracle.Renderer(sc.desc, cfg, threads=8)
    want = ref.render(0, seeds, px, cam)
    want_dbg = ref.debug
    mask = pbr.tiles.rows_of_rank(w, h, world, rank)
    cfg.tile_world, cfg.tile_rank = world, rank

    dev = pbr.Device(gpu_device)
    try:
        dev.pin_plan(PLANS[plan])
        dev.upload_scene(sc.desc)
        dev.configure(cfg)
        order, first = dev.tile_order()
        assert first[0] == 0 and first[8] == order.size == len(pbr.tiles.local_tile_ids(w, h, world, rank))
        assert sorted(order.tolist()) == list(range(order.size))          # the table names every local tile once
        rng = np.random.default_rng(7)
        results = {}
        for label, perm in [("library", None), ("library: cost classes", 1), ("library: expensive last", 2)] + list(_orders(order, first, rng).items()):
            if isinstance(perm, int):                                      # the library's own two cost orders, forced by the knob
                dev.set_tile_order(None)
                dev.set_knob("deal_order", perm)
                dev.pin_plan(PLANS[plan])                                  # (a knob resets the tuner, not the pin: say it again all the same)
            else:
                dev.set_knob("deal_order", -1)
                dev.set_tile_order(perm)
            dev.reset_accum()
            dev.render(0, seeds, px, cam)                                  # one multi-frame launch
            results[label] = (dev.read_output(), dev.read_debug(), dev.counters())
            dev.reset_accum()
            for k in range(frames):                                        # the viewer's frame-by-frame sequence
                dev.render_frame(float(seeds[k]), k / (k + 1.0), px, cam)
                dev.accumulate()
            dev.accumulate()                                               # the result is in imageOut again
            single = dev.read_output()
            assert same_values(single, results[label][0]), label + " frame by frame: " + describe_mismatch(single, results[label][0])
        for label, (img, dbg, cnt) in results.items():
            assert same_values(img[mask], want[mask]), label + ": " + describe_mismatch(img[mask], want[mask])
            assert same_values(dbg[mask][:, :2], want_dbg[mask][:, :2]), label + " debug: " + describe_mismatch(dbg[mask], want_dbg[mask])
            assert cnt == results["library"][2], label
        if world == 1:
            assert results["library"][2] == ref.counter_dict()
    finally:
        dev.close()


@pytest.mark.parametrize("kind,triangles,w,h", [("sponza", 20000, 640, 360), ("cornell", 0, 256, 256)])
def test_the_librarys_cost_order(pbr, oracle, gpu_device, kind, triangles, w, h):
    """After the first render the library has learnt the tiles' costs from the debug image: its cost order is a permutation
    of every band's tiles in eight classes of falling cost (spatial order inside a class), short launches are dealt in it,
    long ones in the spatial order, and the bits are the oracle's either way."""
    sc = _scene(pbr, kind, triangles, depth=4)
    cfg, cam, px = sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h)
    seeds = pbr.frame_seeds(0, 3)
    ref = oracle.Renderer(sc.desc, cfg, threads=8)
    want = ref.render(0, seeds, px, cam)
    dev = pbr.Device(gpu_device)
    try:
        dev.pin_plan(PLANS["phased-mid"])
        dev.upload_scene(sc.desc)
        dev.configure(cfg)
        with pytest.raises(pbr.PbrError, match="no cost order"):
            dev.tile_order(cost_ordered=True)
        dev.render(0, seeds[:1], px, cam)
        assert dev.last_deal() == ("spatial", True)
        dbg = dev.read_debug()
        cost = (dbg[..., 1].astype(np.float64) * 1265.0).reshape(h // 8, 8, w // 8, 8).sum((1, 3)).reshape(-1)
        spatial, first = dev.tile_order()
        order, first2 = dev.tile_order(cost_ordered=True)
        assert np.array_equal(first, first2)
        for b in range(8):
            seg, base = order[first[b]:first[b + 1]], spatial[first[b]:first[b + 1]]
            assert sorted(seg.tolist()) == sorted(base.tolist())
            c = cost[seg]
            # eight runs of falling cost: the run's smallest cost is never below the next run's largest, and inside a run the
            # tiles keep their spatial sequence
            place = {int(t): k for k, t in enumerate(base)}
            runs, start = [], 0
            for k in range(1, len(seg) + 1):
                if k == len(seg) or place[int(seg[k])] < place[int(seg[k - 1])]:
                    runs.append((start, k)); start = k
            assert len(runs) <= 8, (b, len(runs))
            for (a0, a1), (b0, b1) in zip(runs, runs[1:]):
                assert c[a0:a1].min() >= c[b0:b1].max() - 1e-3 * max(1.0, c.max())
        dev.render(1, seeds[1:], px, cam)                       # 2 frames x few tiles: a short launch
        assert dev.last_deal() == ("cost-classes", True)
        got = dev.read_output()
        assert same_values(got, want), describe_mismatch(got, want)
        assert same_values(dev.read_debug(), ref.debug)
        assert dev.counters() == ref.counter_dict()
        dev.set_knob("deal_order", 0)                           # the knob: always spatial
        dev.reset_accum()
        dev.render(0, seeds, px, cam)
        assert dev.last_deal()[0] == "spatial"
        assert same_values(dev.read_output(), want)
        # the order of LONG render calls: every band's most expensive quarter last, spatial inside both parts
        last, first3 = dev.tile_order(which=2)
        assert np.array_equal(first, first3)
        for b in range(8):
            seg, base = last[first[b]:first[b + 1]], spatial[first[b]:first[b + 1]]
            assert sorted(seg.tolist()) == sorted(base.tolist())
            place = {int(t): k for k, t in enumerate(base)}
            breaks = [k for k in range(1, len(seg)) if place[int(seg[k])] < place[int(seg[k - 1])]]
            assert len(breaks) <= 1, (b, breaks)                                   # two spatial runs
            if breaks:
                k = breaks[0]
                assert cost[seg[:k]].max() <= cost[seg[k:]].min() + 1e-3 * max(1.0, cost.max()) and len(seg) - k <= 0.3 * len(seg) + 1
        dev.set_knob("deal_order", 2)                           # ... forced onto this short render: the oracle's bits all the same
        dev.reset_accum()
        dev.render(0, seeds, px, cam)
        assert dev.last_deal()[0] == "expensive-last"
        assert same_values(dev.read_output(), want) and same_values(dev.read_debug(), ref.debug)
        dev.set_knob("deal_order", -1)
        dev.set_knob("chunk_frames", -1)
        tiles = (w // 8) * (h // 8)
        many = pbr.frame_seeds(0, 1 + (128 * 1024) // tiles + 4)          # a render call above the size limit of the cost classes ...
        dev.reset_accum()
        dev.render(0, many[:1], px, cam)
        dev.render(1, many[1:], px, cam)
        assert dev.last_deal()[0] == "spatial"
        more = pbr.frame_seeds(len(many), (192 * 1024) // tiles + 4)        # ... and one above the spatial order's
        dev.render(len(many), more, px, cam)
        assert dev.last_deal()[0] == "expensive-last"
    finally:
        dev.close()


def test_a_shard_deals_falling_classes_for_longer(pbr, gpu_device):
    """A rank's share (tile_world > 1) is dealt in falling cost classes up to 1 Mi tiles x frames, the whole frame up to 128 Ki:
    the same render call, sharded and not."""
    sc = _scene(pbr, "cornell", 0, depth=3)
    w = h = 256
    cam, px = sc.camera(), pbr.pixel_dimension(w, h)
    for world, expected in ((1, "spatial"), (2, "cost-classes")):
        cfg = sc.config(w, h)
        cfg.tile_world, cfg.tile_rank = world, 0
        dev = pbr.Device(gpu_device)
        try:
            dev.pin_plan(PLANS["phased-mid"])
            dev.upload_scene(sc.desc)
            dev.configure(cfg)
            tiles = len(dev.tile_order()[0])
            dev.render(0, pbr.frame_seeds(0, 1), px, cam)
            frames = (128 * 1024) // tiles + 40                  # above 128 Ki, below 192 Ki tiles x frames
            assert 128 * 1024 < tiles * frames <= 192 * 1024
            dev.render(1, pbr.frame_seeds(1, frames), px, cam)
            assert dev.last_deal() == (expected, True), (world, dev.last_deal())
            if world > 1:
                dev.render(1 + frames, pbr.frame_seeds(1 + frames, (1024 * 1024) // tiles + 8), px, cam)
                assert dev.last_deal()[0] == "expensive-last"
        finally:
            dev.close()


def test_a_table_that_is_not_a_permutation_is_refused(pbr, gpu_device):
    sc = _scene(pbr, "cornell", 0)
    dev = pbr.Device(gpu_device)
    try:
        dev.upload_scene(sc.desc)
        dev.configure(sc.config(64, 64))
        order, first = dev.tile_order()
        twice = order.copy(); twice[1] = twice[0]
        with pytest.raises(pbr.PbrError, match="named twice|not a tile of band"):
            dev.set_tile_order(twice)
        swapped = order.copy(); swapped[first[0]], swapped[first[7]] = order[first[7]], order[first[0]]      # a tile in another band's stretch
        with pytest.raises(pbr.PbrError, match="not a tile of band"):
            dev.set_tile_order(swapped)
        with pytest.raises(pbr.PbrError, match="entries"):
            dev.set_tile_order(order[:-1])
        got, _ = dev.tile_order()
        assert np.array_equal(got, order)                                  # a refused table changes nothing
    finally:
        dev.close()
