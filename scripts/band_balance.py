"""Cost-BALANCED bands (round 6): the queue's eight bands are equal ROW ranges, one per XCD, and an XCD works on its own band until
it is empty — so an XCD whose band is cheap idles its memory link while the XCDs of the expensive bands saturate theirs, and only
the end of the launch is shared.  Variants, all placement only (pbr_diag_set_tile_order with band_first):
  equal rows (the library's spatial order) · rows split so that every band has the same COST (from the debug image), spatial inside ·
  the same + each band's most expensive quarter last · equal rows + most expensive quarter last (what deal_order.py found).
usage: python scripts/band_balance.py [scene ...]    PBR_PLANS, DEAL_WORLDS, DEAL_FRAMES, DEAL_REPS as scripts/deal_order.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PBR_LAB_ENV", "1")
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
from pbr_amd import tiles as T

W, H = int(os.environ.get("DEAL_W", 1920)), int(os.environ.get("DEAL_H", 1080))
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}


def band_rows_equal(rows):
    return [(b * rows) // 8 for b in range(9)]


def band_rows_balanced(row_cost):
    """Row boundaries such that every band's cost is as close to 1/8 of the total as whole rows allow."""
    cum = np.concatenate([[0.0], np.cumsum(row_cost)])
    bounds = [0]
    for b in range(1, 8):
        target = cum[-1] * b / 8
        r = int(np.searchsorted(cum, target))
        r = r if abs(cum[r] - target) <= abs(cum[r - 1] - target) else r - 1
        bounds.append(max(bounds[-1] + 1, min(r, len(row_cost) - (8 - b))))
    return bounds + [len(row_cost)]


def table(bounds, width, tiles, cost, expensive_last=0.0, interleave=0, first_too=0.0):
    order, first = [], [0]
    for b in range(8):
        seg = [r * width + c for c in range(width) for r in range(bounds[b], bounds[b + 1]) if r * width + c < tiles]
        seg = np.array(seg, np.int64)
        if interleave > 1 and len(seg):
            # `interleave` cost classes (quantiles), each in its spatial sequence, dealt proportionally: every stretch of the band
            # has the band's average cost
            edges = np.quantile(cost[seg], np.linspace(0, 1, interleave + 1)[1:-1])
            cls = np.searchsorted(edges, cost[seg], side="right")
            lists = [seg[cls == k] for k in range(interleave)]
            pos = np.concatenate([(np.arange(len(l)) + 0.5) / max(1, len(l)) for l in lists])
            seg = np.concatenate(lists)[np.argsort(pos, kind="stable")]
        elif first_too > 0 and len(seg):
            # the cheapest `first_too` first, the most expensive `expensive_last` last, the middle between them
            lo, hi = np.quantile(cost[seg], first_too), np.quantile(cost[seg], 1.0 - expensive_last)
            seg = seg[np.argsort(np.where(cost[seg] <= lo, 0, np.where(cost[seg] > hi, 2, 1)), kind="stable")]
        elif expensive_last > 0 and len(seg):
            cut = np.quantile(cost[seg], 1.0 - expensive_last)
            seg = seg[np.argsort((cost[seg] > cut).astype(int), kind="stable")]
        order.extend(seg.tolist()); first.append(len(order))
    return np.array(order, np.uint32), np.array(first, np.uint32)


def table_paired(bounds, width, tiles, cost, expensive_last=0.0):
    """The eight row ranges assigned to band INDICES so that in the queue's cyclic stealing order (home + 1, home + 2, ...) every
    light band is followed by a heavy one: ranges sorted by cost c0 <= ... <= c7 take the indices of c0 c7 c1 c6 c2 c5 c3 c4."""
    segs = []
    for b in range(8):
        seg = np.array([r * width + c for c in range(width) for r in range(bounds[b], bounds[b + 1]) if r * width + c < tiles], np.int64)
        if expensive_last > 0 and len(seg):
            cut = np.quantile(cost[seg], 1.0 - expensive_last)
            seg = seg[np.argsort((cost[seg] > cut).astype(int), kind="stable")]
        segs.append(seg)
    by_cost = sorted(range(8), key=lambda b: cost[segs[b]].sum())
    arrangement = [by_cost[0], by_cost[7], by_cost[1], by_cost[6], by_cost[2], by_cost[5], by_cost[3], by_cost[4]]
    order, first = [], [0]
    for b in arrangement:
        order.extend(segs[b].tolist()); first.append(len(order))
    return np.array(order, np.uint32), np.array(first, np.uint32)


def table_stripes(rows, width, tiles, cost, g, expensive_last=0.0):
    """Stripes of g tile rows dealt round-robin to the eight bands (band b holds stripes b, b + 8, ...), column by column inside a
    stripe: every XCD gets a sample of the whole image (balanced at all times) at the price of its L2 seeing all of the scene."""
    segs = [[] for _ in range(8)]
    for k, r0 in enumerate(range(0, rows, g)):
        seg = [r * width + c for c in range(width) for r in range(r0, min(rows, r0 + g)) if r * width + c < tiles]
        segs[k % 8].extend(seg)
    order, first = [], [0]
    for b in range(8):
        seg = np.array(segs[b], np.int64)
        if expensive_last > 0 and len(seg):
            cut = np.quantile(cost[seg], 1.0 - expensive_last)
            seg = seg[np.argsort((cost[seg] > cut).astype(int), kind="stable")]
        order.extend(seg.tolist()); first.append(len(order))
    return np.array(order, np.uint32), np.array(first, np.uint32)


def table_coda(bounds, width, tiles, cost, expensive=0.25, coda=0.10):
    """Per band: [ the middle ][ the most expensive `expensive` ][ the cheapest `coda` ] — the heavy tiles late (shared by the XCDs that
    have run out of their own), and the very last paths to start short ones."""
    order, first = [], [0]
    for b in range(8):
        seg = np.array([r * width + c for c in range(width) for r in range(bounds[b], bounds[b + 1]) if r * width + c < tiles], np.int64)
        if len(seg):
            lo, hi = np.quantile(cost[seg], coda), np.quantile(cost[seg], 1.0 - expensive)
            key = np.where(cost[seg] > hi, 1, np.where(cost[seg] <= lo, 2, 0))
            seg = seg[np.argsort(key, kind="stable")]
        order.extend(seg.tolist()); first.append(len(order))
    return np.array(order, np.uint32), np.array(first, np.uint32)


def table_global(bounds, width, tiles, cost, share, spread):
    """The most expensive `share` of ALL tiles (one global cut) last: in their own bands (spread = False), or dealt round-robin
    to the tails of the eight bands whatever band they lie in (spread = True: every XCD gets an eighth of the heavy work)."""
    cut = np.quantile(cost, 1.0 - share)
    light, heavy = [], []
    for b in range(8):
        seg = np.array([r * width + c for c in range(width) for r in range(bounds[b], bounds[b + 1]) if r * width + c < tiles], np.int64)
        light.append(seg[cost[seg] <= cut]); heavy.append(seg[cost[seg] > cut])
    if spread:
        every = np.concatenate(heavy)
        heavy = [every[b::8] for b in range(8)]
    order, first = [], [0]
    for b in range(8):
        order.extend(light[b].tolist()); order.extend(heavy[b].tolist()); first.append(len(order))
    return np.array(order, np.uint32), np.array(first, np.uint32)


def main():
    plans = [int(p) for p in os.environ.get("PBR_PLANS", "6 4").split()]
    worlds = [int(p) for p in os.environ.get("DEAL_WORLDS", "1").split()]
    lengths = [int(p) for p in os.environ.get("DEAL_FRAMES", "20 64").split()]
    reps = int(os.environ.get("DEAL_REPS", 9))
    trav = int(os.environ.get("AB_TRAVERSAL", "0"))
    for name in (sys.argv[1:] or ["sponza"]):
        kind, seed, tris, depth = SCENES[name]
        pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
        sc = pbr.HostScene.generate(kind, seed, tris)
        cam, px = sc.camera(), pbr.pixel_dimension(W, H)
        for world in worlds:
            for plan in plans:
                cfg = sc.config(W, H); cfg.tile_world, cfg.tile_rank, cfg.traversal = world, 0, trav
                dev = pbr.Device(0); dev.pin_plan(plan); dev.upload_scene(sc.desc); dev.configure(cfg)
                dev.set_knob("deal_order", 0)
                dev.render(0, pbr.frame_seeds(0, 1), px, cam)
                nodes = dev.read_debug()[..., 1].astype(np.float64) * 1265.0
                cost = nodes.reshape(H // 8, 8, W // 8, 8).sum((1, 3)).reshape(-1)[T.local_tile_ids(W, H, world, 0)]
                spatial, first = dev.tile_order()
                tiles = len(spatial)
                width = max(1, (W // 8 + world - 1) // world)
                rows = (tiles + width - 1) // width
                padded = np.zeros(rows * width); padded[:tiles] = cost
                row_cost = padded.reshape(rows, width).sum(1)
                equal, balanced = band_rows_equal(rows), band_rows_balanced(row_cost)
                share = [row_cost[equal[b]:equal[b + 1]].sum() / row_cost.sum() for b in range(8)]
                dev.render(1, pbr.frame_seeds(1, 40), px, cam)
                desc8 = None
                if os.environ.get("BAND_SHORT"):
                    # the short-launch order of the library (eight classes of falling cost) next to the candidates, by launch length
                    import importlib
                    deal = importlib.import_module("deal_order")
                    desc8 = (deal.cost_order(spatial, first, cost, 8), first)
                variants = [("equal rows (library)", None), ("equal rows via table", table(equal, width, tiles, cost)), ("balanced rows", table(balanced, width, tiles, cost)),
                            ("balanced + exp25 last", table(balanced, width, tiles, cost, 0.25)), ("equal + exp25 last", table(equal, width, tiles, cost, 0.25)),
                            ("equal + exp40 last", table(equal, width, tiles, cost, 0.40)), ("equal + exp15 last", table(equal, width, tiles, cost, 0.15)),
                            ("balanced + interleave 4", table(balanced, width, tiles, cost, interleave=4)), ("equal + interleave 4", table(equal, width, tiles, cost, interleave=4)),
                            ("equal rows (library) again", None)]
                if os.environ.get("BAND_GLOBAL"):
                    variants = [("equal rows (library)", None), ("equal + exp25 last", table(equal, width, tiles, cost, 0.25)),
                                ("global heavy 25 % last, own band", table_global(equal, width, tiles, cost, 0.25, False)),
                                ("global heavy 25 % last, spread", table_global(equal, width, tiles, cost, 0.25, True)),
                                ("global heavy 40 % last, own band", table_global(equal, width, tiles, cost, 0.40, False)),
                                ("equal rows (library) again", None)]
                if os.environ.get("BAND_STRIPES"):
                    variants = [("equal rows (library)", None), ("equal + exp25 last", table(equal, width, tiles, cost, 0.25)),
                                ("stripes of 2 rows", table_stripes(rows, width, tiles, cost, 2)), ("stripes of 4 rows", table_stripes(rows, width, tiles, cost, 4)),
                                ("stripes of 8 rows", table_stripes(rows, width, tiles, cost, 8)), ("stripes of 4 rows + exp25 last", table_stripes(rows, width, tiles, cost, 4, 0.25)),
                                ("stripes of 8 rows + exp25 last", table_stripes(rows, width, tiles, cost, 8, 0.25)),
                                ("equal rows (library) again", None)]
                if os.environ.get("BAND_PAIRED"):
                    variants = [("equal rows (library)", None), ("equal + exp25 last", table(equal, width, tiles, cost, 0.25)),
                                ("light next to heavy, spatial", table_paired(equal, width, tiles, cost)),
                                ("light next to heavy + exp25 last", table_paired(equal, width, tiles, cost, 0.25)),
                                ("equal rows (library) again", None)]
                if desc8 is not None:
                    variants = [("equal rows (library)", None), ("8 classes falling", desc8), ("equal + exp25 last", table(equal, width, tiles, cost, 0.25)), ("equal rows (library) again", None)]
                    if os.environ.get("BAND_CODA"):
                        variants[3:3] = [("exp25 late + cheapest 10 % coda", table_coda(equal, width, tiles, cost, 0.25, 0.10)),
                                         ("exp25 late + cheapest 25 % coda", table_coda(equal, width, tiles, cost, 0.25, 0.25)),
                                         ("exp40 late + cheapest 20 % coda", table_coda(equal, width, tiles, cost, 0.40, 0.20))]
                print("%-8s N=%d: cost share of the eight equal-row bands %s; balanced row bounds %s" % (name, world, " ".join("%.3f" % s for s in share), balanced), flush=True)
                for label, tb in variants:
                    if tb is None:
                        dev.set_tile_order(None)
                    else:
                        dev.set_tile_order(tb[0], tb[1])
                    cells = []
                    for n in lengths:
                        best = 1e9
                        for rep in range(reps):
                            dev.render(41, pbr.frame_seeds(41, n), px, cam)
                            best = min(best, dev.last_trace()[0])
                        cells.append(best)
                    print("%-8s N=%d %-12s t%d %-28s " % (name, world, dev.last_plan()[0], trav, label) + "  ".join("%3d fr: %8.3f ms" % (n, ms) for n, ms in zip(lengths, cells)), flush=True)
                dev.close()


if __name__ == "__main__":
    main()
