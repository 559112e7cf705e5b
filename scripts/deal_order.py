"""What the ORDER in which the queue deals its tiles is worth (round 6): the same launches with the spatial order and with
cost-ordered dealing (expensive tiles first), the order built here from the debug image of one frame and handed to the library
through pbr_diag_set_tile_order.  Per scene and plan: single-frame launch, 20-frame launch, rank 0's share of a 20-frame render
split 8 ways, 64-frame launch — trace ms (HIP events around the path-tracing launch), best of `reps`.

usage: python scripts/deal_order.py [scene ...]      PBR_PLANS="6 4" chooses plans; DEAL_WORLDS="1 8"; DEAL_FRAMES="1 20 64"
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
from pbr_amd import tiles as T

W, H = int(os.environ.get("DEAL_W", 1920)), int(os.environ.get("DEAL_H", 1080))
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}


def get_order(dev):
    return dev.tile_order()


def set_order(dev, order):
    dev.set_tile_order(order)


def local_tile_cost(debug, world, rank):
    """Node visits per local tile (sum over its 64 pixels) from a full debug image (other ranks' pixels are 0)."""
    nodes = debug[..., 1].astype(np.float64) * 1265.0
    per_tile = nodes.reshape(H // 8, 8, W // 8, 8).sum((1, 3)).reshape(-1)
    return per_tile[T.local_tile_ids(W, H, world, rank)]


def cost_order(spatial, first, cost, classes):
    """Per band: the spatial order stably partitioned into `classes` cost classes, the most expensive class first
    (classes = 0: fully sorted, descending; classes < 0: ascending = the adversarial order)."""
    out = spatial.copy()
    for b in range(8):
        seg = spatial[first[b]:first[b + 1]]
        c = cost[seg]
        if classes == 0:
            out[first[b]:first[b + 1]] = seg[np.argsort(-c, kind="stable")]
        elif classes < 0:
            out[first[b]:first[b + 1]] = seg[np.argsort(c, kind="stable")]
        elif classes == 3:
            # top 10 % first, cheapest 20 % last
            hi, lo = np.quantile(c, 0.9), np.quantile(c, 0.2)
            cls = np.where(c >= hi, 0, np.where(c <= lo, 2, 1))
            out[first[b]:first[b + 1]] = seg[np.argsort(cls, kind="stable")]
        elif classes >= 300:
            # 300 + p: the most expensive p % of the band's tiles LAST, the rest first, spatial inside both
            cut = np.quantile(c, 1.0 - (classes - 300) / 100.0)
            out[first[b]:first[b + 1]] = seg[np.argsort((c > cut).astype(int), kind="stable")]
        elif classes >= 200:
            # 200 + p: the cheapest p % of the band's tiles FIRST, the rest after them, spatial inside both
            cut = np.quantile(c, (classes - 200) / 100.0)
            out[first[b]:first[b + 1]] = seg[np.argsort((c > cut).astype(int), kind="stable")]
        elif classes >= 100:
            # 100 + n: n classes ASCENDING (cheapest class first), spatial inside a class
            n = classes - 100
            edges = np.quantile(c, np.linspace(0, 1, n + 1)[1:-1])
            cls = np.searchsorted(edges, c, side="right")
            out[first[b]:first[b + 1]] = seg[np.argsort(cls, kind="stable")]
        else:
            edges = np.quantile(c, np.linspace(0, 1, classes + 1)[1:-1])
            cls = classes - 1 - np.searchsorted(edges, c, side="right")
            out[first[b]:first[b + 1]] = seg[np.argsort(cls, kind="stable")]
    return out


def main():
    plans = [int(p) for p in os.environ.get("PBR_PLANS", "6 4").split()]
    worlds = [int(p) for p in os.environ.get("DEAL_WORLDS", "1 8").split()]
    lengths = [int(p) for p in os.environ.get("DEAL_FRAMES", "1 20 64").split()]
    variants = [("spatial", None), ("3 classes", 3), ("8 classes", 8), ("32 classes", 32), ("sorted", 0), ("ascending", -1), ("asc 2", 102), ("asc 3", 103), ("asc 4", 104), ("asc 8", 108), ("cheap10 first", 210), ("cheap25 first", 225), ("cheap75 first", 275),
                ("exp10 last", 310), ("exp25 last", 325), ("spatial again", None)]
    if os.environ.get("DEAL_VARIANTS"):
        variants = [v for v in variants if v[0] in os.environ["DEAL_VARIANTS"].split(",")]
    reps = int(os.environ.get("DEAL_REPS", 7))
    trav, arith = int(os.environ.get("AB_TRAVERSAL", "0")), int(os.environ.get("AB_ARITH", "0"))
    for name in (sys.argv[1:] or ["sponza"]):
        kind, seed, tris, depth = SCENES[name]
        pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
        sc = pbr.HostScene.generate(kind, seed, tris)
        cam, px = sc.camera(), pbr.pixel_dimension(W, H)
        for world in worlds:
            for plan in plans:
                cfg = sc.config(W, H); cfg.tile_world, cfg.tile_rank = world, 0
                cfg.traversal, cfg.arith = trav, arith
                dev = pbr.Device(0); dev.pin_plan(plan); dev.upload_scene(sc.desc); dev.configure(cfg)
                dev.render(0, pbr.frame_seeds(0, 1), px, cam)
                cost = local_tile_cost(dev.read_debug(), world, 0)
                spatial, first = get_order(dev)
                dev.render(1, pbr.frame_seeds(1, 40), px, cam)        # clocks up
                want = dev.read_output()
                rows = {}
                for label, classes in variants:
                    set_order(dev, None if classes is None else cost_order(spatial, first, cost, classes))
                    cells = []
                    for n in lengths:
                        best = 1e9
                        for rep in range(reps):
                            dev.render(41, pbr.frame_seeds(41, n), px, cam)
                            best = min(best, dev.last_trace()[0])
                        cells.append(best)
                    rows[label] = cells
                    print("%-8s N=%d %-12s t%d a%d %-10s " % (name, world, dev.last_plan()[0], trav, arith, label) +
                          "  ".join("%3d fr: %7.3f ms" % (n, ms) for n, ms in zip(lengths, cells)), flush=True)
                dev.close()


if __name__ == "__main__":
    main()
