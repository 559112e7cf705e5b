"""Lab: sub-heads per XCD WITHOUT giving up the XCD's compact working set.  lab/libpbrhip_heads32g.so (scripts/build_heads_lab.py 32
grouped) has four queue heads per XCD and spreads a band's thieves over them; its default tables give every head its own range of tile
rows (an XCD then works in four places).  Here the tables are written from outside (pbr_diag_set_tile_order): the product's eight
row bands, each band's order dealt round-robin to its four heads in runs of `run` tiles — the XCD's waves stay in one neighbourhood.
usage: PBR_HIP_LIB=lab/libpbrhip_heads32g.so HEADS=32 python scripts/heads_tables.py [scene ...]   (HEADS=8 with the product library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PBR_LAB_ENV", "1")
import numpy as np
import pbr_loader
pbr = pbr_loader.load()

W, H = 1920, 1080
HEADS = int(os.environ.get("HEADS", 8))
SUB = HEADS // 8
SCENES = {"cornell": ("cornell", 1, 0, 8, 5, 256), "sponza": ("sponza", 2, 260000, 3, 6, 64), "dragon": ("dragon", 1, 870000, 3, 4, 64), "hairball": ("hairball", 3, 2000000, 3, 4, 32)}


def bands8(cost, expensive_last):
    width, rows = W // 8, H // 8
    out = []
    for b in range(8):
        seg = np.array([r * width + c for c in range(width) for r in range((b * rows) // 8, ((b + 1) * rows) // 8)], np.int64)
        if expensive_last > 0:
            cut = np.quantile(cost[seg], 1.0 - expensive_last)
            seg = seg[np.argsort((cost[seg] > cut).astype(int), kind="stable")]
        out.append(seg)
    return out


def split(bands, run):
    """Every band's sequence dealt to its SUB heads in runs of `run` tiles."""
    order, first = [], [0]
    for seg in bands:
        which = (np.arange(len(seg)) // run) % SUB
        for s in range(SUB):
            order.extend(seg[which == s].tolist()); first.append(len(order))
    return np.array(order, np.uint32), np.array(first, np.uint32)


def main():
    reps = int(os.environ.get("DEAL_REPS", 7))
    for name in (sys.argv[1:] or ["sponza"]):
        kind, seed, tris, depth, plan, frames = SCENES[name]
        pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
        sc = pbr.HostScene.generate(kind, seed, tris)
        cam, px = sc.camera(), pbr.pixel_dimension(W, H)
        cfg = sc.config(W, H)
        dev = pbr.Device(0); dev.pin_plan(plan); dev.upload_scene(sc.desc); dev.configure(cfg)
        dev.set_knob("deal_order", 0)
        dev.render(0, pbr.frame_seeds(0, 1), px, cam)
        nodes = dev.read_debug()[..., 1].astype(np.float64) * 1265.0
        cost = nodes.reshape(H // 8, 8, W // 8, 8).sum((1, 3)).reshape(-1)
        dev.render(1, pbr.frame_seeds(1, 40), px, cam)
        variants = [("library: spatial", None, 0), ("library: expensive-last", None, 2)]
        for label, share in (("spatial", 0.0), ("exp25 last", 0.25)):
            for run in ((1,) if SUB == 1 else (1, 4, 16)):
                variants.append(("8 row bands, %s%s" % (label, "" if SUB == 1 else ", runs of %d tiles to %d heads" % (run, SUB)), split(bands8(cost, share), run), 0))
        variants.append(("library: spatial again", None, 0))
        for label, tb, knob in variants:
            dev.set_tile_order(None) if tb is None else dev.set_tile_order(tb[0], tb[1])
            dev.set_knob("deal_order", knob)
            best = 1e9
            for rep in range(reps):
                dev.render(41, pbr.frame_seeds(41, frames), px, cam)
                best = min(best, dev.last_trace()[0])
            print("%-8s heads %2d plan %d %3d frames  %-58s %8.3f ms  %8.1f Msamples/s" % (name, HEADS, plan, frames, label, best, W * H * frames / best / 1e3), flush=True)
        dev.close()


main()
