"""Same number of (pixel, frame) units, different shapes: 1/N of the tiles x N x 32 frames.  Is a unit of a shard as cheap
as a unit of the whole frame?  usage: python scripts/shard_equal_work.py [plan]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import pbr_loader
pbr = pbr_loader.load()
W, H = 1920, 1080
os.environ["PBR_PLAN"] = sys.argv[1] if len(sys.argv) > 1 else "5"
pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": 8})
sc = pbr.HostScene.generate("cornell", 1, 0)
cam, px = sc.camera(), pbr.pixel_dimension(W, H)
for world, rank in ((1, 0), (2, 0), (4, 0), (8, 0), (8, 3)):
    cfg = sc.config(W, H); cfg.tile_world, cfg.tile_rank = world, rank
    dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(cfg)
    n = 32 * world
    dev.render(0, pbr.frame_seeds(0, n), px, cam)
    best = 1e9
    for rep in range(3):
        dev.render(n, pbr.frame_seeds(n, n), px, cam)
        best = min(best, dev.last_trace()[0])
    c0 = dev.counters(); dev.render(n, pbr.frame_seeds(n, n), px, cam); c1 = dev.counters()
    paths = c1["paths"] - c0["paths"]
    print("N=%d rank %d  %3d frames  %-11s trace %.3f ms   per sample: %.2f nodes %.2f tris %.3f hits" % (
        world, rank, n, dev.last_plan()[0], best, (c1["nodes"] - c0["nodes"]) / paths, (c1["tris"] - c0["tris"]) / paths, (c1["hits"] - c0["hits"]) / paths), flush=True)
    dev.close()
