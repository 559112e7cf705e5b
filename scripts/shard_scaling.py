"""One rank's share of an N-way tile-sharded render on one GPU: device time of rank 0's K frames for N = 1, 2, 4, 8 and
the scaling efficiency the kernel side alone would give, T(1) / (N x T(N)) (no gather, no barrier).
usage: python scripts/shard_scaling.py [scene] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import pbr_loader
pbr = pbr_loader.load()
W, H = 1920, 1080
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
kind, seed, tris, depth = SCENES[name]
pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
sc = pbr.HostScene.generate(kind, seed, tris)
cam, px = sc.camera(), pbr.pixel_dimension(W, H)
base = None
for world in (1, 2, 4, 8):
    cfg = sc.config(W, H); cfg.tile_world, cfg.tile_rank = world, 0
    cfg.traversal, cfg.arith = int(os.environ.get("AB_TRAVERSAL", "0")), int(os.environ.get("AB_ARITH", "0"))     # pbr_config's opt-in modes
    dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(cfg)
    warm = dev.tune_budget()
    dev.render(0, pbr.frame_seeds(0, warm), px, cam)
    best = 1e9
    for rep in range(4):
        dev.render(warm, pbr.frame_seeds(warm, frames), px, cam)
        best = min(best, dev.last_kernel_ms())
    base = base or best
    print("%-8s traversal %d arith %d N=%d  %-12s %8.3f ms for %d frames of 1/%d of the tiles   efficiency %.3f" % (name, cfg.traversal, cfg.arith, world, dev.last_plan()[0], best, frames, world, base / (world * best)), flush=True)
    dev.close()
