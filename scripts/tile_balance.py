"""How evenly do tile -> rank mappings split the work?  Per-tile cost from the debug image (node visits + face tests
per pixel, summed over a few frames), then max / mean over the ranks for candidate mappings.
usage: python scripts/tile_balance.py [scene]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
W, H = 1920, 1080
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
kind, seed, tris, depth = SCENES[name]
pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
sc = pbr.HostScene.generate(kind, seed, tris)
cam, px = sc.camera(), pbr.pixel_dimension(W, H)
dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(sc.config(W, H))
cost = np.zeros((H, W))
for k in range(12):
    dev.render(k, pbr.frame_seeds(k, 1), px, cam)
    dbg = dev.read_debug()
    cost += dbg[..., 1] * 1265.0 * 39 + dbg[..., 0] * 1082.0 * 75 + 400.0     # ~ instructions: node visits, face tests, a floor for shading
tiles = cost.reshape(H // 8, 8, W // 8, 8).sum(axis=(1, 3))      # [ty, tx]
ty, tx = np.indices(tiles.shape)
tilesX = tiles.shape[1]
t = ty * tilesX + tx
for N in (2, 4, 8):
    out = []
    for label, rank in (("t % N", t % N), ("(x + y) % N", (tx + ty) % N), ("(x + 3 y) % N", (tx + 3 * ty) % N),
                        ("(x + 5 y) % N", (tx + 5 * ty) % N), ("hash", ((t * 2654435761) >> 7) % N)):
        w = np.array([tiles[rank == r].sum() for r in range(N)])
        out.append("%s: %.4f" % (label, w.max() / w.mean()))
    print(name, "N=%d  max / mean work   " % N + "   ".join(out), flush=True)
