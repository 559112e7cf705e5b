import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
W, H = 1920, 1080
for kind, seed, tris in (("dragon", 1, 870000), ("hairball", 3, 2000000), ("sponza", 2, 260000), ("cornell", 1, 0)):
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": 3 if kind != "cornell" else 8})
    sc = pbr.HostScene.generate(kind, seed, tris)
    dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(sc.config(W, H))
    dev.render(0, pbr.frame_seeds(0, 1), pbr.pixel_dimension(W, H), sc.camera())
    d = dev.read_debug()[..., 1].astype(np.float64) * 1265.0
    t = d.reshape(H // 8, 8, W // 8, 8).sum((1, 3))
    m = t.mean()
    print(kind, "tile cost mean %.0f  min %.0f max %.0f; share of tiles below 25%% of the mean: %.3f, below 50%%: %.3f; their share of the work: %.3f / %.3f" % (
        m, t.min(), t.max(), (t < 0.25 * m).mean(), (t < 0.5 * m).mean(), t[t < 0.25 * m].sum() / t.sum(), t[t < 0.5 * m].sum() / t.sum()))
    dev.close()
