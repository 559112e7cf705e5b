"""Device time of a render as a function of the number of frames per launch: T(n) = a + b n?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
W, H = 1920, 1080
for kind, seed, tris, depth in (("cornell", 1, 0, 8), ("sponza", 2, 260000, 3)):
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris)
    dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(sc.config(W, H))
    cam, px = sc.camera(), pbr.pixel_dimension(W, H)
    dev.render(0, pbr.frame_seeds(0, 72), px, cam)
    out = []
    for n in (1, 2, 4, 8, 16, 32, 64):
        best = 1e9
        for rep in range(3):
            dev.render(72, pbr.frame_seeds(72, n), px, cam)
            best = min(best, dev.last_trace()[0])
        out.append("%d: %.3f ms (%.3f/frame)" % (n, best, best / n))
    print(kind, dev.last_plan()[0], " | ".join(out))
    dev.close()
