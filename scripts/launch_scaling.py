"""Device time of the path-tracing launch as a function of the number of frames per launch, per plan:
is T(n) = a + b n?  usage: python scripts/launch_scaling.py [scene ...]   (PBR_PLANS="0 5" to choose plans)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
W, H = 1920, 1080
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
for name in (sys.argv[1:] or ["cornell", "sponza"]):
    kind, seed, tris, depth = SCENES[name]
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris)
    for plan in os.environ.get("PBR_PLANS", "0 5 2 4").split():
        os.environ["PBR_PLAN"] = plan
        dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(sc.config(W, H))
        cam, px = sc.camera(), pbr.pixel_dimension(W, H)
        dev.render(0, pbr.frame_seeds(0, 112), px, cam)
        out = []
        for n in (1, 2, 4, 8, 16, 32, 64, 128, 256):
            times = []
            for rep in range(3):
                dev.render(72, pbr.frame_seeds(72, n), px, cam)
                times.append(dev.last_trace()[0])
            out.append("%d: %.3f..%.3f" % (n, min(times) / n, max(times) / n))
        print("%-8s %-12s ms/frame  " % (name, dev.last_plan()[0]) + " | ".join(out), flush=True)
        dev.close()
