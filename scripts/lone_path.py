"""How fast is a path when it has the machine to itself?  One wave's worth of pixels (an 8 x 8 image) or a few, one frame per
launch: the launch lasts as long as its longest path.  Fit of launch time against the longest path's node visits over many
seeds -> microseconds per visit of a lone path, the floor of what the end of a launch can cost ("How a launch ends").

usage: python scripts/lone_path.py [scene] [width] [height]      PBR_PLANS="6 4 2"
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")
import numpy as np
import pbr_loader
pbr = pbr_loader.load()

SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H = int(sys.argv[3]) if len(sys.argv) > 3 else 8
kind, seed, tris, depth = SCENES[name]
pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
sc = pbr.HostScene.generate(kind, seed, tris)
cam, px = sc.camera(), pbr.pixel_dimension(W, H)
for plan in [int(p) for p in os.environ.get("PBR_PLANS", "6 4 2").split()]:
    cfg = sc.config(W, H)
    cfg.traversal, cfg.arith = int(os.environ.get("AB_TRAVERSAL", "0")), int(os.environ.get("AB_ARITH", "0"))
    dev = pbr.Device(0); dev.pin_plan(plan); dev.upload_scene(sc.desc); dev.configure(cfg)
    seeds = pbr.frame_seeds(0, 400)
    for k in range(40):
        dev.render_frame(float(seeds[k]), 0.0, px, cam)
    rows = []
    for k in range(40, 400):
        best = 1e9
        for rep in range(3):
            dev.render_frame(float(seeds[k]), 0.0, px, cam)
            best = min(best, dev.last_trace()[0])
        d = dev.read_debug()
        nodes, faces = d[..., 1] * 1265.0, d[..., 0] * 1082.0
        i = np.unravel_index(np.argmax(nodes), nodes.shape)
        rows.append((best * 1e3, nodes.max(), faces[i], nodes.sum(), faces.sum()))
    r = np.array(rows)
    A = np.stack([np.ones(len(r)), r[:, 1]], 1)
    (a, b), *_ = np.linalg.lstsq(A, r[:, 0], rcond=None)
    top = r[np.argsort(-r[:, 1])[:5]]
    print("%-8s %dx%d %-12s launch = %.1f us + %.3f us x (node visits of the longest path);  longest paths: " % (name, W, H, dev.last_plan()[0], a, b) +
          ", ".join("%d visits %.0f us" % (v, t) for t, v, *_ in top) + ";  mean launch %.0f us for mean longest %.0f visits, mean sum %.0f visits" % (r[:, 0].mean(), r[:, 1].mean(), r[:, 3].mean()), flush=True)
    dev.close()
