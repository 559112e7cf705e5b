"""How long is the longest path?  One 1080p frame (1 sample per pixel): the debug image holds every path's node visits and
face tests (pathtracing.cl:73-78).  The end of a launch lasts as long as the longest paths in flight when the queue runs
dry (DESIGN.md "How a launch ends").  usage: python scripts/path_lengths.py [scene ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
W, H = 1920, 1080
for name in (sys.argv[1:] or ["sponza", "dragon", "hairball", "cornell"]):
    kind, seed, tris, depth = SCENES[name]
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris)
    dev = pbr.Device(0); dev.pin_plan(4); dev.upload_scene(sc.desc); dev.configure(sc.config(W, H))
    cam, px = sc.camera(), pbr.pixel_dimension(W, H)
    dev.render(0, pbr.frame_seeds(0, 8), px, cam)
    dev.reset_accum()
    dev.render(0, pbr.frame_seeds(0, 1), px, cam)
    ms = dev.last_trace()[0]
    dbg = dev.read_debug().astype(np.float64)
    nodes = np.rint(dbg[..., 1] * 1265.0).ravel()
    tris_ = np.rint(dbg[..., 0] * 1082.0).ravel()
    q = np.quantile(nodes, [0.5, 0.9, 0.99, 0.999, 0.9999])
    order = np.sort(nodes)[::-1]
    # work (node visits) held by the paths above a quantile: what a lane that starts one of them late still has to do
    print("%-8s single frame %.3f ms | node visits per path: mean %.0f  median %.0f  p90 %.0f  p99 %.0f  p99.9 %.0f  p99.99 %.0f  max %.0f  (%.1f x the mean) | "
          "face tests mean %.1f max %.0f | the 393 216th-longest path (one per resident lane) has %.0f visits" % (
              name, ms, nodes.mean(), q[0], q[1], q[2], q[3], q[4], nodes.max(), nodes.max() / nodes.mean(), tris_.mean(), tris_.max(), order[min(393215, order.size - 1)]), flush=True)
    dev.close()
