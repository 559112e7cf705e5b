"""Per-frame cost of the interactive call sequence (pbr_render_frame + pbr_accumulate, no read-back): wall vs device time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
W, H = 1920, 1080
for kind, seed, tris, depth in (("cornell", 1, 0, 8), ("sponza", 2, 260000, 3)):
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris)
    dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(sc.config(W, H))
    cam, px = sc.camera(), pbr.pixel_dimension(W, H)
    seeds = pbr.frame_seeds(0, 300)
    for k in range(100):                      # tuner + warm-up
        dev.render_frame(float(seeds[k]), k / (k + 1.0), px, cam); dev.accumulate()
    dev_ms = 0.0; t0 = time.perf_counter()
    for k in range(100, 300):
        dev.render_frame(float(seeds[k]), k / (k + 1.0), px, cam); dev_ms += dev.last_kernel_ms(); dev.accumulate()
    wall = (time.perf_counter() - t0) * 1e3 / 200
    print("%-8s single-frame calls: %.3f ms wall per frame, %.3f ms on the device (%s) -> %.0f Msamples/s" % (kind, wall, dev_ms / 200, dev.last_plan()[0], W * H / wall / 1e3))
    dev.close()
