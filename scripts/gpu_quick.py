"""Hang-safe parity probe: renders in a worker thread, gives up after a few seconds."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
from oracle import oracle

def attempt(label, fn, limit=20.0):
    box = []
    t = threading.Thread(target=lambda: box.append(fn()), daemon=True)
    t.start(); t.join(limit)
    if not box:
        print(label, "HANG", flush=True); os._exit(3)
    return box[0]

for brdf in (1, 0):
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": 4, "render.brdf": brdf})
    sc = pbr.HostScene.generate("cornell")
    W = H = 64
    cfg = sc.config(W, H); cam = sc.camera(); px = pbr.pixel_dimension(W, H, 45.0)
    seeds = pbr.frame_seeds(0, 6)
    ref = oracle.Renderer(sc.desc, cfg, threads=4)
    want = ref.render(0, seeds, px, cam)
    dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(cfg)
    for sched in ("refill", "tile", "wavefront"):
        os.environ["PBR_SCHEDULE"] = sched
        dev.reset_accum()
        attempt("brdf%d %s fused" % (brdf, sched), lambda: dev.render(0, seeds, px, cam))
        got = dev.read_output()
        print("brdf%d %-6s fused 6 frames: %.3f ms bit-exact=%s counters-equal=%s guard=%s" % (
            brdf, sched, dev.last_kernel_ms(), np.array_equal(got, want, equal_nan=True), dev.counters() == ref.counter_dict(), dev.guard_trips()), flush=True)
        dbg_ok = np.array_equal(dev.read_debug(), ref.debug, equal_nan=True)
        dev.reset_accum()
        def frames():
            for k, s in enumerate(seeds):
                dev.render_frame(float(s), float(np.float32(k) / np.float32(k + 1)), px, cam)
                if k + 1 < len(seeds): dev.accumulate()
        attempt("brdf%d %s stepwise" % (brdf, sched), frames)
        print("brdf%d %-6s frame-by-frame: bit-exact=%s debug-image=%s" % (brdf, sched, np.array_equal(dev.read_output(), want, equal_nan=True), dbg_ok), flush=True)
    dev.close()
del os.environ["PBR_SCHEDULE"]

pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": 8})
sc = pbr.HostScene.generate("cornell")
W, H = 1920, 1080
cfg = sc.config(W, H); cam = sc.camera(); px = pbr.pixel_dimension(W, H, 45.0)
dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(cfg)
for sched in ("refill", "tile", "wavefront"):
    os.environ["PBR_SCHEDULE"] = sched
    for frames in (16,):
        dev.reset_accum()
        attempt("1080p %s %d" % (sched, frames), lambda: dev.render(0, pbr.frame_seeds(0, frames), px, cam), 120.0)
        print("1080p depth8 %-6s %3d frames: kernel %.2f ms -> %.1f Msamples/s" % (sched, frames, dev.last_kernel_ms(), W * H * frames / dev.last_kernel_ms() / 1e3), flush=True)
print(dev.counters())
os._exit(0)
