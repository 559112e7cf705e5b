"""First-light check on a GPU box: HIP vs oracle on the Cornell scene, then a short 1080p timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
from oracle import oracle

def same(a, b):
    return np.array_equal(a, b, equal_nan=True)

def compare(name, a, b):
    ok = same(a, b)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    d = d[np.isfinite(d)]
    bad = np.sum(~((a == b) | (np.isnan(a) & np.isnan(b))))
    print("%-40s %s  mismatching values %d / %d  max|d| %.3g" % (name, "BIT-EXACT" if ok else "DIFFERENT", bad, a.size, d.max() if d.size else 0.0), flush=True)
    return ok

for brdf in (1, 0):
    pbr.cfg_reset()
    pbr.cfg_set(**{"render.max_depth": 4, "render.brdf": brdf})
    sc = pbr.HostScene.generate("cornell")
    W = H = 64
    cfg = sc.config(W, H); cam = sc.camera(); px = pbr.pixel_dimension(W, H, 45.0)
    seeds = pbr.frame_seeds(0, 4)
    ref = oracle.Renderer(sc.desc, cfg, threads=8)
    want = ref.render(0, seeds, px, cam).copy()
    dev = pbr.Device(0)
    dev.upload_scene(sc.desc); dev.configure(cfg)
    dev.render(0, seeds, px, cam)
    got = dev.read_output()
    compare("brdf %d fused 4 frames 64x64" % brdf, got, want)
    print("   counters hip", dev.counters(), "\n   counters orc", ref.counter_dict())
    compare("brdf %d debug image" % brdf, dev.read_debug(), ref.debug)
    # frame by frame
    dev.reset_accum()
    for k, s in enumerate(seeds):
        dev.render_frame(float(s), float(np.float32(k) / np.float32(k + 1)), px, cam)
        if k + 1 < len(seeds):
            dev.accumulate()
    compare("brdf %d frame-by-frame" % brdf, dev.read_output(), want)
    dev.close()

pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": 8})
sc = pbr.HostScene.generate("cornell")
W, H = 1920, 1080
cfg = sc.config(W, H); cam = sc.camera(); px = pbr.pixel_dimension(W, H, 45.0)
dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(cfg)
for frames in (1, 4, 16, 64):
    t = time.time(); dev.render(0, pbr.frame_seeds(0, frames), px, cam); dt = time.time() - t
    print("1080p depth8 %3d frames: wall %.3f s, kernel %.3f ms -> %.1f Msamples/s" % (frames, dt, dev.last_kernel_ms(), W * H * frames / dev.last_kernel_ms() / 1e3), flush=True)
print(dev.counters())
