"""pbr_denoise at 1080p: device time of feature pass + filter, and the mean squared error against a converged frame before
and after, per sample count.  usage: python scripts/denoise_demo.py [scene] [triangles] [--sweep]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
kind = args[0] if args else "sponza"
tris = int(args[1]) if len(args) > 1 else 260000
W, H = 1920, 1080
pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": 3})
sc = pbr.HostScene.generate(kind, 2, tris)
cam, px = sc.camera(), pbr.pixel_dimension(W, H)
dev = pbr.Device(0)
dev.upload_scene(sc.desc); dev.configure(sc.config(W, H))
dev.render(0, pbr.frame_seeds(5000, 1024), px, cam)
truth = dev.read_output()
ok = np.isfinite(truth[..., :3]).all(-1)
mse = lambda a: float(((a[..., :3] - truth[..., :3])[ok & np.isfinite(a[..., :3]).all(-1)] ** 2).mean())
sets = [pbr.DenoiseParams()]
if "--sweep" in sys.argv:
    sets = [pbr.DenoiseParams(sigma_color=c, sigma_normal=n, sigma_world=wd, sigma_albedo=a)
            for c in (0.3, 0.6, 1.2, 4.0) for n in (0.25, 0.5) for wd in (3.0,) for a in (0.1, 0.3)]
for spp in (1, 4, 16, 64):
    dev.reset_accum()
    dev.render(0, pbr.frame_seeds(0, spp), px, cam)
    noisy = dev.read_output()
    for p in sets:
        out = dev.denoise(px, cam, p)
        print("%-8s %4d spp  colour %.2f normal %.2f world %.1f albedo %.2f passes %d: mse %.3e -> %.3e (x%.1f)  %.2f ms on the device" % (
            kind, spp, p.sigma_color, p.sigma_normal, p.sigma_world, p.sigma_albedo, p.passes, mse(noisy), mse(out), mse(noisy) / mse(out), dev.last_kernel_ms()))
dev.close()
