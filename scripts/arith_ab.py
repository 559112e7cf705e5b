"""GPU A/B of the arithmetic modes (pbr_config.arith): exact against native on the bench scenes — rate with the tuner's
plan, and how far the two accumulated images are apart (they agree statistically, not bit for bit).

  python scripts/arith_ab.py [scene:frames ...] [--traversal 0] [--size 1920x1080]
"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbr_loader

pbr = pbr_loader.load()
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
ap = argparse.ArgumentParser()
ap.add_argument("jobs", nargs="*", default=["cornell:64", "sponza:32", "dragon:32", "hairball:16"])
ap.add_argument("--traversal", default="0")
ap.add_argument("--size", default="1920x1080")
args = ap.parse_args()
W, H = (int(v) for v in args.size.split("x"))
for job in args.jobs:
    name, frames = job.split(":"); frames = int(frames)
    kind, seed, tris, depth = SCENES[name]
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris)
    cam, px = sc.camera(), pbr.pixel_dimension(W, H)
    dev = pbr.Device(0); dev.upload_scene(sc.desc)
    for trav in (int(t) for t in args.traversal.split(",")):
        images, base = {}, None
        for arith in (0, 1):
            cfg = sc.config(W, H); cfg.traversal = trav; cfg.arith = arith
            dev.configure(cfg)
            dev.render(0, pbr.frame_seeds(0, 224), px, cam)
            times = []
            for rep in range(3):
                dev.reset_accum()
                c0 = dev.counters()
                dev.render(0, pbr.frame_seeds(0, frames), px, cam)
                c1 = dev.counters()
                times.append(dev.last_kernel_ms())
            images[arith] = dev.read_output()
            rate = W * H * frames / min(times) / 1e3
            base = base or rate
            n = W * H * frames
            print("%-9s traversal %d arith %-6s -> %-12s %3d frames %8.2f ms %8.1f Msamples/s (%.3fx)  %6.1f nodes %5.1f tris %5.2f hits /sample  mean rgb %s" % (
                name, trav, ("exact", "native")[arith], dev.last_plan()[0], frames, min(times), rate, rate / base,
                (c1["nodes"] - c0["nodes"]) / n, (c1["tris"] - c0["tris"]) / n, (c1["hits"] - c0["hits"]) / n,
                np.array2string(np.nanmean(images[arith][..., :3], axis=(0, 1)), precision=5)), flush=True)
        d = np.abs(images[1][..., :3].astype(np.float64) - images[0][..., :3])
        print("%-9s traversal %d native vs exact at %d spp: mean |d| %.4g, 99th percentile %.4g, max %.4g; non-finite pixels %d / %d" % (
            name, trav, frames, np.nanmean(d), np.nanpercentile(d, 99), np.nanmax(d),
            int((~np.isfinite(images[1][..., :3])).any(axis=2).sum()), int((~np.isfinite(images[0][..., :3])).any(axis=2).sum())), flush=True)
    dev.close(); sc.close()
