"""GPU A/B of the traversal modes (pbr_config.traversal): for each scene, a small whole-frame parity check of every mode
against the oracle in the same mode (image, debug image, counters), then the timing of every mode with the plan the tuner
keeps and with selected plans pinned.

  python scripts/walk_ab.py [scene:frames ...] [--modes 0,1,2] [--plans auto,phased-mid,...] [--size 1920x1080]
"""
import argparse
import hashlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbr_loader
from oracle import oracle

pbr = pbr_loader.load()
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
PLANS = ["refill-lean", "refill-wide", "phased-lean", "phased-wide", "phased-mid", "refill-mid", "phased-dual"]
MODE = {0: "reference", 1: "six-order", 2: "eight-order"}

ap = argparse.ArgumentParser()
ap.add_argument("jobs", nargs="*", default=["sponza:32", "dragon:32", "hairball:16"])
ap.add_argument("--modes", default="0,1,2")
ap.add_argument("--plans", default="auto")
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--no-parity", action="store_true")
args = ap.parse_args()
W, H = (int(v) for v in args.size.split("x"))
modes = [int(m) for m in args.modes.split(",")]

for job in args.jobs:
    name, frames = job.split(":"); frames = int(frames)
    kind, seed, tris, depth = SCENES[name]
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris)
    cam = sc.camera()
    dev = pbr.Device(0); dev.upload_scene(sc.desc)
    base = {}
    for mode in modes:
        if not args.no_parity:
            w, h = 256, 144
            cfg = sc.config(w, h); cfg.traversal = mode
            px = pbr.pixel_dimension(w, h)
            seeds = pbr.frame_seeds(0, 2)
            ref = oracle.Renderer(sc.desc, cfg, threads=os.cpu_count() or 8)
            want = ref.render(0, seeds, px, cam)
            dev.configure(cfg)
            verdicts = []
            for plan in (4, 5, 6):
                dev.pin_plan(plan); dev.reset_accum()
                c0 = dev.counters()
                dev.render(0, seeds, px, cam)
                got, dbg = dev.read_output(), dev.read_debug()
                spent = {k: v - c0[k] for k, v in dev.counters().items()}
                ok = np.array_equal(got, want, equal_nan=True) and np.array_equal(dbg, ref.debug, equal_nan=True) and spent == ref.counter_dict()
                verdicts.append("%s %s" % (PLANS[plan], "==" if ok else "DIFFERS"))
                if not ok:
                    print("   ", spent, ref.counter_dict(), "image equal:", np.array_equal(got, want, equal_nan=True))
            dev.pin_plan(-1)
            print("%-9s %-11s parity vs oracle(%s) at %dx%d: %s" % (name, MODE[mode], MODE[mode], w, h, ", ".join(verdicts)), flush=True)
        cfg = sc.config(W, H); cfg.traversal = mode
        px = pbr.pixel_dimension(W, H)
        for plan in args.plans.split(","):
            dev.configure(cfg)
            dev.pin_plan(-1 if plan == "auto" else PLANS.index(plan))
            dev.render(0, pbr.frame_seeds(0, 224 if plan == "auto" else 16), px, cam)     # warm-up (and the tuner's budget)
            times = []
            for rep in range(3):
                dev.reset_accum()
                c0 = dev.counters()
                dev.render(0, pbr.frame_seeds(0, frames), px, cam)
                c1 = dev.counters()
                times.append(dev.last_kernel_ms())
            img = dev.read_output()
            digest = hashlib.sha1(np.ascontiguousarray(img).tobytes()).hexdigest()[:12]
            best = min(times)
            rate = W * H * frames / best / 1e3
            key = plan
            base.setdefault(key, rate)
            n = W * H * frames
            print("%-9s %-11s %-6s -> %-12s %3d frames %9.2f ms %8.1f Msamples/s (%.3fx of reference order)  %6.1f nodes %5.1f tris /sample  sha1 %s" % (
                name, MODE[mode], plan, dev.last_plan()[0], frames, best, rate, rate / base[key],
                (c1["nodes"] - c0["nodes"]) / n, (c1["tris"] - c0["tris"]) / n, digest), flush=True)
    dev.close(); sc.close()
