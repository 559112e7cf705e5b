"""Threshold and LDS-share sweeps of the state-machine plans in a given mode (pbr_config.traversal / arith):
  python scripts/sweep_modes.py [scene:frames ...] [--traversal 2] [--arith 0] [--what thresholds,lds]
The thresholds of round 3 / 4 (phased-mid 16 / 40, phased-dual 28 / 48) were swept on the reference's walk; an ordered walk
makes fewer visits per leaf and per shading, so the balance between the phases moves."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbr_loader
pbr = pbr_loader.load()
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
ap = argparse.ArgumentParser()
ap.add_argument("jobs", nargs="*", default=["sponza:32", "dragon:32", "hairball:16"])
ap.add_argument("--traversal", type=int, default=2)
ap.add_argument("--arith", type=int, default=0)
ap.add_argument("--what", default="thresholds,lds")
ap.add_argument("--size", default="1920x1080")
args = ap.parse_args()
W, H = (int(v) for v in args.size.split("x"))


def rate(dev, frames, px, cam, reps=3):
    best = 1e9
    for rep in range(reps):
        dev.reset_accum()
        dev.render(0, pbr.frame_seeds(0, frames), px, cam)
        best = min(best, dev.last_trace()[0])
    return W * H * frames / best / 1e3


for job in args.jobs:
    name, frames = job.split(":"); frames = int(frames)
    kind, seed, tris, depth = SCENES[name]
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris)
    cam, px = sc.camera(), pbr.pixel_dimension(W, H)
    cfg = sc.config(W, H); cfg.traversal, cfg.arith = args.traversal, args.arith
    dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(cfg)
    for plan, pname, parks, shades in ((4, "phased-mid", (8, 12, 16, 20, 24, 32), (24, 32, 40, 48)), (6, "phased-dual", (16, 22, 28, 36, 44, 56), (32, 40, 48, 56))):
        dev.pin_plan(plan)
        dev.render(0, pbr.frame_seeds(0, 16), px, cam)
        if "thresholds" in args.what:
            for park in parks:
                row = []
                for shade in shades:
                    dev.set_knob("ph_park", park); dev.set_knob("ph_shade", shade)
                    row.append("%7.1f" % rate(dev, frames, px, cam))
                print("%-8s traversal %d arith %d %-11s park %2d | shade %s: %s Msamples/s" % (name, args.traversal, args.arith, pname, park, " / ".join(str(s) for s in shades), " ".join(row)), flush=True)
            dev.set_knob("ph_park", -1); dev.set_knob("ph_shade", -1)
        if "lds" in args.what:
            row = []
            slots = (0, 256, 512, 1024, 2048, 5112)
            for n in slots:
                dev.set_knob("lds_slots", n)
                row.append("%7.1f" % rate(dev, frames, px, cam))
            dev.set_knob("lds_slots", -1)
            print("%-8s traversal %d arith %d %-11s staged records <= %s: %s Msamples/s" % (name, args.traversal, args.arith, pname, " / ".join(str(s) for s in slots), " ".join(row)), flush=True)
    dev.close(); sc.close()
