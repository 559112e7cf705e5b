"""Re-sweep of the lane state machine's two thresholds (lanes that leave a node phase before it ends / lanes that wait
before a shade phase runs) on the round's kernels.  usage: python scripts/sweep_thresholds.py [scene:frames ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import pbr_loader
pbr = pbr_loader.load()
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
W, H = 1920, 1080
for job in (sys.argv[1:] or ["sponza:32", "dragon:32", "hairball:16"]):
    name, frames = job.split(":"); frames = int(frames)
    kind, seed, tris, depth = SCENES[name]
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris)
    cam, px = sc.camera(), pbr.pixel_dimension(W, H)
    dev = pbr.Device(0); dev.pin_plan(4); dev.upload_scene(sc.desc); dev.configure(sc.config(W, H))
    dev.render(0, pbr.frame_seeds(0, 32), px, cam)
    for park in (8, 12, 16, 20, 24):
        row = []
        for shade in (24, 32, 40, 48):
            dev.set_knob("ph_park", park); dev.set_knob("ph_shade", shade)
            best = 1e9
            for rep in range(3):
                dev.reset_accum()
                dev.render(0, pbr.frame_seeds(0, frames), px, cam)
                best = min(best, dev.last_trace()[0])
            row.append("%7.1f" % (W * H * frames / best / 1e3))
        print("%-8s phased-mid park %2d | shade 24 / 32 / 40 / 48: %s Msamples/s" % (name, park, " ".join(row)), flush=True)
    dev.close()
