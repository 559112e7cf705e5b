"""Lock-step kernels: how many lanes of a wave wait with a finished unit before the wave refills them together
(knob refill_batch; 1 = every lane at once, as up to round 2).  usage: python scripts/sweep_refill.py [scene:frames:plan ...]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
W, H = 1920, 1080
for job in (sys.argv[1:] or ["cornell:64:5", "cornell:64:0", "cornell:1:5", "sponza:32:5", "sponza:32:1"]):
    name, frames, plan = job.split(":"); frames = int(frames); plan = int(plan)
    kind, seed, tris, depth = SCENES[name]
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris)
    cam, px = sc.camera(), pbr.pixel_dimension(W, H)
    dev = pbr.Device(0); dev.pin_plan(plan); dev.upload_scene(sc.desc); dev.configure(sc.config(W, H))
    dev.render(0, pbr.frame_seeds(0, 32), px, cam)
    row = []
    for batch in (1, 4, 8, 16, 24, 32, 48):
        dev.set_knob("refill_batch", batch)
        best = 1e9
        for rep in range(3):
            dev.reset_accum()
            dev.render(0, pbr.frame_seeds(0, frames), px, cam)
            best = min(best, dev.last_trace()[0])
        digest = hashlib.sha1(np.ascontiguousarray(dev.read_output()).tobytes()).hexdigest()[:8]
        row.append("%2d: %7.1f %s" % (batch, W * H * frames / best / 1e3, digest))
    print("%-8s %3d frames %-12s | refill_batch -> Msamples/s | %s" % (name, frames, dev.last_plan()[0], "  ".join(row)), flush=True)
    dev.close()
