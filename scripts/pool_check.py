"""Pooled schedule (PBR_PLAN=6) against the state machine (PBR_PLAN=4): same bits?  how fast?
usage: PBR_HIP_LIB=lab/libpbrhip_pool.so python scripts/pool_check.py [scene:w:h:frames ...]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
for job in (sys.argv[1:] or ["cornell:64:48:3"]):
    name, w, h, frames = job.split(":"); w, h, frames = int(w), int(h), int(frames)
    kind, seed, tris, depth = SCENES[name]
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris if w >= 1000 else min(tris, 20000))
    cfg, cam, px = sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h)
    out = {}
    for plan in (4, 6):
        os.environ["PBR_PLAN"] = str(plan)
        dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(cfg)
        dev.render(0, pbr.frame_seeds(0, frames), px, cam)
        best = 1e9
        for rep in range(2):
            dev.reset_accum()
            dev.render(0, pbr.frame_seeds(0, frames), px, cam)
            best = min(best, dev.last_trace()[0])
        img, dbg, c = dev.read_output(), dev.read_debug(), dev.counters()
        if plan == 6 and os.environ.get("PBR_STATS"):
            import ctypes
            raw = (ctypes.c_uint64 * 16)()
            pbr.hip.pbr_diag_raw_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            pbr.hip.pbr_diag_raw_counters(dev._ctx, raw)
            print("    pool: walker loops idle %.3e / busy %.3e (%.1f lanes with a ray), node iterations %.3e (%.1f per loop); shader polls empty %.3e, batches %.3e of %.1f paths" % (
                raw[4], raw[5], raw[6] / max(raw[5], 1), raw[7], raw[7] / max(raw[5], 1), raw[8], raw[9], raw[10] / max(raw[9], 1)), flush=True)
        out[plan] = (img, dbg, c)
        print("%-8s %dx%d x%d  %-12s %9.3f ms  %8.1f Msamples/s  guard %s  counters %s" % (name, w, h, frames, dev.last_plan()[0], best, w * h * frames / best / 1e3, dev.guard_trips(), c), flush=True)
        dev.close()
    same = np.array_equal(out[4][0], out[6][0], equal_nan=True) and np.array_equal(out[4][1], out[6][1], equal_nan=True) and out[4][2] == out[6][2]
    print("    bit-identical images, debug images and counters:", same, flush=True)
