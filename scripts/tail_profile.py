"""How a launch of the lane state machine ends (lab build -DPBR_EXP_TAIL): per wave, start / first empty queue / end on
the 100 MHz wall clock.  usage: PBR_HIP_LIB=lab/libpbrhip_tail.so PBR_PLAN=4 python scripts/tail_profile.py [scene:frames ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
W, H = 1920, 1080
M = (1 << 64) - 1
for job in (sys.argv[1:] or ["sponza:1", "sponza:8", "sponza:64", "dragon:1", "dragon:8", "hairball:1", "cornell:1"]):
    name, frames = job.split(":"); frames = int(frames)
    kind, seed, tris, depth = SCENES[name]
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris)
    cfg = sc.config(W, H)
    cfg.traversal, cfg.arith = int(os.environ.get("AB_TRAVERSAL", "0")), int(os.environ.get("AB_ARITH", "0"))     # pbr_config's opt-in modes
    dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(cfg)
    cam, px = sc.camera(), pbr.pixel_dimension(W, H)
    dev.render(0, pbr.frame_seeds(0, 16), px, cam)
    dev.reset_accum()
    dev.render(0, pbr.frame_seeds(0, frames), px, cam)
    raw = (ctypes.c_uint64 * 16)()
    pbr.hip.pbr_diag_raw_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    pbr.hip.pbr_diag_raw_counters(dev._ctx, raw)
    waves = raw[15]
    first_start = (~raw[14]) & M
    last_start = raw[7]
    end = raw[13]
    span = end - first_start
    first_dry = (~raw[6]) & M
    us = lambda t: t / 100.0
    print("%-8s traversal %d %3d frame(s) %s kernel %.3f ms | %d waves, span %.0f us, last wave starts at +%.0f us, queue first empty at +%.0f us (%.0f %% of the span), "
          "mean wave busy %.1f %% of the span, mean drain per wave %.0f us, longest drain %.0f us" % (
              name, cfg.traversal, frames, dev.last_plan()[0], dev.last_trace()[0], waves, us(span), us(last_start - first_start), us(first_dry - first_start),
              100.0 * (first_dry - first_start) / span, 100.0 * raw[12] / (waves * span), us(raw[4] / max(waves, 1)), us(raw[5])), flush=True)
    dev.close()
