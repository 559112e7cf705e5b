"""A/B harness: times the fused launch on the four bench scenes for the library in PBR_HIP_LIB
(or the default build) and prints a hash of every output image, so variants can be compared for
speed AND for bit-identical results.  usage: python scripts/ab.py [scene:frames ...]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import numpy as np
import pbr_loader
pbr = pbr_loader.load()

SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
jobs = sys.argv[1:] or ["cornell:64", "sponza:32", "dragon:32", "hairball:16"]
tag = os.path.basename(os.environ.get("PBR_HIP_LIB", "default")) + "/t%s a%s/" % (os.environ.get("AB_TRAVERSAL", "0"), os.environ.get("AB_ARITH", "0")) + ("plan " + os.environ["PBR_PLAN"] if os.environ.get("PBR_PLAN") else "auto")
W, H = (int(v) for v in os.environ.get("AB_SIZE", "1920x1080").split("x"))
for job in jobs:
    name, frames = job.split(":"); frames = int(frames)
    kind, seed, tris, depth = SCENES[name]
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris)
    cfg, cam, px = sc.config(W, H), sc.camera(), pbr.pixel_dimension(W, H)
    cfg.traversal, cfg.arith = int(os.environ.get("AB_TRAVERSAL", "0")), int(os.environ.get("AB_ARITH", "0"))     # pbr_config's opt-in modes
    dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(cfg)
    dev.render(0, pbr.frame_seeds(0, 112), px, cam)                     # warm-up (and schedule auto-tuning: 6 x 2 + up to 3 x 32 frames)
    times = []
    for rep in range(2):
        dev.reset_accum()
        c0 = dev.counters()
        dev.render(0, pbr.frame_seeds(0, frames), px, cam)
        c1 = dev.counters()
        times.append(dev.last_kernel_ms())
    img = dev.read_output()
    digest = hashlib.sha1(np.ascontiguousarray(img).tobytes()).hexdigest()[:12]
    best = min(times)
    import ctypes as _ct
    plan = _ct.create_string_buffer(48); tuned = _ct.c_int(-2)
    if hasattr(pbr.hip, "pbr_diag_last_plan"):
        pbr.hip.pbr_diag_last_plan.argtypes = [_ct.c_void_p, _ct.c_char_p, _ct.c_size_t, _ct.POINTER(_ct.c_int)]
        pbr.hip.pbr_diag_last_plan(dev._ctx, plan, 48, _ct.byref(tuned))
    tag2 = tag + " " + plan.value.decode()
    visits = c1["nodes"] - c0["nodes"]
    print("%-46s %-9s %3d frames  %9.2f ms  %8.1f Msamples/s  %6.1f G visits/s (%5.1f nodes %4.1f tris /sample)  sha1 %s" % (
        tag2, name, frames, best, W * H * frames / best / 1e3, visits / best / 1e6, visits / (W * H * frames),
        (c1["tris"] - c0["tris"]) / (W * H * frames), digest), flush=True)
    import ctypes
    raw = (ctypes.c_uint64 * 16)()
    if hasattr(pbr.hip, "pbr_diag_raw_counters") and os.environ.get("PBR_STATS"):
        pbr.hip.pbr_diag_raw_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        pbr.hip.pbr_diag_raw_counters(dev._ctx, raw)
        if os.environ["PBR_STATS"] == "leafhist":       # lab build -DPBR_EXP_LEAFHIST (lab/src/pt_lab_hooks.hpp)
            hist, phases, parked, finished, entered = [raw[4 + k] for k in range(8)], raw[12], raw[13], raw[14], raw[15]
            print("    leaf phases by parked lanes  0: %.1f %%  1-4: %.1f %%  5-8: %.1f %%  9-12: %.1f %%  13-16: %.1f %%  17-24: %.1f %%  25-32: %.1f %%  33+: %.1f %%" % tuple(100.0 * h / max(phases, 1) for h in hist))
            print("    node phases %.3e: %.1f lanes enter, %.2f park, %.2f end their walk per phase" % (phases, entered / max(phases, 1), parked / max(phases, 1), finished / max(phases, 1)), flush=True)
            dev.close()
            continue
        it, act, lit, lact = raw[4], raw[5], raw[6], raw[7]
        if it:
            print("    stats: wave-iterations %.3e  active lanes/iteration %.1f  iterations with a leaf %.1f %%  lanes in the leaf branch %.2f" % (
                it, act / it, 100.0 * lit / it, lact / max(lit, 1)), flush=True)
        if raw[15]:
            span = raw[13] - ((~raw[14]) & 0xFFFFFFFFFFFFFFFF)
            print("    tail: %d waves, launch span %.3f ms (100 MHz clock), mean wave busy %.1f %% of the span" % (raw[15], span / 1e5, 100.0 * raw[12] / (raw[15] * span)), flush=True)
        if raw[8]:
            print("    phased: node iterations %.3e (%.1f lanes, %.1f per phase)  leaf phases %.3e (%.1f lanes)  shade phases %.3e (%.1f lanes)" % (
                raw[8], raw[9] / raw[8], raw[8] / max(raw[14], 1), raw[10], raw[11] / max(raw[10], 1), raw[12], raw[13] / max(raw[12], 1)), flush=True)
    dev.close()
