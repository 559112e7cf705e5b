"""Where a wave of the lane state machine spends its life (lab build -DPBR_EXP_PHASE_TIME: shader-clock deltas around the
node phase, the leaf phase and the shade phase, summed over the waves).
usage: PBR_HIP_LIB=lab/libpbrhip_ptime.so PBR_PLAN=4 python scripts/phase_time.py [scene:frames ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import pbr_loader
pbr = pbr_loader.load()
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
W, H = 1920, 1080
for job in (sys.argv[1:] or ["sponza:32", "dragon:32", "hairball:16", "cornell:64"]):
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
    node, leaf, shade, life = raw[4], raw[5], raw[6], raw[7]
    print("%-8s traversal %d arith %d %3d frames %s %.2f ms: of a wave's life %.1f %% node phases, %.1f %% leaf phases, %.1f %% shade phases, %.1f %% the rest" % (
        name, cfg.traversal, cfg.arith, frames, dev.last_plan()[0], dev.last_trace()[0], 100.0 * node / life, 100.0 * leaf / life, 100.0 * shade / life, 100.0 * (life - node - leaf - shade) / life), flush=True)
    dev.close()
