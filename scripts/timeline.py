"""Live lanes over a launch's time (lab build -DPBR_EXP_TIMELINE=<bucket us>): how the lane state machine's launch ends.
usage: PBR_HIP_LIB=lab/libpbrhip_tl.so python scripts/timeline.py <bucket us> [scene:frames ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import pbr_loader
pbr = pbr_loader.load()
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
bucket = int(sys.argv[1])
W, H = 1920, 1080
for job in (sys.argv[2:] or ["sponza:1", "dragon:1", "cornell:1"]):
    name, frames = job.split(":"); frames = int(frames)
    kind, seed, tris, depth = SCENES[name]
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris)
    dev = pbr.Device(0); dev.pin_plan(int(os.environ.get("PBR_PLAN", "4")))
    dev.upload_scene(sc.desc); dev.configure(sc.config(W, H))
    cam, px = sc.camera(), pbr.pixel_dimension(W, H)
    dev.render(0, pbr.frame_seeds(0, 16), px, cam)
    raw0 = (ctypes.c_uint64 * 16)(); raw1 = (ctypes.c_uint64 * 16)()
    pbr.hip.pbr_diag_raw_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    dev.reset_accum()
    pbr.hip.pbr_diag_raw_counters(dev._ctx, raw0)
    dev.render(0, pbr.frame_seeds(0, frames), px, cam)
    pbr.hip.pbr_diag_raw_counters(dev._ctx, raw1)
    d = [raw1[i] - raw0[i] for i in range(16)]
    lanes = [d[4 + b] / (100.0 * bucket) for b in range(8)]          # lane-ticks / ticks per bucket = mean live lanes in the bucket
    print("%-8s %d frame(s) %s kernel %.3f ms | mean live lanes per %d-us bucket (k lanes): %s | loop rounds per %d us: %s" % (
        name, frames, dev.last_plan()[0], dev.last_trace()[0], bucket, " ".join("%6.1f" % (v / 1e3) for v in lanes), 2 * bucket, " ".join("%d" % v for v in d[12:16])), flush=True)
    dev.close()
