"""A/B harness: times the fused launch on the four bench scenes for the library in PBR_HIP_LIB
(or the default build) and prints a hash of every output image, so variants can be compared for
speed AND for bit-identical results.  usage: python scripts/ab.py [scene:frames ...]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbr_loader
pbr = pbr_loader.load()

SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
jobs = sys.argv[1:] or ["cornell:64", "sponza:32", "dragon:32", "hairball:16"]
tag = os.path.basename(os.environ.get("PBR_HIP_LIB", "default")) + "/" + os.environ.get("PBR_SCHEDULE", "auto")
W, H = 1920, 1080
for job in jobs:
    name, frames = job.split(":"); frames = int(frames)
    kind, seed, tris, depth = SCENES[name]
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris)
    cfg, cam, px = sc.config(W, H), sc.camera(), pbr.pixel_dimension(W, H)
    dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(cfg)
    dev.render(0, pbr.frame_seeds(0, 4), px, cam)                      # warm-up
    times = []
    for rep in range(2):
        dev.reset_accum()
        dev.render(0, pbr.frame_seeds(0, frames), px, cam)
        times.append(dev.last_kernel_ms())
    img = dev.read_output()
    digest = hashlib.sha1(np.ascontiguousarray(img).tobytes()).hexdigest()[:12]
    best = min(times)
    print("%-34s %-9s %3d frames  %9.2f ms  %8.1f Msamples/s  sha1 %s" % (tag, name, frames, best, W * H * frames / best / 1e3, digest), flush=True)
    dev.close()
