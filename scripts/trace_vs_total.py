import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import pbr_loader
pbr = pbr_loader.load()
W, H = 1920, 1080
for kind, seed, tris, depth, n in (("cornell", 1, 0, 8, 256), ("dragon", 1, 870000, 3, 64)):
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris)
    dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(sc.config(W, H))
    cam, px = sc.camera(), pbr.pixel_dimension(W, H)
    dev.render(0, pbr.frame_seeds(0, 112), px, cam)
    best = (1e9, 0)
    for rep in range(3):
        dev.render(112, pbr.frame_seeds(112, n), px, cam)
        best = min(best, (dev.last_trace()[0], dev.last_kernel_ms()))
    print(os.path.basename(os.environ.get("PBR_HIP_LIB", "")), kind, dev.last_plan()[0], "trace %.3f ms  total %.3f ms" % best, flush=True)
    dev.close()
