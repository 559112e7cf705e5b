import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import pbr_loader
pbr = pbr_loader.load()
W, H = 1920, 1080
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3)}
name = sys.argv[1]; frames = int(sys.argv[2]); world = int(sys.argv[3])
kind, seed, tris, depth = SCENES[name]
pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
sc = pbr.HostScene.generate(kind, seed, tris)
cam, px = sc.camera(), pbr.pixel_dimension(W, H)
for plan in range(7):
    os.environ["PBR_PLAN"] = str(plan)
    cfg = sc.config(W, H); cfg.tile_world, cfg.tile_rank = world, 0
    dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(cfg)
    dev.render(0, pbr.frame_seeds(0, 32), px, cam)
    out = []
    for n in (8, 32, frames):
        best = 1e9
        for rep in range(2):
            dev.render(32, pbr.frame_seeds(32, n), px, cam)
            best = min(best, dev.last_trace()[0])
        out.append("%d: %.3f" % (n, best))
    print(name, "N=%d" % world, dev.last_plan()[0], " | ".join(out), flush=True)
    dev.close()
