import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import pbr_loader
pbr = pbr_loader.load()
W, H = 1920, 1080
for kind, seed, tris in (("sponza", 2, 2000), ("dragon", 1, 20000), ("hairball", 3, 60000)):
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": 3})
    sc = pbr.HostScene.generate(kind, seed, tris)
    cam, px = sc.camera(), pbr.pixel_dimension(W, H)
    res = []
    for plan in ("auto", "0", "1", "2", "3", "4", "5"):
        if plan == "auto": os.environ.pop("PBR_PLAN", None)
        else: os.environ["PBR_PLAN"] = plan
        dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(sc.config(W, H))
        dev.render(0, pbr.frame_seeds(0, 112), px, cam)
        best = 1e9
        for rep in range(2):
            dev.render(112, pbr.frame_seeds(112, 64), px, cam); best = min(best, dev.last_kernel_ms())
        res.append("%s=%s %.0f" % (plan, dev.last_plan()[0], W * H * 64 / best / 1e3))
        dev.close()
    print(kind, tris, " | ".join(res), flush=True)
