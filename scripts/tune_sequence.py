import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/pbr_loader.py') else os.getcwd())
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import numpy as np, pbr_loader
pbr = pbr_loader.load()
W, H = 1920, 1080
pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": 8})
sc = pbr.HostScene.generate("cornell", 1, 0)
dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(sc.config(W, H))
cam, px = sc.camera(), pbr.pixel_dimension(W, H)
for k in range(40):
    dev.render(k, pbr.frame_seeds(k, 1), px, cam)
print("after 40 single-frame renders:", dev.last_plan())
dev.render(40, pbr.frame_seeds(40, 256), px, cam)
print("first 256-frame render:", dev.last_plan(), dev.last_trace())
dev.render(296, pbr.frame_seeds(296, 256), px, cam)
print("second 256-frame render:", dev.last_plan(), dev.last_trace())
a = dev.read_output()
# same frames in one go on a fresh context
dev2 = pbr.Device(0); dev2.upload_scene(sc.desc); dev2.configure(sc.config(W, H))
os.environ["PBR_PLAN"] = "0"
dev2.render(0, pbr.frame_seeds(0, 552), px, cam)
b = dev2.read_output()
print("bit-identical to one forced-plan render of 552 frames:", np.array_equal(a.view(np.uint32), b.view(np.uint32)))
