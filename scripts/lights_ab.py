"""Lab: the kernels for scenes WITH lights (LIGHTS = true) got 20 - 32 B of scratch (SGPR spill slots) with the published word of
empty queue heads.  Does it cost anything?  The Sponza-class scene with an orb light and a point light (tests/test_gpu_full_configs.py),
shadow rays off / on, the 6-waves state machine, the two-paths plan and the lock-step 6-waves plan, 20 frames, best of 7.
usage: PBR_HIP_LIB=<library> python scripts/lights_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
W, H, FRAMES = 1920, 1080, 20
pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": 3, "render.brdf": 1})
sc = pbr.HostScene.generate("sponza", 2, 260000)
cam, px = sc.camera(), pbr.pixel_dimension(W, H)
v = sc.arrays()["vertices"][:, :3]
centre, size = (v.min(0) + v.max(0)) / 2, (v.max(0) - v.min(0))
lights = np.zeros((2, 12), np.float32)
lights[0] = [centre[0], centre[1] + 0.2 * size[1], centre[2], 0, 4.0, 3.5, 3.0, 0, 2, 0.05 * float(size.max()), 0, 0]
lights[1] = [centre[0] - 0.2 * size[0], centre[1], centre[2] + 0.1 * size[2], 0, 1, 1, 1, 0, 1, 0, 0, 0]
desc = pbr.SceneDesc.from_buffer_copy(sc.desc)
desc.lights, desc.num_lights = lights.ctypes.data, 2
for shadow in (0, 1):
    for plan in (4, 6, 5):
        cfg = sc.config(W, H); cfg.shadow_rays = shadow
        dev = pbr.Device(0); dev.pin_plan(plan); dev.upload_scene(desc); dev.configure(cfg)
        dev.render(0, pbr.frame_seeds(0, 2), px, cam)
        best = 1e9
        for rep in range(7):
            dev.render(2, pbr.frame_seeds(2, FRAMES), px, cam)
            best = min(best, dev.last_trace()[0])
        print("%-40s lights 2 shadow rays %d  %-12s %8.3f ms for %d frames  %7.1f Msamples/s  (%s)" % (
            os.environ.get("PBR_HIP_LIB", "product"), shadow, dev.last_plan()[0], best, FRAMES, W * H * FRAMES / best / 1e3, dev.last_plan()[1] if len(dev.last_plan()) > 1 else ""), flush=True)
        dev.close()
