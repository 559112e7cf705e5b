"""Device-built BVH (pbr_build_bvh; PBR_BVH_BUILDER=lbvh for round 1's radix tree) vs the host replica of the reference's builder: build time and render rate.
usage: python scripts/lbvh_compare.py [scene ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
SCENES = {"cornell": ("cornell", 1, 0, 8), "sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
W, H, FRAMES = 1920, 1080, 32
for name in (sys.argv[1:] or ["sponza", "dragon", "hairball"]):
    kind, seed, tris, depth = SCENES[name]
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    t0 = time.perf_counter(); sc = pbr.HostScene.generate(kind, seed, tris); host_s = time.perf_counter() - t0
    arr = sc.arrays()
    dev = pbr.Device(0)
    cam, px = sc.camera(), pbr.pixel_dimension(W, H)
    rates, builds = {}, {}
    # round 5: both trees also in the ray-ordered walk (pbr_config.traversal = eight orders) — a tree whose stored child order is
    # arbitrary (the clustering builder's) loses nothing to a walk that orders the children by the ray; pbr_build_bvh takes its
    # search radius from the traversal the context is configured with (32 / 3), so the device tree is built once per mode
    for traversal in (0, 2):
        cfg = sc.config(W, H)
        cfg.traversal = traversal
        dev.configure(cfg)
        t0 = time.perf_counter(); nodes, fv, fn = dev.build_bvh(arr["vertices"], arr["facesV"], arr["facesN"]); wall = time.perf_counter() - t0
        builds[traversal] = (dev.last_kernel_ms(), wall, nodes.shape[0])
        device_desc = pbr.SceneDesc.from_buffer_copy(sc.desc)
        device_desc.bvh, device_desc.num_nodes, device_desc.facesV, device_desc.facesN = nodes.ctypes.data, nodes.shape[0], fv.ctypes.data, fn.ctypes.data
        for label, desc in (("host SAH replica", sc.desc), ("device build", device_desc)):
            label += ", eight orders" if traversal else ""
            dev.upload_scene(desc); dev.configure(cfg)
            dev.render(0, pbr.frame_seeds(0, 2 * dev.tune_budget()), px, cam)     # the tuner's whole budget (a close call included), then some
            dev.reset_accum(); c0 = dev.counters()
            dev.render(0, pbr.frame_seeds(0, FRAMES), px, cam)
            c1 = dev.counters(); ms = dev.last_kernel_ms()
            rates[label] = (W * H * FRAMES / ms / 1e3, (c1["nodes"] - c0["nodes"]) / (W * H * FRAMES), dev.last_plan()[0])
    build_ms, wall, n_nodes = builds[0]
    print("%-9s %8d faces: host build (scene generation + SAH replica) %.1f s; device build %.2f ms on the device, %.0f ms with transfers, %d nodes" % (
        name, arr["facesV"].shape[0], host_s, build_ms, wall * 1e3, n_nodes))
    print("    (built for the ordered walk, radius 3: %.2f ms, %d nodes)" % (builds[2][0], builds[2][2]))
    for label, (rate, visits, plan) in rates.items():
        print("    %-32s %8.1f Msamples/s  %6.1f node visits/sample  (%s)" % (label, rate, visits, plan))
    dev.close()
