"""The device builder's search radius (knob ploc_radius; default 32) under the ray-ordered walk: round 2 found the radius
"noisy on the hairball — the stackless walk's fixed child order decides more there than the tree's area"; an ordered walk takes
the child order out of the picture, so the radius should now move visits and speed monotonically.
usage: python scripts/ploc_radius_ordered.py [scene ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbr_loader
pbr = pbr_loader.load()
SCENES = {"sponza": ("sponza", 2, 260000, 3), "dragon": ("dragon", 1, 870000, 3), "hairball": ("hairball", 3, 2000000, 3)}
W, H, FRAMES = 1920, 1080, 32
for name in (sys.argv[1:] or ["dragon", "hairball"]):
    kind, seed, tris, depth = SCENES[name]
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris)
    arr = sc.arrays()
    cam, px = sc.camera(), pbr.pixel_dimension(W, H)
    dev = pbr.Device(0)
    for radius in (int(r) for r in os.environ.get("PLOC_RADII", "4,8,16,32,64").split(",")):
        dev.set_knob("ploc_radius", radius)
        nodes, fv, fn = dev.build_bvh(arr["vertices"], arr["facesV"], arr["facesN"])
        build_ms = dev.last_kernel_ms()
        desc = pbr.SceneDesc.from_buffer_copy(sc.desc)
        desc.bvh, desc.num_nodes, desc.facesV, desc.facesN = nodes.ctypes.data, nodes.shape[0], fv.ctypes.data, fn.ctypes.data
        row = []
        for traversal in (0, 2):
            cfg = sc.config(W, H); cfg.traversal = traversal
            dev.upload_scene(desc); dev.configure(cfg); dev.pin_plan(4)
            dev.render(0, pbr.frame_seeds(0, 16), px, cam)
            best = 1e9
            for rep in range(3):
                dev.reset_accum(); c0 = dev.counters()
                dev.render(0, pbr.frame_seeds(0, FRAMES), px, cam)
                c1 = dev.counters(); best = min(best, dev.last_trace()[0])
            row.append("%s %7.1f Msamples/s %6.1f visits %5.1f tris" % (("reference order", "eight orders")[traversal == 2], W * H * FRAMES / best / 1e3,
                                                                     (c1["nodes"] - c0["nodes"]) / (W * H * FRAMES), (c1["tris"] - c0["tris"]) / (W * H * FRAMES)))
        print("%-9s radius %2d build %6.2f ms | %s | %s" % (name, radius, build_ms, row[0], row[1]), flush=True)
    dev.close()
