import sys, os, time
sys.path.insert(0, "/root/repo")
import pbr_loader
pbr = pbr_loader.load()
for kind, seed, tris in (("sponza", 2, 260000), ("dragon", 1, 870000), ("hairball", 3, 2000000)):
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": 3})
    sc = pbr.HostScene.generate(kind, seed, tris)
    cam, px = sc.camera(), pbr.pixel_dimension(256, 144)
    dev = pbr.Device(0)
    t0 = time.perf_counter(); dev.upload_scene(sc.desc); t_up = time.perf_counter() - t0
    dev.pin_plan(4)
    out = []
    for mode in (0, 2, 1):
        cfg = sc.config(256, 144); cfg.traversal = mode
        dev.configure(cfg)
        t0 = time.perf_counter(); dev.render(0, pbr.frame_seeds(0, 1), px, cam); first = time.perf_counter() - t0
        t0 = time.perf_counter(); dev.render(1, pbr.frame_seeds(1, 1), px, cam); second = time.perf_counter() - t0
        out.append("traversal %d: first render %.3f s, next %.4f s" % (mode, first, second))
    print("%-9s %d nodes: upload %.3f s | %s | %s" % (kind, sc.desc.num_nodes, t_up, " | ".join(out), dev.scene_bytes()), flush=True)
    dev.close()
