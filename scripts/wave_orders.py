"""Which ORDERS do the 64 paths of a wave walk at one time?  CPU study with the oracle.

In the eight-order walk a ray reads the stream of its direction's sign octant.  The lanes of a wave start with 64 frames of one
pixel — camera rays of one octant — and diverge from the first bounce on; the state machine re-forms its lanes at every shade
phase, so it COULD take up paths sorted by order (lanes that read the same stream next to each other).  Whether that can pay
is a question about the mix: if a wave's walking lanes are in <= 3 distinct orders on average there is little to sort.

Model: the oracle (single thread, eight orders) logs every closest-hit walk (order, depth, node visits).  A wave = 64 lanes;
lane l renders frame l of the pixels of an 8 x 8 tile one after the other (the dealing order of nextSlot: a pixel through all
frames of the launch, then the next pixel), a walk lasts its node visits, a shading step `SHADE` visit-times.  Sampled every 64
visit-times: the walking lanes' orders — how many distinct ones, and the share of the most common one.

usage: python scripts/wave_orders.py [scene ...] > profiles/r06/experiments/wave_orders.txt
"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbr_loader
from oracle import oracle

pbr = pbr_loader.load()
W, H, FRAMES, SHADE = 64, 40, 64, 60
SCENES = {"sponza": ("sponza", 2, 260000), "dragon": ("dragon", 1, 870000), "hairball": ("hairball", 3, 2000000), "cornell": ("cornell", 1, 0)}

for name in (sys.argv[1:] or ["sponza", "dragon", "hairball"]):
    kind, seed, tris = SCENES[name]
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": 8 if kind == "cornell" else 3})
    sc = pbr.HostScene.generate(kind, seed, tris)
    cfg, cam, px = sc.config(W, H), sc.camera(), pbr.pixel_dimension(W, H)
    cfg.traversal = 2
    ref = oracle.Renderer(sc.desc, cfg, threads=1)
    cap = 64 * W * H * FRAMES
    log = np.zeros(cap, np.uint32)
    count = ctypes.c_uint32(0)
    oracle.lib().orc_debug_set_walk_log.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    t0 = time.time()
    per_frame = []
    for f, s in enumerate(pbr.frame_seeds(0, FRAMES)):
        count.value = 0
        oracle.lib().orc_debug_set_walk_log(log.ctypes.data, cap, ctypes.addressof(count))
        ref.image = ref.render_frame(float(s), f / (f + 1.0), px, cam)
        per_frame.append(log[:count.value].copy())
    oracle.lib().orc_debug_set_walk_log(None, 0, None)
    # paths[f][p] = (orders, visits) per walk; a path starts at depth 0, pixels in row-major order (one thread)
    paths = []
    for words in per_frame:
        k, depth, visits = words & 15, (words >> 4) & 255, words >> 12
        starts = np.flatnonzero(depth == 0)
        assert len(starts) == W * H * int(cfg.samples), (len(starts), W * H)
        paths.append([(k[a:b], visits[a:b]) for a, b in zip(starts, list(starts[1:]) + [len(words)])])
    distinct_all, share_all, distinct_late, share_late, walking = [], [], [], [], []
    first_bounce_share = []
    for ty in range(H // 8):
        for tx in range(W // 8):
            pixels = [(ty * 8 + y) * W + tx * 8 + x for y in range(8) for x in range(8)]
            lanes = []
            for lane in range(64):
                ks, ds = [], []
                for p in pixels:
                    order, visits = paths[lane][p]
                    for b, (o, v) in enumerate(zip(order, visits)):
                        ks.append(np.full(int(v), int(o), np.int8)); ds.append(np.full(int(v), min(b, 1), np.int8))
                        ks.append(np.full(SHADE, -1, np.int8)); ds.append(np.full(SHADE, 0, np.int8))
                lanes.append((np.concatenate(ks), np.concatenate(ds)))
            length = min(len(l[0]) for l in lanes)
            for t in range(0, length, 64):
                k = np.array([l[0][t] for l in lanes]); late = np.array([l[1][t] for l in lanes]) == 1
                on = k >= 0
                walking.append(on.sum())
                if on.sum() >= 8:
                    c = np.bincount(k[on], minlength=8)
                    distinct_all.append((c > 0).sum()); share_all.append(c.max() / on.sum())
                if (on & late).sum() >= 8:
                    c = np.bincount(k[on & late], minlength=8)
                    distinct_late.append((c > 0).sum()); share_late.append(c.max() / (on & late).sum())
    all_k = np.concatenate([w & 15 for w in per_frame]); all_d = np.concatenate([(w >> 4) & 255 for w in per_frame]); all_v = np.concatenate([w >> 12 for w in per_frame])
    hist_late = np.bincount(all_k[all_d > 0], weights=all_v[all_d > 0], minlength=8)
    print("%-8s %dx%d x %d frames (%.0f s): walking lanes per wave %.1f of 64;  distinct orders among them %.2f (most common order's share %.2f);  "
          "among the lanes past their first bounce %.2f (share %.2f);  visits after bounce 1 by order: %s;  camera rays' visits: %.0f %% of all" % (
              name, W, H, FRAMES, time.time() - t0, np.mean(walking), np.mean(distinct_all), np.mean(share_all), np.mean(distinct_late), np.mean(share_late),
              " ".join("%.0f%%" % (100 * h / hist_late.sum()) for h in hist_late), 100 * all_v[all_d == 0].sum() / all_v.sum()), flush=True)
