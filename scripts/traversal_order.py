"""CPU study (oracle only, no GPU): what a ray-ordered walk over the SAME flat BVH saves.

For each large scene, a small frame is rendered by the oracle in the reference's order (traversal 0), with six orders
(1: dominant axis x sign) and with eight (2: sign octant, each container sorted on its own axis); node visits and face
tests per path, the longest single walk, and how the images differ.  VERDICT r04 item 1's numbers to reproduce:
hairball 0.66x visits / 0.56x face tests, Dragon-class 0.79x / 0.70x, Sponza-class 0.97x / 0.87x (six orders).

    python scripts/traversal_order.py [--width 480 --height 272 --frames 2] > profiles/r05/experiments/traversal_order.txt
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbr_loader
from oracle import oracle

SCENES = [("sponza", 2, 260000), ("dragon", 1, 870000), ("hairball", 3, 2000000)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--height", type=int, default=272)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--scenes", default="sponza,dragon,hairball")
    args = ap.parse_args()
    pbr = pbr_loader.load()
    pbr.cfg_reset()
    pbr.cfg_set(**{"render.max_depth": 3})
    print("# oracle, %dx%d, %d frames, depth 3, BRDF 1; per path: node visits (box tests) / face tests; longest single walk" % (args.width, args.height, args.frames))
    print("# 'stack, nearest first' is an analysis aid (oracle only, cfg.traversal = oracle.STACK_AID): every hit container's children box-tested, visited nearest first by entry distance — the bound a stackless ordered walk can approach")
    for kind, seed, tris in SCENES:
        if kind not in args.scenes.split(","):
            continue
        scene = pbr.HostScene.generate(kind, seed=seed, triangles=tris)
        cam, px = scene.camera(), pbr.pixel_dimension(args.width, args.height)
        seeds = pbr.frame_seeds(0, args.frames)
        rows = {}
        for mode, name in ((0, "reference order"), (1, "six orders"), (2, "eight orders"), (oracle.STACK_AID, "stack, nearest first")):
            cfg = scene.config(args.width, args.height)
            cfg.traversal = mode
            t0 = time.time()
            ref = oracle.Renderer(scene.desc, cfg, threads=args.threads)
            longest = ctypes.c_uint32(0)
            oracle.lib().orc_debug_set_walk_max(ctypes.addressof(longest))
            img = ref.render(0, seeds, px, cam).copy()
            oracle.lib().orc_debug_set_walk_max(None)
            c = ref.counter_dict()
            rows[mode] = (c, longest.value, img, ref.debug.copy())
            base = rows[0][0]
            same = np.all((img == rows[0][2]) | (np.isnan(img) & np.isnan(rows[0][2])), axis=2)
            d = np.abs(img[..., :3].astype(np.float64) - rows[0][2][..., :3])
            print("%-9s %-16s nodes %8.2f (%.3fx)  faces %6.2f (%.3fx)  hits/path %.4f  longest walk %5d (%.3fx)  pixels bit-identical to the reference order %.4f %%  max |d| %.3g  mean |d| %.3g  [%.1f s]" % (
                kind, name, c["nodes"] / c["paths"], c["nodes"] / base["nodes"], c["tris"] / c["paths"], c["tris"] / base["tris"],
                c["hits"] / c["paths"], longest.value, longest.value / rows[0][1], 100.0 * same.mean(), np.nanmax(d), np.nanmean(d), time.time() - t0))
            sys.stdout.flush()
        scene.close()


if __name__ == "__main__":
    main()
