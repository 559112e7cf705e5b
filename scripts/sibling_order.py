"""CPU study: would a cold node stream that stores SIBLINGS next to each other (a 64-byte-aligned
pair, never another line) make the adjacent-record fetch of round 4 (nodePhasePair, lab) pay?  It fetched the DFS
neighbour — the hit successor — which is used 48 % of the time and is another 128-byte line one time in four.  The right
sibling follows a node whenever that node is a leaf or its box is missed.

From the oracle's visit log: for the visits OUTSIDE the ranked prefix that is staged in LDS (the 5112 most-visited nodes by
the upload's ranking: those never reach memory), the share whose predecessor in the same walk was their previous sibling —
i.e. the share of cold fetches a sibling-pair fetch would have brought along already.  The bar set by the review: >= 60 %.

  python scripts/sibling_order.py [--rays 30000] > profiles/r05/experiments/sibling_order.txt
"""
import argparse
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbr_loader
from oracle import oracle

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=30000)
ap.add_argument("--scenes", default="sponza,dragon,hairball")
args = ap.parse_args()
pbr = pbr_loader.load()
ALL = {"sponza": ("sponza", 2, 260000), "dragon": ("dragon", 1, 870000), "hairball": ("hairball", 3, 2000000)}
lib = oracle.lib()
lib.orc_debug_set_visit_log.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
print("# %d rays per scene: primary rays of the scene's camera and their first two diffuse-like bounces (random directions from the hit points)" % args.rays)
for name in args.scenes.split(","):
    kind, seed, tris = ALL[name]
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": 3})
    sc = pbr.HostScene.generate(kind, seed, tris)
    arr = sc.arrays()
    bvh = arr["bvh"]
    N = len(bvh)
    leaf = bvh[:, 3] >= 0
    # the tree: end of every subtree, parent, previous sibling in DFS order
    end = np.zeros(N, np.int64); parent = np.full(N, -1, np.int64); prev_sib = np.full(N, -1, np.int64)
    stack = []
    for i in range(N):
        while stack and i >= end[stack[-1]]:
            stack.pop()
        if stack:
            parent[i] = stack[-1]
        if leaf[i]:
            end[i] = i + 1
        else:
            link = int(bvh[i, 7])
            end[i] = link if link > i else (end[stack[-1]] if stack else N)
            stack.append(i)
    last_child = {}
    for i in range(1, N):
        p = parent[i]
        if p in last_child:
            prev_sib[i] = last_child[p]
        last_child[p] = i
    # the ranked prefix, as pbr_upload_scene ranks it (parent's surface area)
    ext = np.abs(bvh[:, 4:7].astype(np.float64) - bvh[:, 0:3])
    area = 2 * (ext[:, 0] * ext[:, 1] + ext[:, 2] * ext[:, 1] + ext[:, 0] * ext[:, 2])
    weight = np.where(parent >= 0, area[np.maximum(parent, 0)], area[0])
    order = np.lexsort((np.arange(N), -weight)); order = order[order != 0]
    hot = np.zeros(N, bool); hot[order[:5112]] = True

    # rays: camera rays through random pixels, then two generations of random-direction rays from their hit points
    cam = sc.camera()
    rng = np.random.default_rng(7)
    n0 = args.rays // 3
    eye = np.array([cam.eye.x, cam.eye.y, cam.eye.z]); w = np.array([cam.w.x, cam.w.y, cam.w.z]); u = np.array([cam.u.x, cam.u.y, cam.u.z]); v = np.array([cam.v.x, cam.v.y, cam.v.z])
    px = pbr.pixel_dimension(1920, 1080)
    sx, sy = rng.uniform(-960, 960, n0) * px, rng.uniform(-540, 540, n0) * px
    d = w[None, :] + sx[:, None] * u[None, :] + sy[:, None] * v[None, :]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([np.tile(eye, (n0, 1)), d], axis=1).astype(np.float32)
    cfg = sc.config(64, 64)
    gens = [rays]
    for g in range(2):
        t, face, normal, counts = oracle.trace_rays(sc.desc, cfg, gens[-1])
        ok = np.isfinite(t)
        o = gens[-1][ok, :3] + gens[-1][ok, 3:] * t[ok, None]
        nd = rng.normal(size=(ok.sum(), 3)); nd /= np.linalg.norm(nd, axis=1, keepdims=True)
        flip = np.einsum("ij,ij->i", nd, normal[ok]) < 0
        nd[flip] *= -1
        gens.append(np.concatenate([o + 1e-4 * nd, nd], axis=1).astype(np.float32))
    rays = np.concatenate(gens)
    for mode, label in ((0, "reference order"), (2, "eight orders")):
        cfg = sc.config(64, 64); cfg.traversal = mode
        cap = 2000 * len(rays)
        log = np.zeros(cap, np.int32); count = ctypes.c_uint64(0)
        lib.orc_debug_set_visit_log(log.ctypes.data, cap, ctypes.byref(count))
        t, face, normal, counts = oracle.trace_rays(sc.desc, cfg, rays)
        lib.orc_debug_set_visit_log(None, 0, None)
        total = int(count.value); assert total < cap
        visits = log[:total].astype(np.int64)
        per_ray = counts[:, 0].astype(np.int64); assert per_ray.sum() == total
        first = np.zeros(total, bool); first[np.concatenate([[0], np.cumsum(per_ray)[:-1]])] = True
        prev = np.concatenate([[-1], visits[:-1]]); prev[first] = -1
        cold = ~hot[visits]
        if mode == 0:
            after_sibling = prev == prev_sib[visits]
        else:
            # in an ordered walk the "previous sibling" is the node whose next-link leads here and which shares the parent
            after_sibling = (prev >= 0) & (parent[np.maximum(prev, 0)] == parent[visits])
        after_parent = prev == parent[visits]
        print("%-9s %-16s %7.1f visits / ray, %4.1f %% of them outside the staged prefix; of those: predecessor = previous sibling %5.1f %%, = parent (first child of a hit box) %5.1f %%, = deeper in the sibling's subtree %5.1f %%" % (
            name, label, total / len(rays), 100 * cold.mean(), 100 * after_sibling[cold].mean(), 100 * after_parent[cold].mean(),
            100 * (1 - after_sibling[cold].mean() - after_parent[cold].mean())), flush=True)
    sc.close()
