#!/usr/bin/env python3
"""Which builder settings give the node count the reference quotes for its test model?

pathtracing.cl:75-76 divides the debug counters by "1082 // number of faces in the test model" and "1265 // number of
BVH nodes in the test model"; resources/models/testing/suzanne.obj has 1082 faces.  This sweeps the builder replica
(host/bvh_builder.cpp) over every knob config.json exposes (bvh.max_faces, bvh.sah_faces_limit, bvh.skip_ahead,
bvh.skip_ahead_compare, render.phong_tessellation) and three readings of the source that could differ between
toolchains or revisions (stable vs unstable sort of equal centres, one tree for the whole scene instead of one per
object, ties in the SAH sweep going to the last split instead of the first), and prints every count a "number of BVH
nodes" could mean: tree nodes (BVH::getNodes().size(), the "[BVH] ... Contains %lu nodes" log line), flat nodes
(bvhNodesCL.size() = #BVH_NUM_NODES#, after skip-ahead deletion), leaves, containers.

Container only (needs /root/reference).  Result: DESIGN.md section 3.
"""
import ctypes
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import pbr_loader  # noqa: E402

MODELS = "/root/reference/resources/models/testing/"
TARGET = 1265


def main():
    pbr = pbr_loader.load()
    pbr.host.pbrh_lab_bvh_flags.argtypes = [ctypes.c_uint]
    model = sys.argv[1] if len(sys.argv) > 1 else "suzanne.obj"
    rows, hits = [], []
    space = itertools.product(
        [1, 2, 3, 4],                         # bvh.max_faces ("Must be [1,2]", config.json:42; the code clamps only below)
        [100000, 500, 100, 20, 2],            # bvh.sah_faces_limit (mean split above it)
        [0.0, 0.3, 0.6, 1.0],                 # render.phong_tessellation (grows the face boxes, MathHelp.cpp:263-309)
        range(8),                             # lab flags: 1 stable sort, 2 one tree, 4 last best split
    )
    for max_faces, sah_limit, alpha, flags in space:
        pbr.cfg_reset()
        pbr.cfg_set(**{"bvh.max_faces": max_faces, "bvh.sah_faces_limit": sah_limit, "render.phong_tessellation": alpha,
                       "bvh.skip_ahead": True})
        pbr.host.pbrh_lab_bvh_flags(flags)
        try:
            sc = pbr.HostScene.load_obj(MODELS, model)
        except pbr.PbrError as exc:           # max_faces > 2 leaves cannot be flattened (PathTracer.cpp:266-268 addresses two)
            rows.append((max_faces, sah_limit, alpha, flags, "rejected: %s" % str(exc)[:60]))
            continue
        info = sc.info
        tree, leaves = info["tree_nodes"], info["leaves"]
        # flat count per skip_ahead_compare: rebuild only the marks (cheap: same tree)
        flats = {}
        for cmp_ in (0.5, 0.6, 0.7, 0.8, 0.9, 1.0):
            pbr.cfg_set(**{"bvh.skip_ahead_compare": cmp_})
            flats[cmp_] = pbr.HostScene.load_obj(MODELS, model).info["flat_nodes"]
        counts = {"tree": tree, "leaves": leaves, "containers": tree - leaves, "tree-1": tree - 1}
        counts.update({"flat@%.1f" % k: v for k, v in flats.items()})
        rows.append((max_faces, sah_limit, alpha, flags, counts))
        for what, value in counts.items():
            if value == TARGET:
                hits.append((max_faces, sah_limit, alpha, flags, what))
    pbr.host.pbrh_lab_bvh_flags(0)
    pbr.cfg_reset()
    print("%-9s %-9s %-5s %-5s  counts" % ("max_faces", "sah_limit", "alpha", "flags"))
    for r in rows:
        print("%-9d %-9d %-5.1f %-5d  %s" % (r[0], r[1], r[2], r[3], r[4]))
    print("\nsettings that give %d: %s" % (TARGET, hits if hits else "NONE in %d combinations" % len(rows)))


if __name__ == "__main__":
    main()
