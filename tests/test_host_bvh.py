"""Host side of the path: BVH builder replica + flattening (the wire format the kernels walk).
CPU only.  Structural invariants of source/accelstructures/BVH.cpp and
PathTracer::initOpenCLBuffers_BVH (PathTracer.cpp:238-347)."""
import os

import numpy as np
import pytest

from conftest import REFERENCE_MODELS


def surface_area(lo, hi):
    d = np.abs(hi - lo)
    return 2.0 * (d[0] * d[1] + d[2] * d[1] + d[0] * d[2])


def walk_all_hit(bvh):
    """The stackless walk of pt_bvh.cl:82-123 when EVERY box is hit: visits nodes 1, 2, 3, ..."""
    n = len(bvh)
    order, index = [], 1
    while 0 < index < n:
        order.append(index)
        index += 1
    return order


def walk_all_miss_from(bvh, start):
    """Follow miss links only, from `start`."""
    n = len(bvh)
    seen, index = [], start
    while 0 < index < n:
        seen.append(index)
        node = bvh[index]
        index = int(node[7]) if node[3] <= -1.0 else index + 1
    return seen


@pytest.mark.parametrize("kind,triangles", [("cornell", 0), ("sponza", 8000), ("dragon", 6000), ("hairball", 4000)])
@pytest.mark.parametrize("skip", [True, False])
def test_flat_bvh_invariants(cfg_defaults, kind, triangles, skip):
    pbr = cfg_defaults
    pbr.cfg_set(**{"bvh.skip_ahead": skip})
    sc = pbr.HostScene.generate(kind, 3, triangles)
    arr, info = sc.arrays(), sc.info
    bvh, facesV, verts = arr["bvh"], arr["facesV"], arr["vertices"][:, :3]
    n = len(bvh)

    assert info["flat_nodes"] == n == info["tree_nodes"] - info["skipped"]
    assert (info["skipped"] > 0) == skip or kind == "cornell" and not skip
    assert info["tree_nodes"] == 2 * info["leaves"] - 1          # proper binary tree
    assert bvh[0, 3] == -1.0 and bvh[0, 7] == -1.0               # root: container, no link

    leaf = bvh[:, 3] >= 0
    inner = ~leaf
    assert np.all(bvh[inner, 3] == -1.0)

    # leaves address faces k, k+1 in leaf order and cover every face exactly once (max_faces = 2)
    first = bvh[leaf, 3].astype(int)
    second = bvh[leaf, 7].astype(int)
    assert np.all((second == -1) | (second == first + 1))
    covered = np.concatenate([first, second[second >= 0]])
    assert sorted(covered.tolist()) == list(range(len(facesV)))

    # every leaf box is the exact min / max of its faces' corners
    for node in bvh[leaf][:200]:
        ids = [int(node[3])] + ([int(node[7])] if node[7] >= 0 else [])
        pts = verts[facesV[ids, :3].ravel()]
        assert np.array_equal(pts.min(0), node[0:3]) and np.array_equal(pts.max(0), node[4:7])

    # miss links: integers in [-1, n), always forward (the walk never revisits a node)
    links = bvh[inner, 7]
    assert np.all(links == np.floor(links)) and np.all(links >= -1) and np.all(links < n)
    idx = np.nonzero(inner)[0]
    fwd = links[1:] if inner[0] else links
    assert np.all((fwd == -1) | (fwd > idx[1:] if inner[0] else fwd > idx))

    # a node's miss link skips exactly its subtree: everything between node and link lies inside its box
    for i in idx[1:50]:
        link = int(bvh[i, 7])
        end = link if link > 0 else n
        sub = bvh[i + 1:end]
        if len(sub):
            assert np.all(sub[:, 0:3] >= bvh[i, 0:3] - 1e-6) and np.all(sub[:, 4:7] <= bvh[i, 4:7] + 1e-6)

    # all-miss from node 1 terminates and only moves forward
    seen = walk_all_miss_from(bvh, 1)
    assert seen == sorted(seen) and len(seen) <= n


def test_children_ordered_by_surface_area_without_skip(cfg_defaults):
    """combineNodes (BVH.cpp:335-343): the bigger child is stored first (directly after its parent)."""
    pbr = cfg_defaults
    pbr.cfg_set(**{"bvh.skip_ahead": False})
    sc = pbr.HostScene.generate("sponza", 1, 6000)
    bvh = sc.arrays()["bvh"]
    inner = np.nonzero(bvh[:, 3] < 0)[0]
    checked = 0
    for i in inner:
        left = i + 1
        # right sibling of the left child: its miss link (container) or the node after its leaf
        right = int(bvh[left, 7]) if bvh[left, 3] < 0 else left + 1
        if 0 < right < len(bvh):
            sa_l = surface_area(bvh[left, 0:3], bvh[left, 4:7])
            sa_r = surface_area(bvh[right, 0:3], bvh[right, 4:7])
            assert sa_l >= sa_r
            checked += 1
    assert checked > 100


def test_skip_ahead_removes_marked_left_children(cfg_defaults):
    pbr = cfg_defaults
    sc_on = pbr.HostScene.generate("dragon", 1, 5000)
    pbr.cfg_set(**{"bvh.skip_ahead_compare": 1.01})      # nothing can qualify
    sc_off = pbr.HostScene.generate("dragon", 1, 5000)
    assert sc_off.info["skipped"] == 0
    assert sc_on.info["skipped"] > 0
    assert sc_on.info["tree_nodes"] == sc_off.info["tree_nodes"]
    assert sc_on.info["flat_nodes"] == sc_off.info["flat_nodes"] - sc_on.info["skipped"]
    # same faces in the same leaf order either way
    assert np.array_equal(sc_on.arrays()["facesV"], sc_off.arrays()["facesV"])


def test_mean_split_above_sah_limit(cfg_defaults):
    """Above bvh.sah_faces_limit the builder splits at the mean centre (BVH.cpp:255-273)."""
    pbr = cfg_defaults
    pbr.cfg_set(**{"bvh.sah_faces_limit": 500})
    sc = pbr.HostScene.generate("dragon", 1, 4000)
    assert sc.info["faces"] >= 3900
    arr = sc.arrays()
    assert sorted(np.concatenate([arr["bvh"][arr["bvh"][:, 3] >= 0, 3], arr["bvh"][arr["bvh"][:, 7] >= 0][arr["bvh"][arr["bvh"][:, 7] >= 0][:, 3] >= 0, 7]]).astype(int).tolist()) == list(range(sc.info["faces"]))


def test_generators_are_reproducible(cfg_defaults):
    pbr = cfg_defaults
    a = pbr.HostScene.generate("hairball", 7, 3000).arrays()
    b = pbr.HostScene.generate("hairball", 7, 3000).arrays()
    c = pbr.HostScene.generate("hairball", 8, 3000).arrays()
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert not np.array_equal(a["vertices"], c["vertices"])


def test_scene_sizes_track_the_budget(cfg_defaults):
    pbr = cfg_defaults
    for kind, want in (("dragon", 20000), ("sponza", 30000), ("hairball", 20000)):
        got = pbr.HostScene.generate(kind, 1, want).info["faces"]
        assert 0.75 * want < got < 1.25 * want, (kind, got)
    assert pbr.HostScene.generate("cornell").info["faces"] == 48
    with pytest.raises(pbr.PbrError):
        pbr.HostScene.generate("teapot")


@pytest.mark.skipif(not os.path.isdir(REFERENCE_MODELS), reason="reference assets only exist in the build container")
def test_suzanne_known_answer(cfg_defaults):
    """The only known answer in the reference tree: writeDebugImage normalises by '1082 // number of faces in the
    test model' and '1265 // number of BVH nodes in the test model' (pathtracing.cl:75-76).  The face count is
    reproduced (suzanne.obj, 10 objects).  The node count is what a tree with 633 leaves has (2 L - 1, however the
    ten per-object trees are grouped); the builder as BVH.cpp stands today gives 622 leaves, and none of the 640
    settings of scripts/bvh_sweep.py (every config.json knob x three readings of the source, table in
    profiles/r02/bvh_sweep_suzanne.txt) gives 1265 for any count a 'number of BVH nodes' could mean — the comment
    predates the CHANGELOG's 'Faster BVH construction'.  What CAN be held: the tree is a proper binary tree over all
    1082 faces with one or two faces per leaf, i.e. between 541 and 1082 leaves, and 1265 lies in that range."""
    pbr = cfg_defaults
    sc = pbr.HostScene.load_obj(REFERENCE_MODELS, "suzanne.obj")
    assert sc.info["faces"] == 1082
    assert sc.info["objects"] == 10
    leaves, tree = sc.info["leaves"], sc.info["tree_nodes"]
    assert tree == 2 * leaves - 1 and 541 <= leaves <= 1082 and 2 * 541 - 1 <= 1265 <= 2 * 1082 - 1
    assert abs(tree - 1265) <= 0.02 * 1265                      # within 2 % of the quoted count (1243 vs 1265)
    # ... and, as a regression value of THIS builder (not a reference-held answer): 622 leaves, 1243 tree nodes — a
    # change of the replica that moves these must be a deliberate one
    assert (leaves, tree) == (622, 1243)
    assert sc.info["flat_nodes"] == tree - sc.info["skipped"]


@pytest.mark.skipif(not os.path.isdir(REFERENCE_MODELS), reason="reference assets only exist in the build container")
def test_phong_tessellation_grows_the_face_boxes(cfg_defaults):
    """MathHelp::triCalcAABB (MathHelp.cpp:263-309): with render.phong_tessellation > 0 a face with unequal vertex
    normals gets a box that also covers the tessellated patch; flat faces keep the box of their corners."""
    pbr = cfg_defaults
    flat = pbr.HostScene.load_obj(REFERENCE_MODELS, "suzanne.obj").arrays()
    pbr.cfg_set(**{"render.phong_tessellation": 1.0})
    grown = pbr.HostScene.load_obj(REFERENCE_MODELS, "suzanne.obj").arrays()

    def leaf_boxes(arr):
        out = {}
        leaf = arr["bvh"][:, 3] >= 0
        for node in arr["bvh"][leaf]:
            ids = [int(node[3])] + ([int(node[7])] if node[7] >= 0 else [])
            key = tuple(sorted(tuple(arr["facesV"][i, :3].tolist()) for i in ids))
            out[key] = (node[0:3].copy(), node[4:7].copy(), ids)
        return out

    a, b = leaf_boxes(flat), leaf_boxes(grown)
    verts = grown["vertices"][:, :3]
    bigger = same = 0
    for key, (lo, hi, ids) in b.items():
        pts = verts[np.array(key).ravel()]
        assert np.all(lo <= pts.min(0)) and np.all(hi >= pts.max(0))          # never smaller than the corners' box
        if np.array_equal(lo, pts.min(0)) and np.array_equal(hi, pts.max(0)):
            same += 1
        else:
            bigger += 1
            assert np.all(pts.min(0) - lo < 0.2) and np.all(hi - pts.max(0) < 0.2)   # a bulge, not an explosion
    assert bigger > 300 and same > 50                                           # the monkey is smooth, the walls are flat
    for key, (lo, hi, ids) in a.items():                                        # alpha = 0: exactly the corners' box
        pts = flat["vertices"][:, :3][np.array(key).ravel()]
        assert np.array_equal(lo, pts.min(0)) and np.array_equal(hi, pts.max(0))
