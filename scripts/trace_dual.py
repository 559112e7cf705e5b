"""[round 4 only — needs a lab library built from round 4's sources: `git worktree add ../r04 047ae5b`; round 5 removed the
-DPBR_LAB plumbing (pbr_lab_trace_stream_dual among it) from csrc/.]
Traversal-only probe, one walk per lane against two (lab build: scripts/lab.sh dual "").
   PBR_HIP_LIB=lab/libpbrhip_dual.so python3 scripts/trace_dual.py [sponza dragon hairball]
Rays: `coherent` = camera-like bundles (64 consecutive rays share an origin and differ by a small jitter), `random` =
uniform origins and directions inside the scene's box.  Every variant must give the same (t, face) per ray and the same counts."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
hip = pbr.hip
fp = ctypes.POINTER(ctypes.c_float)
hip.pbr_lab_trace_stream_dual.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, ctypes.c_uint32, ctypes.c_int, fp, ctypes.POINTER(ctypes.c_double)]
N = 8 * 1024 * 1024
rng = np.random.default_rng(0)
ALL = {"sponza": ("sponza", 2, 260000), "dragon": ("dragon", 1, 870000), "hairball": ("hairball", 3, 2000000)}
WANT = [a for a in sys.argv[1:] if a in ALL] or list(ALL)
for kind, seed, tris in [ALL[k] for k in WANT]:
    pbr.cfg_reset()
    sc = pbr.HostScene.generate(kind, seed, tris)
    v = sc.arrays()["vertices"][:, :3]
    lo, hi = v.min(0), v.max(0)
    dev = pbr.Device(0); dev.upload_scene(sc.desc)
    for flavour in ("random", "coherent"):
        rays = np.zeros((N, 8), np.float32)
        if flavour == "random":
            rays[:, 0:3] = rng.uniform(lo, hi, (N, 3))
            d = rng.normal(size=(N, 3))
        else:
            o = rng.uniform(lo, hi, (N // 64, 3)); d0 = rng.normal(size=(N // 64, 3))
            rays[:, 0:3] = np.repeat(o, 64, axis=0)
            d = np.repeat(d0 / np.linalg.norm(d0, axis=1, keepdims=True), 64, axis=0) + rng.normal(scale=0.01, size=(N, 3))
        rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
        first = None
        for waves, dual in ((4, 0), (6, 0), (8, 0), (4, 1), (6, 1)):
            hot = {4: 5112, 6: 2552, 8: 2552}[waves]
            out = np.empty((N, 2), np.float32); ms = ctypes.c_double()
            before = dev.counters()
            st = hip.pbr_lab_trace_stream_dual(dev._ctx, hot, waves, dual, rays.ctypes.data_as(fp), N, 3, out.ctypes.data_as(fp), ctypes.byref(ms))
            assert st == 0, hip.pbr_last_error(dev._ctx)
            after = dev.counters()
            nodes = (after["nodes"] - before["nodes"]) / 3; tr = (after["tris"] - before["tris"]) / 3
            same = "reference" if first is None else ("identical" if np.array_equal(out.view(np.uint32), first[0].view(np.uint32)) and (nodes, tr) == first[1] else "DIFFERENT")
            first = first or (out.copy(), (nodes, tr))
            print("%-9s %-8s rays  %d waves/SIMD x %d walk(s)/lane: %8.2f ms  %7.1f Mrays/s  %6.1f nodes/ray %5.1f tris/ray  %7.1f G node-visits/s  %s" % (
                kind, flavour, waves, 1 + dual, ms.value, N / ms.value / 1e3, nodes / N, tr / N, nodes / ms.value / 1e6, same), flush=True)
    dev.close()
