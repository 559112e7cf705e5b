import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
hip = pbr.hip
hip.pbr_diag_trace_stream.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)]
fp = ctypes.POINTER(ctypes.c_float)
N = 8 * 1024 * 1024
rng = np.random.default_rng(0)
ALL = {"sponza": ("sponza", 2, 260000), "dragon": ("dragon", 1, 870000), "hairball": ("hairball", 3, 2000000)}
WANT = [a for a in sys.argv[1:] if a in ALL] or list(ALL)
MODES = [int(a) for a in sys.argv[1:] if a.isdigit()] or [0, 512, 1024, 2048, 2552]
for kind, seed, tris in [ALL[k] for k in WANT]:
    pbr.cfg_reset()
    sc = pbr.HostScene.generate(kind, seed, tris)
    v = sc.arrays()["vertices"][:, :3]
    lo, hi = v.min(0), v.max(0)
    rays = np.zeros((N, 8), np.float32)
    rays[:, 0:3] = rng.uniform(lo, hi, (N, 3))
    d = rng.normal(size=(N, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 4:7] = d
    dev = pbr.Device(0); dev.upload_scene(sc.desc)
    for mode in MODES:
        out = np.empty((N, 2), np.float32); ms = ctypes.c_double()
        before = dev.counters()
        st = hip.pbr_diag_trace_stream(dev._ctx, mode, rays.ctypes.data_as(fp), N, 3, out.ctypes.data_as(fp), ctypes.byref(ms))
        assert st == 0, hip.pbr_last_error(dev._ctx)
        after = dev.counters()
        nodes = (after["nodes"] - before["nodes"]) / 3; tr = (after["tris"] - before["tris"]) / 3
        print("%-9s lds-slots %4d: %8.2f ms  %7.1f Mrays/s  %6.1f nodes/ray %5.1f tris/ray  %7.1f G node-visits/s  hit-rate %.2f  sum_t %.6e" % (
            kind, mode, ms.value, N / ms.value / 1e3, nodes / N, tr / N, nodes / ms.value / 1e6, np.isfinite(out[:, 0]).mean(), np.where(np.isfinite(out[:,0]), out[:,0], 0).astype(np.float64).sum()), flush=True)
    dev.close()
