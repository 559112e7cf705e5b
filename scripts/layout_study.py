"""Analysis aid (CPU, uses the oracle as a visit recorder): how many distinct 128-B lines of the cold part of the
node stream does a walk touch, and would a hot set chosen from MEASURED visit counts beat the surface-area ranking?  usage: python scripts/layout_study.py [scene] [rays]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbr_loader
pbr = pbr_loader.load()
from oracle import oracle

kind = sys.argv[1] if len(sys.argv) > 1 else "dragon"
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
ALL = {"sponza": ("sponza", 2, 260000), "dragon": ("dragon", 1, 870000), "hairball": ("hairball", 3, 2000000), "cornell": ("cornell", 1, 0)}
k, seed, tris = ALL[kind]
pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": 3})
sc = pbr.HostScene.generate(k, seed, tris)
arr = sc.arrays()
bvh = arr["bvh"]                      # (N, 8): min.xyz, w0, max.xyz, w1
N = bvh.shape[0]
leaf = bvh[:, 3] >= 0
link = np.where(leaf, -1, bvh[:, 7]).astype(np.int64)
v = arr["vertices"][:, :3]; lo, hi = v.min(0), v.max(0)
rng = np.random.default_rng(1)
rays = np.zeros((n_rays, 6), np.float32)
rays[:, 0:3] = rng.uniform(lo, hi, (n_rays, 3))
d = rng.normal(size=(n_rays, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays[:, 3:6] = d

lib = oracle.lib()
cap = 400 * n_rays
log = np.zeros(cap, np.int32); count = ctypes.c_uint64(0)
lib.orc_debug_set_visit_log.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
lib.orc_debug_set_visit_log(log.ctypes.data, cap, ctypes.byref(count))
os.environ["OMP_NUM_THREADS"] = "1"
t, face, normal, counts = oracle.trace_rays(sc.desc, sc.config(64, 64), rays)   # trace_rays is single-threaded
lib.orc_debug_set_visit_log(None, 0, None)
total = int(count.value); assert total < cap
visits = log[:total]
per_ray = counts[:, 0].astype(np.int64)
assert per_ray.sum() == total, (per_ray.sum(), total)
starts = np.concatenate([[0], np.cumsum(per_ray)[:-1]])
print("%s: %d nodes, %d rays, %.1f visits/ray" % (kind, N, n_rays, total / n_rays))

# ---- hot ranking as pbr_upload_scene does it (parent surface area) ----
ext = np.abs(bvh[:, 4:7].astype(np.float64) - bvh[:, 0:3]); area = 2 * (ext[:, 0] * ext[:, 1] + ext[:, 2] * ext[:, 1] + ext[:, 0] * ext[:, 2])
weight = np.zeros(N); stack = []
for i in range(N):
    while stack and i >= stack[-1][0]: stack.pop()
    weight[i] = stack[-1][1] if stack else area[0]
    if not leaf[i]:
        esc = link[i] if link[i] > i else N
        stack.append((esc, area[i]))
order_hot = np.lexsort((np.arange(N), -weight)); order_hot = order_hot[order_hot != 0]

def positions(order):
    pos = np.full(N, -1, np.int64); pos[order] = np.arange(len(order)); return pos

def evaluate(name, pos, hot):
    """pos: record index per node; the first `hot` records are in LDS (free).  Counts per ray the distinct
    128-B lines among the cold records it touches, and the share of cold visits whose line differs from
    the previous cold visit's."""
    p = pos[visits]
    cold = p >= hot
    line = p // 4
    distinct = 0; changes = 0
    for s, c in zip(starts, per_ray):
        l = line[s:s + c][cold[s:s + c]]
        if len(l):
            distinct += len(np.unique(l)); changes += 1 + int((l[1:] != l[:-1]).sum())
    print("  %-34s hot %5d: cold visits/ray %6.1f  distinct lines/ray %6.1f  line changes/ray %6.1f" % (
        name, hot, cold.sum() / n_rays, distinct / n_rays, changes / n_rays))

for hot in (2552, 5112):
    hotset = order_hot[:hot]; ishot = np.zeros(N, bool); ishot[hotset] = True
    rest_dfs = np.array([i for i in range(1, N) if not ishot[i]])
    evaluate("DFS (shipped)", positions(np.concatenate([hotset, rest_dfs])), hot)

    # measured-frequency hot set (upper bound for the ranking) with DFS rest
    freq = np.bincount(visits, minlength=N)
    best = np.argsort(-freq, kind="stable"); best = best[best != 0][:hot]
    isb = np.zeros(N, bool); isb[best] = True
    evaluate("DFS, hot set by measured visits", positions(np.concatenate([best, np.array([i for i in range(1, N) if not isb[i]])])), hot)

