"""Analysis aid (CPU, uses the oracle as a visit recorder): how many distinct 128-B lines of the cold part of the
node stream does a walk touch, and would a hot set chosen from MEASURED visit counts beat the surface-area ranking?  usage: python scripts/layout_study.py [scene] [rays]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
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


    # ---- round 3: treelets — the cold records packed so that a 128-B line holds a CONNECTED piece of the tree ----
    # (a node is visited iff its parent's box was hit: the visited set is prefix-closed, so lines that hold a parent with
    # its most-visited descendants are used by more visits than lines that hold four consecutive DFS indices)
    parent = np.full(N, -1, np.int64); stack2 = []
    for i in range(N):
        while stack2 and i >= stack2[-1][0]: stack2.pop()
        parent[i] = stack2[-1][1] if stack2 else -1
        if not leaf[i]:
            esc = link[i] if link[i] > i else N
            stack2.append((esc, i))
    children = [[] for _ in range(N)]
    for i in range(1, N):
        if parent[i] >= 0: children[parent[i]].append(i)
    for per_line in (4,):
        order_cold = []; holes = 0
        roots = [i for i in range(1, N) if not ishot[i] and (parent[i] < 0 or ishot[parent[i]] or parent[i] == 0)]
        import heapq
        todo = list(roots)[::-1]        # DFS-ish order of treelet roots
        line_fill = 0
        while todo:
            r = todo.pop()
            treelet = [r]; frontier = [(-weight[c], c) for c in children[r] if not ishot[c]]
            heapq.heapify(frontier)
            while frontier and len(treelet) < per_line:
                w_, c = heapq.heappop(frontier)
                treelet.append(c)
                for g in children[c]:
                    if not ishot[g]: heapq.heappush(frontier, (-weight[g], g))
            rest = sorted([c for _, c in frontier], reverse=True)
            todo.extend(rest)           # their subtrees next, first child first
            if line_fill + len(treelet) > per_line:      # does not fit the open line: pad it
                pad = per_line - line_fill
                order_cold.extend([-1] * pad); holes += pad; line_fill = 0
            order_cold.extend(treelet); line_fill = (line_fill + len(treelet)) % per_line
        # hot nodes' cold children that were never reached (none expected) and positions
        pos = np.full(N, -1, np.int64); pos[hotset] = np.arange(hot)
        # the cold part starts on a line boundary
        base = (hot + 3) // 4 * 4
        oc = np.array(order_cold, np.int64)
        pos[oc[oc >= 0]] = base + np.flatnonzero(oc >= 0)
        assert (pos[1:] >= 0).all(), int((pos[1:] < 0).sum())
        evaluate("treelets of %d by parent area (%d holes)" % (per_line, holes), pos, hot)
