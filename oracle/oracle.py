"""ctypes binding of the CPU oracle (oracle/pt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product never imports this module.  PARITY UNPINNED (see pt_oracle.h).

The oracle's structs are layout-identical to the C ABI's wire formats (both are the
reference's host structs, source/PathTracer.h:25-73), so a pbr_scene_desc / pbr_camera built
by the host library is handed to the oracle by reinterpretation, never by conversion.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_SRC = [os.path.join(_HERE, "pt_oracle.c"), os.path.join(_HERE, "pt_oracle.h")]


def build(force=False):
    stale = force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in _SRC)
    if stale:
        tmp = "%s.%d.tmp" % (_LIB, os.getpid())       # built aside and moved into place: never a half-written library under its name
        try:
            subprocess.check_call([
                "gcc", "-O2", "-std=c11", "-ffp-contract=off", "-mfma", "-mavx2", "-fopenmp", "-fPIC", "-shared",
                "-o", tmp, _SRC[0], "-lm"])
            os.replace(tmp, _LIB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return _LIB


def _cpu_tag():
    """A name for THIS machine's CPU: a library built with -march=native must never be loaded on another one (built
    libraries travel with the tree to the GPU box)."""
    import hashlib
    try:
        with open("/proc/cpuinfo") as f:
            text = f.read()
        lines = [l for l in text.splitlines() if l.startswith(("model name", "flags"))][:2]
    except OSError:
        lines = []
    import platform
    return hashlib.sha1(("\n".join(lines) + platform.machine()).encode()).hexdigest()[:10]


def build_native():
    """The baseline leg's build (bench.py, cpu_baseline): -O3 -march=native, as SURVEY 8(d) planned for the CPU timing —
    compiled ON the machine that times it, under a name that carries that machine's CPU, next to the portable library the
    tests check everything against.  -ffp-contract=off stays: same bits (tests/test_oracle_kernel.py compares the two)."""
    target = os.path.join(_HERE, "liboracle_native_%s.so" % _cpu_tag())
    stale = not os.path.exists(target) or any(os.path.getmtime(s) > os.path.getmtime(target) for s in _SRC)
    if stale:
        tmp = "%s.%d.tmp" % (target, os.getpid())
        try:
            subprocess.check_call([
                "gcc", "-O3", "-march=native", "-std=c11", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared",
                "-o", tmp, _SRC[0], "-lm"])
            os.replace(tmp, target)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return target


class OrcConfig(ctypes.Structure):
    _fields_ = [
        ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("brdf", ctypes.c_int32),
        ("shadow_rays", ctypes.c_int32), ("max_depth", ctypes.c_int32), ("max_added_depth", ctypes.c_int32),
        ("samples", ctypes.c_int32), ("num_nodes", ctypes.c_int32), ("num_lights", ctypes.c_int32),
        ("anti_aliasing", ctypes.c_float), ("sky_light", ctypes.c_float * 4), ("phong_tessellation", ctypes.c_float),
        ("traversal", ctypes.c_int32),
    ]


class OrcScene(ctypes.Structure):
    _fields_ = [
        ("bvh", ctypes.c_void_p), ("facesV", ctypes.c_void_p), ("vertices", ctypes.c_void_p),
        ("materials", ctypes.c_void_p), ("lights", ctypes.c_void_p),
        ("num_faces", ctypes.c_uint32), ("num_vertices", ctypes.c_uint32), ("num_materials", ctypes.c_uint32),
        ("facesN", ctypes.c_void_p), ("normals", ctypes.c_void_p), ("num_normals", ctypes.c_uint32),
        ("walk_links", ctypes.c_void_p), ("walk_first", ctypes.c_void_p),
    ]


class OrcCounters(ctypes.Structure):
    _fields_ = [("nodes", ctypes.c_uint64), ("tris", ctypes.c_uint64), ("hits", ctypes.c_uint64), ("paths", ctypes.c_uint64)]


_fp = ctypes.POINTER(ctypes.c_float)
_lib = None
_native = None


def native_lib():
    """The -O3 -march=native build (build_native), for timing; None where it cannot be built (no gcc)."""
    global _native
    if _native is None:
        try:
            _native = ctypes.CDLL(build_native())
        except (OSError, subprocess.CalledProcessError):
            return None
        _native.orc_render_frame.argtypes = lib().orc_render_frame.argtypes
        _native.orc_render_frame.restype = None
    return _native


def lib():
    global _lib
    if _lib is None:
        build()
        try:
            _lib = ctypes.CDLL(_LIB)
        except OSError:
            build(force=True)
            _lib = ctypes.CDLL(_LIB)
        _lib.orc_render_frame.argtypes = [
            ctypes.POINTER(OrcScene), ctypes.POINTER(OrcConfig), ctypes.c_void_p,
            ctypes.c_float, ctypes.c_float, ctypes.c_float, _fp, _fp, _fp,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(OrcCounters)]
        _lib.orc_render_frame.restype = None
        _lib.orc_trace_rays.argtypes = [
            ctypes.POINTER(OrcScene), ctypes.POINTER(OrcConfig), _fp, ctypes.c_int,
            _fp, ctypes.POINTER(ctypes.c_int32), _fp, ctypes.POINTER(ctypes.c_uint32)]
        _lib.orc_trace_rays.restype = None
        _lib.orc_math.argtypes = [ctypes.c_int, _fp, _fp, ctypes.c_int, _fp]
        _lib.orc_math.restype = None
        _lib.orc_brdf_eval.argtypes = [ctypes.c_int, ctypes.c_void_p, _fp, ctypes.c_int, _fp]
        _lib.orc_brdf_eval.restype = None
        _lib.orc_new_ray.argtypes = [ctypes.c_int, ctypes.c_void_p, _fp, ctypes.c_int, _fp]
        _lib.orc_new_ray.restype = None
        _lib.orc_walk_order_count.argtypes = [ctypes.c_int]
        _lib.orc_walk_order_count.restype = ctypes.c_int
        _lib.orc_build_walk_orders.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib.orc_build_walk_orders.restype = ctypes.c_int
        _lib.orc_debug_set_walk_max.argtypes = [ctypes.c_void_p]
        _lib.orc_debug_set_walk_max.restype = None
    return _lib


def _ptr(a):
    return a.ctypes.data_as(_fp)


MATH_OPS = {"sin": 0, "cos": 1, "tan": 2, "acos": 3, "atan": 4, "pow": 5, "randhash": 6}


def math(op, x, y=None):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(x if y is None else y, np.float32)
    out = np.empty_like(x)
    lib().orc_math(MATH_OPS[op], _ptr(x), _ptr(y), x.size, _ptr(out))
    return out


def scene_and_config(desc, cfg):
    """(OrcScene, OrcConfig) from a pbr_scene_desc and a pbr_config (any objects with those fields)."""
    s = OrcScene()
    s.bvh, s.facesV, s.vertices, s.materials, s.lights = desc.bvh, desc.facesV, desc.vertices, desc.materials, desc.lights
    s.num_faces, s.num_vertices, s.num_materials = desc.num_faces, desc.num_vertices, desc.num_materials
    s.facesN, s.normals, s.num_normals = desc.facesN, desc.normals, desc.num_normals
    c = OrcConfig()
    c.width, c.height, c.brdf = cfg.width, cfg.height, cfg.brdf
    c.shadow_rays, c.max_depth, c.max_added_depth, c.samples = cfg.shadow_rays, cfg.max_depth, cfg.max_added_depth, cfg.samples
    c.num_nodes, c.num_lights = desc.num_nodes, desc.num_lights
    c.anti_aliasing = cfg.anti_aliasing
    c.phong_tessellation = getattr(cfg, "phong_tessellation", 0.0)
    for k in range(4):
        c.sky_light[k] = cfg.sky_light[k]
    # pbr_config.traversal -> the oracle's walk.  The product's 3 (PBR_WALK_EIGHT_ORDERS_COMPACT) is the eight-order walk
    # in another memory layout: the same visits, so the oracle's scheme 2 checks it.  STACK_AID selects the oracle-only
    # analysis walk (pt_oracle.c: cfg.traversal == 3), which is no mode of the product.
    c.traversal = {0: 0, 1: 1, 2: 2, 3: 2, STACK_AID: 3}[int(getattr(cfg, "traversal", 0))]
    if c.traversal:
        links, first = walk_orders(desc, 1 if c.traversal == 3 else c.traversal)     # 3: the stack-based analysis aid, over scheme 1's tables
        s.walk_links, s.walk_first = links.ctypes.data, first.ctypes.data
        s._walk = (links, first)          # the scene struct keeps the tables alive
    return s, c


STACK_AID = 99      # cfg.traversal value (oracle only): the stack-based nearest-child-first walk, an analysis aid


def walk_orders(desc, scheme):
    """The successor tables of the product's opt-in ray-ordered walk (pt_oracle.c, "Ray-ordered walk"): (links, first),
    links[k, node] = (hit, miss).  Not a reference algorithm; see the C source."""
    k = lib().orc_walk_order_count(int(scheme))
    if k <= 0:
        raise ValueError("unknown traversal scheme %r" % (scheme,))
    n = int(desc.num_nodes)
    links = np.full((k, n, 2), -1, np.int32)
    first = np.full(k, -1, np.int32)
    bvh = desc.bvh if isinstance(desc.bvh, int) else ctypes.cast(desc.bvh, ctypes.c_void_p).value
    if lib().orc_build_walk_orders(bvh, n, int(scheme), links.ctypes.data, first.ctypes.data) != 0:
        raise ValueError("orc_build_walk_orders failed")
    return links, first


class Renderer:
    """Frame-by-frame driver with the reference's host ping-pong (PathTracer.cpp:59-71)."""

    def __init__(self, desc, cfg, threads=1, native=False):
        self.scene, self.cfg = scene_and_config(desc, cfg)
        self._keep = (desc, cfg)
        self.threads = threads
        self._lib = (native_lib() if native else None) or lib()
        self.native = self._lib is not lib()
        self.width, self.height = int(cfg.width), int(cfg.height)
        self.image = np.zeros((self.height, self.width, 4), np.float32)
        self.debug = np.zeros_like(self.image)
        self.counters = OrcCounters()

    def render_frame(self, seed, pixel_weight, px_dim, cam, rows=None):
        out = np.zeros_like(self.image)
        y0, y1 = rows if rows is not None else (0, self.height)
        self._lib.orc_render_frame(
            ctypes.byref(self.scene), ctypes.byref(self.cfg), ctypes.addressof(cam),
            seed, pixel_weight, px_dim, _ptr(self.image), _ptr(out), _ptr(self.debug),
            y0, y1, self.threads, ctypes.byref(self.counters))
        return out

    def render(self, first_sample_count, seeds, px_dim, cam):
        """== pbr_render: accumulate len(seeds) frames starting at sample count n."""
        for k, seed in enumerate(np.asarray(seeds, np.float32)):
            n = first_sample_count + k
            weight = np.float32(n) / np.float32(n + 1)
            self.image = self.render_frame(float(seed), float(weight), px_dim, cam)
        return self.image

    def counter_dict(self):
        c = self.counters
        return {"nodes": int(c.nodes), "tris": int(c.tris), "hits": int(c.hits), "paths": int(c.paths)}


def trace_rays(desc, cfg, rays):
    s, c = scene_and_config(desc, cfg)
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    n = rays.shape[0]
    t = np.empty(n, np.float32)
    face = np.empty(n, np.int32)
    normal = np.empty((n, 3), np.float32)
    counts = np.empty((n, 2), np.uint32)
    lib().orc_trace_rays(
        ctypes.byref(s), ctypes.byref(c), _ptr(rays), n, _ptr(t),
        face.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _ptr(normal),
        counts.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    return t, face, normal, counts
