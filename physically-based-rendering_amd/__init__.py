"""MI355X path-tracing core behind the sebadorn/Physically-based-Rendering dispatch surface.

Python here is harness glue only (ctypes): the product is

  csrc/libpbrhip.so    hand-written HIP kernels for gfx950 + the C ABI of include/pbr_hip.h
  host/libpbrhost.so   the C++ host side (loaders, BVH builder, buffer packing, PathTracer)

The directory name is not an importable identifier; load it with `pbr_loader.load()` (repo
root), which registers it as module `pbr_amd`.  There is no CPU fallback: if libpbrhip.so is
missing and cannot be built, importing this package raises.
"""
import ctypes
import os

import numpy as np

from . import build as _build
from . import tiles  # noqa: F401  (layout helpers for the multi-GPU path)

_HERE = os.path.dirname(os.path.abspath(__file__))

PBR_OK = 0


class Float4(ctypes.Structure):
    _fields_ = [("x", ctypes.c_float), ("y", ctypes.c_float), ("z", ctypes.c_float), ("w", ctypes.c_float)]


class Camera(ctypes.Structure):
    """pbr_camera / camera_cl (source/PathTracer.h:25-32), 80 bytes."""
    _fields_ = [("eye", Float4), ("w", Float4), ("u", Float4), ("v", Float4),
                ("focusPoint", ctypes.c_int32 * 2), ("lense", ctypes.c_float * 2)]


class DenoiseParams(ctypes.Structure):
    """pbr_denoise_params.  Defaults: what scripts/denoise_demo.py found best around 4 spp (sigma_color: 4.0 at 1 spp,
    0.6 at 16, 0.3 at 64 — the noise it has to bridge shrinks with the sample count)."""
    _fields_ = [("passes", ctypes.c_uint32), ("sigma_color", ctypes.c_float), ("sigma_normal", ctypes.c_float),
                ("sigma_world", ctypes.c_float), ("sigma_albedo", ctypes.c_float)]

    def __init__(self, passes=5, sigma_color=1.2, sigma_normal=0.25, sigma_world=3.0, sigma_albedo=0.1):
        super().__init__(passes, sigma_color, sigma_normal, sigma_world, sigma_albedo)


class SceneDesc(ctypes.Structure):
    """pbr_scene_desc"""
    _fields_ = [
        ("bvh", ctypes.c_void_p), ("num_nodes", ctypes.c_uint32),
        ("facesV", ctypes.c_void_p), ("facesN", ctypes.c_void_p), ("num_faces", ctypes.c_uint32),
        ("vertices", ctypes.c_void_p), ("num_vertices", ctypes.c_uint32),
        ("normals", ctypes.c_void_p), ("num_normals", ctypes.c_uint32),
        ("materials", ctypes.c_void_p), ("num_materials", ctypes.c_uint32),
        ("brdf", ctypes.c_uint32),
        ("lights", ctypes.c_void_p), ("num_lights", ctypes.c_uint32),
    ]


class Config(ctypes.Structure):
    """pbr_config"""
    _fields_ = [
        ("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("brdf", ctypes.c_uint32),
        ("shadow_rays", ctypes.c_uint32), ("max_depth", ctypes.c_uint32), ("max_added_depth", ctypes.c_uint32),
        ("samples", ctypes.c_uint32), ("anti_aliasing", ctypes.c_float), ("phong_tessellation", ctypes.c_float),
        ("sky_light", ctypes.c_float * 4), ("tile_world", ctypes.c_uint32), ("tile_rank", ctypes.c_uint32),
        ("traversal", ctypes.c_uint32), ("arith", ctypes.c_uint32),
    ]


class Counters(ctypes.Structure):
    _fields_ = [("nodes", ctypes.c_uint64), ("tris", ctypes.c_uint64), ("hits", ctypes.c_uint64), ("paths", ctypes.c_uint64)]


def _load(path, builder):
    builder()      # a no-op unless the library is missing or older than its sources (build.py _stale)
    try:
        return ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except OSError:
        builder(force=True)
        return ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


try:
    _lab = os.environ.get("PBR_LAB_ENV") == "1"   # lab runs only: no PBR_* variable steers a product process
    if _lab and os.environ.get("PBR_HIP_LIB"):          # an alternative build of the same source (scripts/lab.sh)
        hip = ctypes.CDLL(os.environ["PBR_HIP_LIB"], mode=ctypes.RTLD_GLOBAL)
    elif _lab and os.environ.get("PBR_GUARD") == "1":   # every device loop bounded (build.py)
        hip = _load(_build.HIP_GUARD_LIB, lambda force=False: _build.build_hip(force, guard=True))
    else:
        hip = _load(_build.HIP_LIB, _build.build_hip)
    host = _load(_build.HOST_LIB, _build.build_host)
except Exception as exc:  # no fallback by design
    raise ImportError("the HIP core (csrc/libpbrhip.so) / host library could not be loaded or built: %s" % exc)

_fp = ctypes.POINTER(ctypes.c_float)
_vp = ctypes.c_void_p

# the struct mirrors below are written against this version of include/pbr_hip.h (pbr_config: 68 bytes since version 5).
# A library of another ABI version must not be handed them; lab runs that load an older build on purpose (PBR_HIP_LIB) say so.
ABI_VERSION = 6
if hasattr(hip, "pbr_abi_version"):
    hip.pbr_abi_version.restype = ctypes.c_uint32
    if hip.pbr_abi_version() != ABI_VERSION and not _lab:
        raise ImportError("libpbrhip.so speaks ABI version %d, this harness %d (include/pbr_hip.h, PBR_ABI_VERSION)" % (hip.pbr_abi_version(), ABI_VERSION))
elif not _lab:
    raise ImportError("libpbrhip.so has no pbr_abi_version(): a build older than round 6")

hip.pbr_create.argtypes = [ctypes.c_int, ctypes.POINTER(_vp)]
hip.pbr_destroy.argtypes = [_vp]
hip.pbr_destroy.restype = None
hip.pbr_last_error.argtypes = [_vp]
hip.pbr_last_error.restype = ctypes.c_char_p
hip.pbr_upload_scene.argtypes = [_vp, ctypes.POINTER(SceneDesc)]
hip.pbr_configure.argtypes = [_vp, ctypes.POINTER(Config)]
hip.pbr_validate_scene.argtypes = [ctypes.POINTER(SceneDesc), ctypes.c_char_p, ctypes.c_size_t]
hip.pbr_write_input.argtypes = [_vp, _fp]
hip.pbr_reset_accum.argtypes = [_vp]
hip.pbr_render_frame.argtypes = [_vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.POINTER(Camera)]
hip.pbr_accumulate.argtypes = [_vp]
hip.pbr_render.argtypes = [_vp, ctypes.c_uint32, ctypes.c_uint32, _fp, ctypes.c_float, ctypes.POINTER(Camera)]
hip.pbr_read_output.argtypes = [_vp, _fp]
hip.pbr_read_debug.argtypes = [_vp, _fp]
hip.pbr_read_display.argtypes = [_vp, ctypes.c_void_p, ctypes.c_int]
hip.pbr_denoise.argtypes = [_vp, ctypes.c_float, ctypes.POINTER(Camera), ctypes.POINTER(DenoiseParams), _fp, _fp]
hip.pbr_bvh_node_capacity.argtypes = [ctypes.c_uint32]
hip.pbr_bvh_node_capacity.restype = ctypes.c_uint32
hip.pbr_build_bvh.argtypes = [_vp, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p, ctypes.c_void_p]
hip.pbr_get_focus_depth.argtypes = [_vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
hip.pbr_set_focus_depth.argtypes = [_vp, ctypes.c_float]
hip.pbr_read_full.argtypes = [_vp, _fp]
hip.pbr_get_counters.argtypes = [_vp, ctypes.POINTER(Counters)]
hip.pbr_last_kernel_ms.argtypes = [_vp]
hip.pbr_last_kernel_ms.restype = ctypes.c_double
hip.pbr_tile_bytes.argtypes = [_vp]
hip.pbr_tile_bytes.restype = ctypes.c_uint64
hip.pbr_export_tiles.argtypes = [_vp, _vp]
hip.pbr_import_tiles.argtypes = [_vp, _vp]
_ip = ctypes.POINTER(ctypes.c_int32)
_up = ctypes.POINTER(ctypes.c_uint32)
hip.pbr_diag_math.argtypes = [_vp, ctypes.c_int, _fp, _fp, ctypes.c_int, _fp]
hip.pbr_diag_trace.argtypes = [_vp, _fp, ctypes.c_int, _fp, _ip, _fp, _up]
hip.pbr_diag_brdf.argtypes = [_vp, _fp, ctypes.c_int, _fp]
hip.pbr_diag_new_ray.argtypes = [_vp, _fp, ctypes.c_int, _fp]
hip.pbr_diag_guard_trips.argtypes = [_vp, _up]
hip.pbr_diag_last_trace.argtypes = [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint32)]
hip.pbr_diag_last_plan.argtypes = [_vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
hip.pbr_diag_pin_plan.argtypes = [_vp, ctypes.c_int]
# entry points that older builds of the library lack are bound only when they are there: lab runs load other builds of the
# library for A/B comparisons (PBR_HIP_LIB), and a missing symbol must fail where it is called, not at import
for _name, _args in (
        ("pbr_diag_scene_bytes", [_vp, ctypes.POINTER(ctypes.c_uint64)]),
        ("pbr_mode_built", [ctypes.c_uint32, ctypes.c_uint32]),
        ("pbr_diag_last_kernel", [_vp, ctypes.c_char_p, ctypes.c_size_t]),
        ("pbr_diag_launch_fit", [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
        ("pbr_diag_set_knob", [_vp, ctypes.c_char_p, ctypes.c_int]),                 # absent from round 2's library
        ("pbr_diag_get_tile_order", [_vp, ctypes.c_int, _up, ctypes.c_uint32, _up, _up]),          # round 6
        ("pbr_diag_set_tile_order", [_vp, _up, ctypes.c_uint32, _up]),
        ("pbr_diag_last_deal", [_vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]),
        ("pbr_diag_bvh_build_info", [_vp, ctypes.POINTER(ctypes.c_int)])):
    if hasattr(hip, _name):
        getattr(hip, _name).argtypes = _args
hip.pbr_diag_tune_budget.argtypes = [_vp, ctypes.POINTER(ctypes.c_uint32)]

host.pbrh_last_error.restype = ctypes.c_char_p
host.pbrh_cfg_set.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
host.pbrh_cfg_get.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
host.pbrh_cfg_load.argtypes = [ctypes.c_char_p]
host.pbrh_scene_load_obj.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
host.pbrh_scene_load_obj.restype = _vp
host.pbrh_scene_generate.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32]
host.pbrh_scene_generate.restype = _vp
host.pbrh_scene_destroy.argtypes = [_vp]
host.pbrh_scene_destroy.restype = None
host.pbrh_scene_desc.argtypes = [_vp, ctypes.POINTER(SceneDesc)]
host.pbrh_scene_info.argtypes = [_vp, ctypes.POINTER(ctypes.c_uint32)]
host.pbrh_scene_config.argtypes = [_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(Config)]
host.pbrh_scene_camera.argtypes = [_vp, ctypes.POINTER(Camera)]
host.pbrh_camera_lookat.argtypes = [_fp, _fp, ctypes.POINTER(Camera)]
host.pbrh_pixel_dimension.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_float]
host.pbrh_pixel_dimension.restype = ctypes.c_float
host.pbrh_pt_create.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32]
host.pbrh_pt_create.restype = _vp
host.pbrh_pt_destroy.argtypes = [_vp]
host.pbrh_pt_destroy.restype = None
host.pbrh_pt_init.argtypes = [_vp, _vp, ctypes.c_uint32, ctypes.c_uint32]
host.pbrh_pt_generate_image.argtypes = [_vp, _fp, _fp]
host.pbrh_pt_generate_images.argtypes = [_vp, ctypes.c_uint32, _fp]
host.pbrh_write_ppm.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
host.pbrh_write_pfm.argtypes = [ctypes.c_char_p, _fp, ctypes.c_uint32, ctypes.c_uint32]
host.pbrh_write_png.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
host.pbrh_write_exr.argtypes = [ctypes.c_char_p, _fp, ctypes.c_uint32, ctypes.c_uint32]
host.pbrh_cl_adaptor_render.argtypes = [_vp, ctypes.c_uint32, ctypes.c_float, _fp, _fp]
host.pbrh_cl_adaptor_render_ex.argtypes = [_vp, ctypes.c_uint32, ctypes.c_float, _fp, _fp, ctypes.c_uint32]
host.pbrh_pt_set_focus.argtypes = [_vp, ctypes.c_int, ctypes.c_int]
host.pbrh_pt_reset_sample_count.argtypes = [_vp]
host.pbrh_pt_sample_count.argtypes = [_vp]
host.pbrh_pt_sample_count.restype = ctypes.c_uint32
host.pbrh_pt_context.argtypes = [_vp]
host.pbrh_pt_context.restype = _vp
host.pbrh_pt_camera.argtypes = [_vp, ctypes.POINTER(Camera)]


class PbrError(RuntimeError):
    pass


def _as_fp(a):
    return a.ctypes.data_as(_fp)


# ----------------------------------------------------------------------------------------------
# Cfg (source/Cfg.h) — keys as in the reference's config.json
# ----------------------------------------------------------------------------------------------

def cfg_reset():
    host.pbrh_cfg_reset()


def cfg_set(**kv):
    """cfg_set(**{"render.max_depth": 4})"""
    for k, v in kv.items():
        if isinstance(v, bool):
            v = "true" if v else "false"
        host.pbrh_cfg_set(k.encode(), str(v).encode())


def cfg_get(key):
    buf = ctypes.create_string_buffer(256)
    host.pbrh_cfg_get(key.encode(), buf, 256)
    return buf.value.decode()


def validate_scene(desc):
    """pbr_validate_scene: the upload's checks without a device.  Returns "" or the reason the scene is rejected."""
    buf = ctypes.create_string_buffer(512)
    status = hip.pbr_validate_scene(ctypes.byref(desc), buf, 512)
    return "" if status == PBR_OK else (buf.value.decode() or "invalid scene")


def pixel_dimension(width, height, fov=45.0):
    return float(host.pbrh_pixel_dimension(width, height, fov))


# ----------------------------------------------------------------------------------------------
# Host scene: loader / generator + BVH + packed buffers (GLWidget::loadModel without GL)
# ----------------------------------------------------------------------------------------------

class HostScene:
    def __init__(self, handle):
        self._h = handle
        if not handle:
            raise PbrError(host.pbrh_last_error().decode())
        self.desc = SceneDesc()
        host.pbrh_scene_desc(self._h, ctypes.byref(self.desc))
        info = (ctypes.c_uint32 * 10)()
        host.pbrh_scene_info(self._h, info)
        keys = ("flat_nodes", "faces", "vertices", "materials", "lights", "tree_nodes", "leaves", "depth", "skipped", "objects")
        self.info = dict(zip(keys, [int(v) for v in info]))

    @classmethod
    def load_obj(cls, directory, filename):
        if not directory.endswith("/"):
            directory += "/"
        return cls(host.pbrh_scene_load_obj(directory.encode(), filename.encode()))

    @classmethod
    def generate(cls, kind, seed=1, triangles=0):
        return cls(host.pbrh_scene_generate(kind.encode(), seed, triangles))

    def close(self):
        if getattr(self, "_h", None):
            host.pbrh_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def render_through_cl_adaptor(self, frames, seed_step=0.0333, refeed_every=0):
        """Drive the `CL` look-alike (host/cl_adaptor.h) the way the reference's PathTracer drives CL, at the
        configured window size (cfg window.width / window.height); returns (image, debug), row 0 = bottom."""
        w, h = int(cfg_get("window.width")), int(cfg_get("window.height"))
        image = np.empty((h, w, 4), np.float32)
        debug = np.empty((h, w, 4), np.float32)
        if host.pbrh_cl_adaptor_render_ex(self._h, frames, seed_step, image.ctypes.data_as(_fp), debug.ctypes.data_as(_fp), refeed_every) != 0:
            raise PbrError(host.pbrh_last_error().decode())
        return image, debug

    def config(self, width, height):
        cfg = Config()
        host.pbrh_scene_config(self._h, width, height, ctypes.byref(cfg))
        return cfg

    def camera(self):
        cam = Camera()
        host.pbrh_scene_camera(self._h, ctypes.byref(cam))
        return cam

    def _array(self, ptr, count, dtype, width):
        if not ptr or count == 0:
            return np.zeros((0, width), dtype)
        n = count * width
        buf = (ctypes.c_uint32 * n).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(count, width).copy()

    def arrays(self):
        """Copies of the flat wire-format arrays (for tests)."""
        d = self.desc
        mat_floats = 12 if d.brdf == 0 else 16
        return {
            "bvh": self._array(d.bvh, d.num_nodes, np.float32, 8),
            "facesV": self._array(d.facesV, d.num_faces, np.uint32, 4),
            "facesN": self._array(d.facesN, d.num_faces, np.uint32, 4),
            "normals": self._array(d.normals, d.num_normals, np.float32, 4),
            "vertices": self._array(d.vertices, d.num_vertices, np.float32, 4),
            "materials": self._array(d.materials, d.num_materials, np.float32, mat_floats),
            "lights": self._array(d.lights, max(1, d.num_lights), np.float32, 12),
        }


# ----------------------------------------------------------------------------------------------
# Device context: the C ABI of include/pbr_hip.h
# ----------------------------------------------------------------------------------------------

# Lab scripts and A/B runs steer the library through environment variables; the LIBRARY reads none (include/pbr_hip_diag.h,
# pbr_diag_set_knob) — this harness maps them onto knobs when a context is created, and ONLY when PBR_LAB_ENV=1 says that
# the process is a lab run (round 4: a stray PBR_* variable in a user's environment must not change the schedule of a
# product render; scripts/*.py and scripts/*.sh set PBR_LAB_ENV themselves).
_ENV_KNOBS = {
    "PBR_LDS_SLOTS": "lds_slots", "PBR_BLOCKS_PER_CU": "blocks_per_cu", "PBR_PH_PARK": "ph_park", "PBR_PH_SHADE": "ph_shade",
    "PBR_PARK_EIGHTHS": "park_eighths", "PBR_DRAIN_MODE": "drain_mode", "PBR_REFILL_BATCH": "refill_batch",
    "PBR_CHUNK_FRAMES": "chunk_frames", "PBR_FACE_NORMALS": "face_normals", "PBR_PLOC_RADIUS": "ploc_radius", "PBR_TUNE_LOG": "tune_log",
    "PBR_DEAL_ORDER": "deal_order",
}


class Device:
    def __init__(self, device=0):
        self._ctx = _vp()
        status = hip.pbr_create(device, ctypes.byref(self._ctx))
        if status != PBR_OK:
            msg = hip.pbr_last_error(self._ctx).decode() if self._ctx else "pbr_create failed"
            if self._ctx:
                hip.pbr_destroy(self._ctx)
                self._ctx = None
            raise PbrError(msg)
        self.width = self.height = 0
        if os.environ.get("PBR_LAB_ENV") == "1":
            self._apply_lab_environment()

    def _apply_lab_environment(self):
        """PBR_* variables -> knobs / pinned plan, for lab scripts (opt-in: PBR_LAB_ENV=1)."""
        if not hasattr(hip, "pbr_diag_set_knob"):
            raise PbrError("PBR_LAB_ENV=1, but this libpbrhip has no pbr_diag_set_knob")
        for var, knob in _ENV_KNOBS.items():
            if os.environ.get(var):
                self.set_knob(knob, int(os.environ[var]))
        builder = os.environ.get("PBR_BVH_BUILDER")
        if builder:
            if builder not in ("ploc", "lbvh"):
                raise PbrError("PBR_BVH_BUILDER must be 'ploc' or 'lbvh', not %r" % builder)
            self.set_knob("bvh_builder", {"ploc": 0, "lbvh": 1}[builder])
        if os.environ.get("PBR_PLAN"):
            self.pin_plan(int(os.environ["PBR_PLAN"]))

    def set_knob(self, name, value):
        """pbr_diag_set_knob: experiment / test knobs of this context (include/pbr_hip_diag.h); -1 = the default."""
        self._check(hip.pbr_diag_set_knob(self._ctx, name.encode(), int(value)))

    def close(self):
        if getattr(self, "_ctx", None):
            hip.pbr_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        self.close()

    def _check(self, status):
        if status != PBR_OK:
            raise PbrError("%d: %s" % (status, hip.pbr_last_error(self._ctx).decode()))

    def upload_scene(self, desc):
        self._check(hip.pbr_upload_scene(self._ctx, ctypes.byref(desc)))

    def configure(self, cfg):
        self._check(hip.pbr_configure(self._ctx, ctypes.byref(cfg)))
        self.width, self.height = int(cfg.width), int(cfg.height)

    def write_input(self, rgba):
        rgba = np.ascontiguousarray(rgba, np.float32)
        assert rgba.size == self.width * self.height * 4
        self._check(hip.pbr_write_input(self._ctx, _as_fp(rgba)))

    def reset_accum(self):
        self._check(hip.pbr_reset_accum(self._ctx))

    def render_frame(self, seed, pixel_weight, px_dim, cam):
        self._check(hip.pbr_render_frame(self._ctx, seed, pixel_weight, px_dim, ctypes.byref(cam)))

    def accumulate(self):
        self._check(hip.pbr_accumulate(self._ctx))

    def render(self, first_sample_count, seeds, px_dim, cam):
        seeds = np.ascontiguousarray(seeds, np.float32)
        self._check(hip.pbr_render(self._ctx, first_sample_count, len(seeds), _as_fp(seeds), px_dim, ctypes.byref(cam)))

    def _read(self, fn):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(fn(self._ctx, _as_fp(out)))
        return out

    def read_output(self):
        return self._read(hip.pbr_read_output)

    def read_debug(self):
        return self._read(hip.pbr_read_debug)

    def read_full(self):
        return self._read(hip.pbr_read_full)

    def build_bvh(self, vertices, facesV, facesN):
        """pbr_build_bvh: BVH built on the device (Morton order + clustering by surface area).  vertices (n, 4) float32, facesV / facesN (m, 4) uint32 ->
        (nodes (k, 8) float32 in the reference's flat format, facesV and facesN in leaf order)."""
        vertices = np.ascontiguousarray(vertices, np.float32).reshape(-1, 4)
        facesV = np.ascontiguousarray(facesV, np.uint32).reshape(-1, 4)
        facesN = np.ascontiguousarray(facesN, np.uint32).reshape(-1, 4)
        m = facesV.shape[0]
        nodes = np.zeros((int(hip.pbr_bvh_node_capacity(m)), 8), np.float32)
        outV, outN = np.empty_like(facesV), np.empty_like(facesN)
        count = ctypes.c_uint32()
        self._check(hip.pbr_build_bvh(self._ctx, vertices.ctypes.data, vertices.shape[0], facesV.ctypes.data, facesN.ctypes.data, m,
                                      nodes.ctypes.data, ctypes.byref(count), outV.ctypes.data, outN.ctypes.data))
        return nodes[:count.value].copy(), outV, outN

    def get_focus_depth(self, x, y):
        """(previous-frame distance of pixel (x, y), whether this rank owns its tile) — DOF with tile sharding."""
        t, owned = ctypes.c_float(), ctypes.c_int()
        self._check(hip.pbr_get_focus_depth(self._ctx, x, y, ctypes.byref(t), ctypes.byref(owned)))
        return float(t.value), bool(owned.value)

    def set_focus_depth(self, t):
        self._check(hip.pbr_set_focus_depth(self._ctx, t))

    def read_display(self, top_row_first=False):
        """imageOut as (H, W, 4) uint8 — what the reference's GL viewer shows (clamped linear colour)."""
        out = np.empty((self.height, self.width, 4), np.uint8)
        self._check(hip.pbr_read_display(self._ctx, out.ctypes.data, 1 if top_row_first else 0))
        return out

    def denoise(self, px_dim, cam, params=None, features=False):
        """pbr_denoise: the accumulated image after the edge-avoiding a-trous filter, (H, W, 4) float32 (the accumulation
        is not modified); with features=True also the first-hit feature buffers (3, H, W, 4): position | t, normal | hit,
        Kd | material."""
        params = params if params is not None else DenoiseParams()
        out = np.empty((self.height, self.width, 4), np.float32)
        feat = np.empty((3, self.height, self.width, 4), np.float32) if features else None
        self._check(hip.pbr_denoise(self._ctx, px_dim, ctypes.byref(cam), ctypes.byref(params), out.ctypes.data_as(_fp),
                                    feat.ctypes.data_as(_fp) if features else None))
        return (out, feat) if features else out

    def counters(self):
        c = Counters()
        self._check(hip.pbr_get_counters(self._ctx, ctypes.byref(c)))
        return {"nodes": int(c.nodes), "tris": int(c.tris), "hits": int(c.hits), "paths": int(c.paths)}

    def last_kernel_ms(self):
        return float(hip.pbr_last_kernel_ms(self._ctx))

    def tile_bytes(self):
        return int(hip.pbr_tile_bytes(self._ctx))

    # ---- diagnostic stages (include/pbr_hip_diag.h) ----

    MATH_OPS = {"sin": 0, "cos": 1, "tan": 2, "acos": 3, "atan": 4, "pow": 5, "randhash": 6}

    def diag_math(self, op, x, y=None):
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(x if y is None else y, np.float32)
        out = np.empty_like(x)
        self._check(hip.pbr_diag_math(self._ctx, self.MATH_OPS[op], _as_fp(x), _as_fp(y), x.size, _as_fp(out)))
        return out

    def diag_trace(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        t = np.empty(n, np.float32)
        face = np.empty(n, np.int32)
        normal = np.empty((n, 3), np.float32)
        counts = np.empty((n, 2), np.uint32)
        self._check(hip.pbr_diag_trace(self._ctx, _as_fp(rays), n, _as_fp(t), face.ctypes.data_as(_ip),
                                       _as_fp(normal), counts.ctypes.data_as(_up)))
        return t, face, normal, counts

    def diag_brdf(self, items):
        items = np.ascontiguousarray(items, np.float32).reshape(-1, 16)
        out = np.empty((items.shape[0], 4), np.float32)
        self._check(hip.pbr_diag_brdf(self._ctx, _as_fp(items), items.shape[0], _as_fp(out)))
        return out

    def diag_new_ray(self, items):
        items = np.ascontiguousarray(items, np.float32).reshape(-1, 12)
        out = np.empty((items.shape[0], 8), np.float32)
        self._check(hip.pbr_diag_new_ray(self._ctx, _as_fp(items), items.shape[0], _as_fp(out)))
        return out

    def guard_trips(self):
        out = (ctypes.c_uint32 * 3)()
        self._check(hip.pbr_diag_guard_trips(self._ctx, out))
        return [int(v) for v in out]

    def last_trace(self):
        """(summed duration in ms, number) of the path-tracing launches of the last render."""
        ms, n = ctypes.c_double(), ctypes.c_uint32()
        self._check(hip.pbr_diag_last_trace(self._ctx, ctypes.byref(ms), ctypes.byref(n)))
        return float(ms.value), int(n.value)

    def last_plan(self):
        """(name of the schedule that rendered the last render, auto-tuner's choice or -1 while measuring)."""
        name, tuned = ctypes.create_string_buffer(48), ctypes.c_int(-1)
        self._check(hip.pbr_diag_last_plan(self._ctx, name, 48, ctypes.byref(tuned)))
        return name.value.decode(), int(tuned.value)

    def scene_bytes(self):
        """Device bytes of the scene: {"nodes": reference-order stream, "walk_streams": the ordered walk's (0 until built), "faces"}."""
        out = (ctypes.c_uint64 * 3)()
        self._check(hip.pbr_diag_scene_bytes(self._ctx, out))
        return {"nodes": int(out[0]), "walk_streams": int(out[1]), "faces": int(out[2])}

    def last_kernel(self):
        """The symbol of the kernel behind last_plan()[0], as a profiler prints it: "ptk_f0::pathTracingDual<1, false, false>"."""
        name = ctypes.create_string_buffer(96)
        self._check(hip.pbr_diag_last_kernel(self._ctx, name, 96))
        return name.value.decode()

    def launch_fit(self):
        """(fixed ms, ms per frame) of a launch of the plan in use, as the schedule tuner fitted them; None without a fit."""
        a, b = ctypes.c_double(), ctypes.c_double()
        if hip.pbr_diag_launch_fit(self._ctx, ctypes.byref(a), ctypes.byref(b)) != PBR_OK:
            return None
        return float(a.value), float(b.value)

    PLAN_NAMES = ("refill-lean", "refill-wide", "phased-lean", "phased-wide", "phased-mid", "refill-mid", "phased-dual")

    def pin_plan(self, plan):
        """Render with schedule `plan` (index into PLAN_NAMES) without tuning; -1 = let the tuner choose."""
        self._check(hip.pbr_diag_pin_plan(self._ctx, int(plan)))

    def tile_order(self, cost_ordered=False, which=None):
        """pbr_diag_get_tile_order: (order, band_first) — the local tiles in the order the queue deals them, band b's
        stretch = order[band_first[b]:band_first[b + 1]]; cost_ordered (which = 1): the library's learnt cost-classes order instead
        of the spatial (or pinned) one; which = 2: its expensive-last order."""
        n, first, which = ctypes.c_uint32(), (ctypes.c_uint32 * 9)(), (which if which is not None else (1 if cost_ordered else 0))
        self._check(hip.pbr_diag_get_tile_order(self._ctx, which, None, 0, ctypes.byref(n), first))
        order = np.empty(n.value, np.uint32)
        self._check(hip.pbr_diag_get_tile_order(self._ctx, which, order.ctypes.data_as(_up), n.value, ctypes.byref(n), first))
        return order, np.array(first[:], np.int64)

    def set_tile_order(self, order, band_first=None):
        """pbr_diag_set_tile_order: deal the tiles in this order until the next pbr_configure — per band a permutation of the
        band's own tiles, or with band_first (9 entries) any partition of the local tiles into eight lists; None hands the order
        back to the library."""
        if order is None:
            self._check(hip.pbr_diag_set_tile_order(self._ctx, None, 0, None))
        else:
            order = np.ascontiguousarray(order, np.uint32)
            first = None if band_first is None else np.ascontiguousarray(band_first, np.uint32)
            self._check(hip.pbr_diag_set_tile_order(self._ctx, order.ctypes.data_as(_up), order.size, None if first is None else first.ctypes.data_as(_up)))

    def bvh_build_radius(self):
        """pbr_diag_bvh_build_info: the clustering radius pbr_build_bvh last used in this context (0: no build yet)."""
        r = ctypes.c_int(0)
        self._check(hip.pbr_diag_bvh_build_info(self._ctx, ctypes.byref(r)))
        return int(r.value)

    def last_deal(self):
        """pbr_diag_last_deal: (the order the last render's largest launch was dealt in, whether a cost order has been learnt)."""
        name, learnt = ctypes.create_string_buffer(16), ctypes.c_int(0)
        self._check(hip.pbr_diag_last_deal(self._ctx, name, 16, ctypes.byref(learnt)))
        return name.value.decode(), bool(learnt.value)

    def tune_budget(self):
        """Frames of the configured size after which the schedule tuner has settled."""
        n = ctypes.c_uint32(0)
        self._check(hip.pbr_diag_tune_budget(self._ctx, ctypes.byref(n)))
        return int(n.value)

    def export_tiles(self, device_ptr):
        self._check(hip.pbr_export_tiles(self._ctx, device_ptr))

    def import_tiles(self, device_ptr):
        self._check(hip.pbr_import_tiles(self._ctx, device_ptr))


def write_ppm(path, rgba8):
    """(H, W, 4) uint8, top row first (Device.read_display(top_row_first=True)) -> binary PPM."""
    rgba8 = np.ascontiguousarray(rgba8, np.uint8)
    if host.pbrh_write_ppm(str(path).encode(), rgba8.ctypes.data, rgba8.shape[1], rgba8.shape[0]) != 0:
        raise PbrError(host.pbrh_last_error().decode())


def write_pfm(path, rgba):
    """(H, W, 4) float32, row 0 = bottom (Device.read_output()) -> PFM, the linear image."""
    rgba = np.ascontiguousarray(rgba, np.float32)
    if host.pbrh_write_pfm(str(path).encode(), _as_fp(rgba), rgba.shape[1], rgba.shape[0]) != 0:
        raise PbrError(host.pbrh_last_error().decode())


def write_png(path, rgba8):
    """(H, W, 4) uint8, top row first (Device.read_display(top_row_first=True)) -> PNG (8-bit RGBA)."""
    rgba8 = np.ascontiguousarray(rgba8, np.uint8)
    if host.pbrh_write_png(str(path).encode(), rgba8.ctypes.data, rgba8.shape[1], rgba8.shape[0]) != 0:
        raise PbrError(host.pbrh_last_error().decode())


def write_exr(path, rgba):
    """(H, W, 4) float32, row 0 = bottom (Device.read_output()) -> OpenEXR: binary32 R, G, B and A = the first-hit distance."""
    rgba = np.ascontiguousarray(rgba, np.float32)
    if host.pbrh_write_exr(str(path).encode(), _as_fp(rgba), rgba.shape[1], rgba.shape[0]) != 0:
        raise PbrError(host.pbrh_last_error().decode())


def frame_seeds(first_sample_count, n_frames, seed_step=0.0333):
    """The fixed seed sequence that stands in for the reference's wall clock
    (PathTracer.cpp:63,78-82): seed_k = step * (k + 1), evaluated in float32."""
    k = np.arange(first_sample_count + 1, first_sample_count + n_frames + 1, dtype=np.float32)
    return (np.float32(seed_step) * k).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# PathTracer driver (host/path_tracer.h), as the reference's GLWidget would use it
# ----------------------------------------------------------------------------------------------

class PathTracer:
    def __init__(self, device, width, height):
        self._h = host.pbrh_pt_create(device, width, height)
        if not self._h:
            raise PbrError(host.pbrh_last_error().decode())
        self.width, self.height = width, height

    def close(self):
        if getattr(self, "_h", None):
            host.pbrh_pt_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def initOpenCLBuffers(self, scene, tile_world=1, tile_rank=0):
        if host.pbrh_pt_init(self._h, scene._h, tile_world, tile_rank) != 0:
            raise PbrError(host.pbrh_last_error().decode())

    def generateImage(self, with_debug=False):
        img = np.empty((self.height, self.width, 4), np.float32)
        dbg = np.empty_like(img) if with_debug else None
        if host.pbrh_pt_generate_image(self._h, _as_fp(img), _as_fp(dbg) if with_debug else None) != 0:
            raise PbrError(host.pbrh_last_error().decode())
        return (img, dbg) if with_debug else img

    def generateImages(self, frames):
        img = np.empty((self.height, self.width, 4), np.float32)
        if host.pbrh_pt_generate_images(self._h, frames, _as_fp(img)) != 0:
            raise PbrError(host.pbrh_last_error().decode())
        return img

    def setFocus(self, x, y):
        host.pbrh_pt_set_focus(self._h, x, y)

    def resetSampleCount(self):
        host.pbrh_pt_reset_sample_count(self._h)

    def sampleCount(self):
        return int(host.pbrh_pt_sample_count(self._h))
