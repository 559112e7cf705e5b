"""ctypes face of host/libpbrmulti.so (include/pbr_multi.h): N contexts of the HIP core in one process, one host thread per
device, one RCCL all-gather per render.  Harness glue for the tests; the product is the C ABI.  Loaded on demand — only a
multi-GPU caller pays for loading RCCL."""
import ctypes

import numpy as np

from . import build as _build

RCCL, PEER_COPY = 0, 1
_lib = None


def lib():
    global _lib
    if _lib is None:
        from . import Camera, Config, SceneDesc
        _build.build_multi()
        l = ctypes.CDLL(_build.MULTI_LIB, mode=ctypes.RTLD_GLOBAL)
        vp, fp, ip = ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
        dp = ctypes.POINTER(ctypes.c_double)
        l.pbr_multi_create.argtypes = [ip, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
        l.pbr_multi_destroy.argtypes = [vp]
        l.pbr_multi_destroy.restype = None
        l.pbr_multi_last_error.argtypes = [vp]
        l.pbr_multi_last_error.restype = ctypes.c_char_p
        l.pbr_multi_size.argtypes = [vp]
        l.pbr_multi_context.argtypes = [vp, ctypes.c_int]
        l.pbr_multi_context.restype = vp
        l.pbr_multi_upload_scene.argtypes = [vp, ctypes.POINTER(SceneDesc)]
        l.pbr_multi_configure.argtypes = [vp, ctypes.POINTER(Config)]
        l.pbr_multi_reset_accum.argtypes = [vp]
        l.pbr_multi_tune.argtypes = [vp, ctypes.c_uint32, ctypes.c_float, ctypes.POINTER(Camera), ip, ip]
        l.pbr_multi_render.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, fp, ctypes.c_float, ctypes.POINTER(Camera), ctypes.c_int]
        l.pbr_multi_render_frame.argtypes = [vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.POINTER(Camera), ctypes.c_int, ctypes.c_int]
        l.pbr_multi_gather.argtypes = [vp]
        l.pbr_multi_read_full.argtypes = [vp, ctypes.c_int, fp]
        l.pbr_multi_timings.argtypes = [vp, dp, dp]
        _lib = l
    return _lib


class MultiDevice:
    """pbr_multi: `devices` = HIP ordinals, one context and one host thread each."""

    def __init__(self, devices, transport=RCCL):
        from . import PbrError
        self._err = PbrError
        self._m = ctypes.c_void_p()
        arr = (ctypes.c_int * len(devices))(*devices)
        status = lib().pbr_multi_create(arr, len(devices), transport, ctypes.byref(self._m))
        if status != 0:
            self._m = None
            raise PbrError("%d: %s" % (status, lib().pbr_multi_last_error(None).decode()))
        self.size = len(devices)
        self.width = self.height = 0

    def close(self):
        if getattr(self, "_m", None):
            lib().pbr_multi_destroy(self._m)
            self._m = None

    def __del__(self):
        self.close()

    def _check(self, status):
        if status != 0:
            raise self._err("%d: %s" % (status, lib().pbr_multi_last_error(self._m).decode()))

    def context(self, rank):
        """A Device-like view of rank's pbr_ctx (borrowed: do not close)."""
        from . import Device
        dev = Device.__new__(Device)
        dev._ctx = ctypes.c_void_p(lib().pbr_multi_context(self._m, rank))
        dev.width, dev.height = self.width, self.height
        dev.close = lambda: None
        return dev

    def upload_scene(self, desc):
        self._check(lib().pbr_multi_upload_scene(self._m, ctypes.byref(desc)))

    def configure(self, cfg):
        self._check(lib().pbr_multi_configure(self._m, ctypes.byref(cfg)))
        self.width, self.height = int(cfg.width), int(cfg.height)

    def reset_accum(self):
        self._check(lib().pbr_multi_reset_accum(self._m))

    def tune(self, frames_per_call, px_dim, cam):
        plan, votes = ctypes.c_int(-1), (ctypes.c_int * self.size)()
        self._check(lib().pbr_multi_tune(self._m, frames_per_call, px_dim, ctypes.byref(cam), ctypes.byref(plan), votes))
        return int(plan.value), [int(v) for v in votes]

    def render(self, first_sample_count, seeds, px_dim, cam, gather=True):
        seeds = np.ascontiguousarray(seeds, np.float32)
        self._check(lib().pbr_multi_render(self._m, first_sample_count, len(seeds), seeds.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                           px_dim, ctypes.byref(cam), 1 if gather else 0))

    def render_frame(self, seed, pixel_weight, px_dim, cam, accumulate=True, gather=True):
        self._check(lib().pbr_multi_render_frame(self._m, seed, pixel_weight, px_dim, ctypes.byref(cam), 1 if accumulate else 0, 1 if gather else 0))

    def gather(self):
        self._check(lib().pbr_multi_gather(self._m))

    def read_full(self, rank=0):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(lib().pbr_multi_read_full(self._m, rank, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        return out

    def timings(self):
        r, g = (ctypes.c_double * self.size)(), (ctypes.c_double * self.size)()
        self._check(lib().pbr_multi_timings(self._m, r, g))
        return [float(v) for v in r], [float(v) for v in g]
