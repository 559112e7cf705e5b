"""Known-traffic kernels for interpreting the memory PMC counters (run under rocprofv3 --pmc)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import pbr_loader
pbr = pbr_loader.load()
hip = pbr.hip
hip.pbr_diag_calibrate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_double)]
dev = pbr.Device(0)
GiB = 1 << 30
for mode, name, reads, useful in ((0, "coalesced stream 16 B/lane", 2 * GiB // 16, 2 * GiB), (1, "random 16-B elements", 64 << 20, (64 << 20) * 16), (2, "random 32-B records", 64 << 20, (64 << 20) * 32)):
    ms = ctypes.c_double()
    assert hip.pbr_diag_calibrate(dev._ctx, mode, 2 * GiB, reads, ctypes.byref(ms)) == 0, hip.pbr_last_error(dev._ctx)
    print("mode %d %-28s reads %11d useful bytes %12d  %.3f ms  %.1f GB/s useful" % (mode, name, reads, useful, ms.value, useful / ms.value / 1e6), flush=True)
