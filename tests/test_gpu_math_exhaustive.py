"""Every float32 there is through the one-argument functions of the arithmetic layer: the HIP
implementations (csrc/pt_math.hpp via pbr_diag_math) against the oracle's, bit for bit, for all 2^32 inputs of sin, cos,
tan, acos, atan and the RNG's hash step; a billion random pairs through pow.  Minutes of CPU on the GPU box's host
cores, so it only runs when asked for: PBR_EXHAUSTIVE=1 python -m pytest tests/test_gpu_math_exhaustive.py -m gpu
(log of the round's run: profiles/r02/math_exhaustive.txt)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("PBR_EXHAUSTIVE") != "1", reason="set PBR_EXHAUSTIVE=1 (takes minutes)")]

CHUNK = 1 << 26          # 256 MB of inputs per device call
PIECE = 1 << 20          # the oracle is single-threaded per call (ctypes releases the GIL): pieces on a thread pool


def oracle_parallel(oracle, pool, op, x, y=None):
    spans = [(a, min(a + PIECE, x.size)) for a in range(0, x.size, PIECE)]
    parts = list(pool.map(lambda s: oracle.math(op, x[s[0]:s[1]], None if y is None else y[s[0]:s[1]]), spans))
    return np.concatenate(parts)


def same_bits(a, b):
    """Bit-identical, except that any NaN equals any NaN (the payload is not part of the contract)."""
    ai, bi = a.view(np.uint32), b.view(np.uint32)
    return (ai == bi) | (np.isnan(a) & np.isnan(b))


@pytest.mark.parametrize("op", ["sin", "cos", "tan", "acos", "atan", "randhash"])
def test_every_float32(pbr, oracle, gpu_device, op):
    dev = pbr.Device(gpu_device)
    threads = min(64, os.cpu_count() or 8)
    with ThreadPoolExecutor(threads) as pool:
        for start in range(0, 1 << 32, CHUNK):
            x = np.arange(start, start + CHUNK, dtype=np.uint32).view(np.float32)
            got = dev.diag_math(op, x)
            want = oracle_parallel(oracle, pool, op, x)
            ok = same_bits(got, want)
            if not ok.all():
                k = int(np.flatnonzero(~ok)[0])
                raise AssertionError("%s( bits 0x%08x = %r ): HIP %r, oracle %r; %d mismatches in this chunk" % (
                    op, start + k, float(x[k]), float(got[k]), float(want[k]), int((~ok).sum())))
    dev.close()


def test_a_billion_pow_pairs(pbr, oracle, gpu_device):
    """pow( x, y ) as the BRDFs use it (pt_brdf.cl: bases in [0, 1] and a little above, exponents from 1e-3 to the
    nu = nv = 100000 of suzanne.mtl, negative ones, and raw bit patterns for the special cases)."""
    dev = pbr.Device(gpu_device)
    rng = np.random.default_rng(12)
    threads = min(64, os.cpu_count() or 8)
    with ThreadPoolExecutor(threads) as pool:
        for rnd in range(16):
            n = CHUNK
            if rnd % 4 == 3:
                x = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
                y = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
            else:
                x = np.concatenate([rng.uniform(0, 1, n // 2), rng.uniform(0, 30, n // 2)]).astype(np.float32)
                y = np.concatenate([10 ** rng.uniform(-3, 5.5, n // 2), rng.uniform(-20, 20, n // 2)]).astype(np.float32)
            got = dev.diag_math("pow", x, y)
            want = oracle_parallel(oracle, pool, "pow", x, y)
            ok = same_bits(got, want)
            if not ok.all():
                k = int(np.flatnonzero(~ok)[0])
                raise AssertionError("pow( %r, %r ): HIP %r, oracle %r; %d mismatches" % (float(x[k]), float(y[k]), float(got[k]), float(want[k]), int((~ok).sum())))
    dev.close()
