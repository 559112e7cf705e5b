"""Every float32 there is through the one-argument functions of the arithmetic layer: the HIP
implementations (csrc/pt_math.hpp via pbr_diag_math) against the oracle's, bit for bit, for all 2^32 inputs of sin, cos,
tan, acos, atan and the RNG's hash step; a billion random pairs through pow; 64 random materials x 2^18 directions
through the BRDF and new-ray stages.  Since round 4 ALL of it runs in the default GPU suite (pow, acos,
atan and tan are the builtins the default BRDF's sampling lives on, pt_brdf.cl:278-330, and a builder-run log is not
driver-run evidence): nine tests, 87 s on the GPU box with its host cores checking against the oracle
(profiles/r04/math_exhaustive.txt).  PBR_QUICK=1 skips them for a fast local iteration."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
quick = pytest.mark.skipif(os.environ.get("PBR_QUICK") == "1", reason="PBR_QUICK=1: the exhaustive sweeps are skipped")

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


@quick
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


@quick
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


@quick
@pytest.mark.parametrize("brdf", [1, 0])
def test_brdf_and_new_ray_stage_soak(pbr, oracle, gpu_device, brdf):
    """64 random materials x 2^18 random (arriving, leaving, normal) triples through the BRDF evaluation and the
    new-ray sampling stages (pbr_diag_brdf / pbr_diag_new_ray vs the oracle's): 1.7e7 evaluations and as many samples
    per BRDF model, glass, mirrors, anisotropic lobes and back-facing normals among them."""
    import ctypes
    from conftest import same_values, describe_mismatch
    pbr.cfg_reset()
    pbr.cfg_set(**{"render.brdf": brdf})
    sc = pbr.HostScene.generate("cornell", 1, 0)
    dev = pbr.Device(gpu_device)
    rng = np.random.default_rng(77 + brdf)
    n = 1 << 18
    fp = ctypes.POINTER(ctypes.c_float)

    def unit(a):
        return a / np.linalg.norm(a, axis=1, keepdims=True)

    for round_ in range(64):
        normal = unit(rng.normal(size=(n, 3)))
        out_dir = unit(rng.normal(size=(n, 3)))
        out_dir -= 2 * np.maximum(0, (out_dir * normal).sum(1, keepdims=True)) * normal
        in_dir = unit(rng.normal(size=(n, 3)))
        in_dir += 2 * np.maximum(0, -(in_dir * normal).sum(1, keepdims=True)) * normal
        ev = np.zeros((n, 16), np.float32)
        ev[:, 0:3], ev[:, 3:6], ev[:, 6:9] = out_dir, in_dir, normal
        nr = np.zeros((n, 12), np.float32)
        nr[:, 0:3] = rng.uniform(-1, 1, (n, 3))
        nr[:, 3:6], nr[:, 6:9] = out_dir, normal
        nr[: n // 4, 6:9] *= -1
        nr[:, 9] = rng.uniform(0.01, 5, n)
        nr[:, 10] = rng.uniform(0, 300, n)
        d = 1.0 if rng.integers(2) else float(rng.uniform(0, 1))
        ni = float(rng.uniform(1, 2))
        kd, ks = rng.uniform(0, 1, 3), rng.uniform(0, 1, 3)
        if brdf == 1:
            nu, nv = [0.0 if rng.integers(5) == 0 else float(10 ** rng.uniform(-1, 5)) for _ in range(2)]
            mtl = [d, ni, nu, nv, float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), 0, 0, *kd, 0, *ks, 0]
        else:
            mtl = [d, ni, float(rng.uniform(0.01, 1)), float(rng.choice([0.0, 1.0, rng.uniform(0, 1)])), *kd, 0, *ks, 0]
        m0 = np.asarray(mtl, np.float32)
        desc = pbr.SceneDesc.from_buffer_copy(sc.desc)
        mats = np.tile(m0, (sc.desc.num_materials, 1)).copy()
        desc.materials, desc.brdf = mats.ctypes.data, brdf
        dev.upload_scene(desc)
        want = np.empty((n, 4), np.float32)
        oracle.lib().orc_brdf_eval(brdf, m0.ctypes.data, ev.ctypes.data_as(fp), n, want.ctypes.data_as(fp))
        got = dev.diag_brdf(ev)
        assert same_values(got, want), "brdf %r: %s" % (mtl, describe_mismatch(got, want))
        want = np.empty((n, 8), np.float32)
        oracle.lib().orc_new_ray(brdf, m0.ctypes.data, nr.ctypes.data_as(fp), n, want.ctypes.data_as(fp))
        got = dev.diag_new_ray(nr)
        assert same_values(got, want), "new ray %r: %s" % (mtl, describe_mismatch(got, want))
    dev.close()
