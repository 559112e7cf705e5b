"""Frozen regression fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py).
CPU: the oracle and the host BVH builder still produce them bit for bit on THIS machine.
GPU: the HIP path produces them too (test_gpu_parity.py)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, same_values, describe_mismatch

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden  # noqa: E402


@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_oracle_and_builder_reproduce_golden(name):
    want = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    got = make_golden.render_case(name)
    assert np.array_equal(got["bvh"], want["bvh"]), "host BVH builder drifted"
    assert got["px_dim"] == want["px_dim"]
    assert same_values(got["ray_t"], want["ray_t"]) and np.array_equal(got["ray_face"], want["ray_face"])
    assert np.array_equal(got["ray_counts"], want["ray_counts"])
    assert same_values(got["image"], want["image"]), describe_mismatch(got["image"], want["image"])
    assert same_values(got["debug"], want["debug"])
    assert np.array_equal(got["counters"], want["counters"])
