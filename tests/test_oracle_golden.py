"""CPU: the C restatement oracle reproduces, bit for bit, the golden vectors generated from the REFERENCE's own kernels
(tests/golden/ref_vectors.npz, made by tests/golden/make_golden.py on a machine that has /root/reference).  This is
the pin that travels: it holds on boxes where the reference tree (and oracle/_ref) does not exist."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.golden import cases

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_port_matches_reference_golden_vectors():
    gold = np.load(os.path.join(GOLD, "ref_vectors.npz"))
    got = cases.run_all(O.Oracle("port"))
    assert set(got) == set(gold.files)
    for k in gold.files:
        a, b = got[k], gold[k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        if a.dtype == np.float32:
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (k, float(np.abs(a - b).max()))
        else:
            assert np.array_equal(a, b), k


def test_encoding_oracle_self_golden():
    """Not a reference pin (the upstream encoding source is absent): guards the frozen conventions against drift."""
    gold = np.load(os.path.join(GOLD, "encoding_vectors.npz"))
    got = cases.run_encoding()
    for k in gold.files:
        if got[k].dtype == np.float32:
            assert np.abs(got[k] - gold[k]).max() <= 1e-6 * max(1.0, np.abs(gold[k]).max()), k
        else:
            assert np.array_equal(got[k], gold[k]), k


def test_known_answers():
    port = O.Oracle("port")
    assert [port.morton3D(1, 0, 0), port.morton3D(0, 1, 0), port.morton3D(0, 0, 1)] == [1, 2, 4]
    for v in range(0, 1 << 12, 37):           # Morton round trip
        x, y, z = port.morton3D_invert(v), port.morton3D_invert(v >> 1), port.morton3D_invert(v >> 2)
        assert port.morton3D(x, y, z) == v
    u, f, _ = port.pcg32(4)
    assert np.all((f >= 0) & (f < 1))
    assert u[0] == 0x152CA78D                   # first output of the default-seeded PCG32 stream
    sh = port.spherical_harmonics(np.array([[0, 0, 1.0]], np.float32), 2)
    assert abs(sh[0, 0] - 0.28209479) < 1e-7 and abs(sh[0, 2] - 0.48860251) < 1e-7
    # ray through the sphere centre
    _, t0, _, t1, hit = port.sphere_intersect(0.5, [0, 0, 0], np.array([[0, 0, -2.0]], np.float32), np.array([[0, 0, 1.0]], np.float32))
    assert t0[0, 0] == 1.5 and t1[0, 0] == 2.5 and hit[0, 0]
    # constant alpha ray: T_i = a^i, background = a^(n-1)
    s = O.Samples(1, 10)
    s.equal, s.fixed = True, 10
    T, bg = port.cumprod(s, np.full((10, 1), 0.9, np.float32))
    assert np.allclose(T[:, 0], 0.9 ** np.arange(10), rtol=1e-6) and abs(bg[0, 0] - 0.9 ** 9) < 1e-6
