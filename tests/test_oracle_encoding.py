"""CPU tests of the encoding oracle itself (no GPU): the vectorised restatement must equal the literal
scalar loop bit for bit, and must satisfy the lattice invariants."""
import numpy as np
import pytest
import torch

from oracle import permuto_oracle as po


@pytest.mark.parametrize("P", [2, 3, 4])
def test_vectorised_equals_scalar(P):
    torch.manual_seed(0)
    L, T, F = 5, 2 ** 10, 2
    lat, sh = po.make_params(P, T, L, F, seed=1, init_scale=1.0)
    pts = torch.rand(200, P) - 0.5
    sl = np.geomspace(1.0, 1e-3, L)
    win = po.coarse2fine_window(0.6, L)
    a = po.encode_scalar(pts, lat, sl, sh, win, True, 1e-3)
    b = po.encode(pts, lat, sl, sh, win, True, 1e-3)
    assert a.shape == (200, po.output_dims(P, L, F, True))
    assert torch.equal(a, b)


@pytest.mark.parametrize("P", [3, 4])
def test_barycentric_invariants(P):
    torch.manual_seed(1)
    L = 8
    sl = np.geomspace(1.0, 1e-4, L)
    _, sh = po.make_params(P, 16, L, 2, seed=2)
    sf = po.scale_factors(sl, P)
    pts = torch.rand(5000, P) - 0.5
    for l in range(L):
        rem0, rank, bary = po.simplex(pts, sh[l], sf[l])
        b = bary[:, :P + 1]
        assert (b.sum(1) - 1).abs().max() < 1e-5          # partition of unity
        assert b.min() > -1e-4                             # inside the simplex
        assert (rem0.sum(1) == 0).all()                    # remainder-0 point lies on the hyperplane
        assert (rank.sort(1).values == torch.arange(P + 1)).all()  # rank is a permutation


def test_continuity_across_simplices():
    """The interpolant is continuous: moving a point by 1e-6 changes the features by O(1e-6 * slope)."""
    torch.manual_seed(2)
    P, L, T, F = 3, 4, 2 ** 12, 2
    lat, sh = po.make_params(P, T, L, F, seed=3, init_scale=1.0)
    sl = np.geomspace(1.0, 0.05, L)
    pts = torch.rand(20000, P) - 0.5
    win = torch.ones(L)
    a = po.encode(pts, lat, sl, sh, win)
    b = po.encode(pts + 1e-6, lat, sl, sh, win)
    assert (a - b).abs().max() < 5e-3


def test_window_and_output_dims():
    assert po.output_dims(3, 24, 2, True) == 52
    assert po.output_dims(4, 24, 2, True) == 52
    assert po.output_dims(3, 24, 2, False) == 48
    w = po.coarse2fine_window(0.3, 24)
    assert w.shape == (24,) and float(w[0]) == 1.0 and float(w[-1]) == 0.0
    assert torch.all(w[:-1] >= w[1:])


def test_hash_known_answers():
    # key (0,0,0) hashes to row 0; key (1,0,0): ((1*M)*M)*M mod 2^32 mod T
    M = po._hash_mult()
    rem0 = torch.tensor([[0, 0, 0, 0]])
    rank = torch.tensor([[0, 1, 2, 3]])
    idx = po.vertex_indices(rem0, rank, 2 ** 18)
    assert int(idx[0, 0]) == 0
    exp = 0
    for k in (1, 0, 0):
        exp = ((exp + k) * M) & 0xFFFFFFFF
    assert po.vertex_index_scalar([1, 0, 0, -1], [0, 1, 2, 3], 0, 3, 2 ** 18) == exp % 2 ** 18
