"""The encoding oracle against the DEFINITION of the permutohedral lattice (CPU, no GPU, no upstream source needed).

Upstream `permutohedral_encoding` is absent, so the oracle cannot be pinned to its outputs.  What can be
pinned without it is the geometry the published algorithm (Adams, Baek, Davis 2010, section 3) is defined by; none of the checks
below re-uses the oracle's own arithmetic:

  * the P+1 vertices the oracle addresses are points of the lattice A*_P scaled by P+1: integer coordinates, all congruent to
    the vertex's remainder modulo P+1, summing to zero;
  * the barycentric weights reproduce the elevated point from those vertices (linear precision), are a partition of unity and
    non-negative: the point lies in the simplex;
  * that simplex is a Delaunay cell of the lattice: no lattice point lies strictly inside its circumsphere (brute force over the
    neighbourhood), so it is THE cell the algorithm must find, not merely some enclosing set of lattice points;
  * interpolation of an affine function of the vertex position returns that function at the point.

What stays convention (hash, scale factors, shift, channel layout) is listed in csrc/encode_conventions.h.
"""
import itertools

import numpy as np
import pytest
import torch

from oracle import permuto_oracle as po


def _elevate(x):
    """[N,P] float64 -> [N,P+1]: the basis of the hyperplane sum = 0 written out as a matrix (Adams et al. eq. for E, without
    the per-axis normalisation, which the oracle keeps inside its scale factors)."""
    N, P = x.shape
    B = np.zeros((P + 1, P))
    B[0, :] = 1.0
    for i in range(1, P + 1):
        B[i, i - 1] = -float(i)
        B[i, i:] = 1.0
    assert np.allclose(B.sum(0), 0.0)                       # every column lies in the hyperplane
    assert np.linalg.matrix_rank(B) == P
    return x @ B.T


def _vertices(rem0, rank):
    """[N,P+1] ints x2 -> [N, P+1 (vertex k), P+1 (coordinate)]: vertex k = rem0 + canonical_k[rank]."""
    N, P1 = rem0.shape
    P = P1 - 1
    v = np.empty((N, P1, P1), dtype=np.int64)
    for k in range(P1):
        v[:, k, :] = rem0 + k - (rank > (P - k)) * P1
    return v


def _setup(P, L=6, N=400, seed=0):
    torch.manual_seed(seed)
    sl = np.geomspace(1.0, 1e-3, L)
    _, sh = po.make_params(P, 16, L, 2, seed=seed + 1)
    sf = po.scale_factors(sl, P)
    pts = torch.rand(N, P) - 0.5
    return pts, sh, sf


@pytest.mark.parametrize("P", [2, 3, 4])
def test_vertices_are_lattice_points_and_weights_reproduce_the_point(P):
    pts, sh, sf = _setup(P)
    for l in range(sf.shape[0]):
        rem0, rank, bary = po.simplex(pts, sh[l], sf[l])
        rem0, rank, bary = rem0.numpy(), rank.numpy(), bary.numpy().astype(np.float64)
        x = (pts.numpy().astype(np.float32) + sh[l].numpy()) * sf[l].numpy()         # the oracle's fp32 input scaling
        E = _elevate(x.astype(np.float64))
        v = _vertices(rem0, rank)
        for k in range(P + 1):
            assert ((v[:, k, :] - k) % (P + 1) == 0).all()                           # remainder-k lattice point
            assert (v[:, k, :].sum(1) == 0).all()                                    # on the hyperplane
        b = bary[:, :P + 1]
        assert np.abs(b.sum(1) - 1).max() < 1e-5 and b.min() > -1e-4
        recon = np.einsum("nk,nkc->nc", b, v.astype(np.float64))
        scale = np.maximum(1.0, np.abs(E).max(1, keepdims=True))
        assert (np.abs(recon - E) / scale).max() < 2e-5, "barycentric weights do not reproduce the elevated point"


def _circumsphere(v):
    """v [P+1 vertices, P+1 coords] in the hyperplane -> (centre, radius^2), solved in hyperplane coordinates."""
    P1 = v.shape[0]
    A = 2.0 * (v[1:] - v[0])                                 # [P, P+1]
    rhs = (v[1:] ** 2).sum(1) - (v[0] ** 2).sum()
    A = np.vstack([A, np.ones((1, P1))])                     # + the hyperplane constraint sum(c) = 0
    rhs = np.append(rhs, 0.0)
    c = np.linalg.lstsq(A, rhs, rcond=None)[0]
    assert np.abs(A @ c - rhs).max() < 1e-9
    return c, ((v[0] - c) ** 2).sum()


@pytest.mark.parametrize("P", [2, 3])
def test_simplex_is_a_delaunay_cell_of_the_lattice(P):
    """No lattice point strictly inside the circumsphere of the simplex the oracle picked; and every simplex has the same
    circumradius (the cells of A*_P are congruent)."""
    pts, sh, sf = _setup(P, L=4, N=60, seed=3)
    P1 = P + 1
    offsets = np.array(list(itertools.product(range(-2, 3), repeat=P1)), dtype=np.int64) * P1     # multiples of P+1
    radii = []
    for l in range(sf.shape[0]):
        rem0, rank, _ = po.simplex(pts, sh[l], sf[l])
        v_all = _vertices(rem0.numpy(), rank.numpy()).astype(np.float64)
        for n in range(pts.shape[0]):
            v = v_all[n]
            c, r2 = _circumsphere(v)
            radii.append(r2)
            for k in range(P1):                               # every remainder class around the remainder-0 vertex
                cand = rem0[n].numpy()[None, :] + k + offsets
                cand = cand[cand.sum(1) == 0].astype(np.float64)
                d2 = ((cand - c) ** 2).sum(1)
                assert (d2 > r2 - 1e-6).all(), "a lattice point lies inside the circumsphere"
                # the simplex's own vertices are among the candidates and sit ON the sphere
                assert np.isclose(d2, r2, atol=1e-6).sum() >= 1
    radii = np.array(radii)
    assert np.allclose(radii, radii[0], rtol=1e-9)
    # circumradius^2 of the A*_P cell scaled by P+1: P (P+1) (P+2) / 12
    assert np.isclose(radii[0], P * (P + 1) * (P + 2) / 12.0)


@pytest.mark.parametrize("P", [2, 3, 4])
def test_affine_functions_are_interpolated_exactly(P):
    """Table values set to an affine function of the vertex position (no collisions: one private row per distinct vertex) ->
    the encoded feature is that function of the elevated point."""
    pts, sh, sf = _setup(P, L=1, N=300, seed=5)
    rem0, rank, _ = po.simplex(pts, sh[0], sf[0])
    T = 2 ** 20
    idx = po.vertex_indices(rem0, rank, T).numpy()
    v = _vertices(rem0.numpy(), rank.numpy())
    rng = np.random.default_rng(0)
    a, b = rng.normal(size=(2, P + 1)), rng.normal(size=2)
    lat = np.zeros((1, T, 2), dtype=np.float32)
    owner = {}
    for n in range(idx.shape[0]):
        for k in range(P + 1):
            key = tuple(v[n, k])
            assert owner.setdefault(int(idx[n, k]), key) == key, "hash collision in the test set-up: pick another seed"
            lat[0, idx[n, k]] = (a @ v[n, k].astype(np.float64) + b).astype(np.float32)
    sl = np.geomspace(1.0, 1e-3, 6)[:1]
    out = po.encode(pts, torch.from_numpy(lat), sl, sh[:1], torch.ones(1)).numpy().astype(np.float64)
    x = (pts.numpy().astype(np.float32) + sh[0].numpy()) * sf[0].numpy()
    E = _elevate(x.astype(np.float64))
    want = E @ a.T + b
    assert np.abs(out - want).max() < 5e-5 * max(1.0, np.abs(want).max())
    # ... and the oracle's gradient with respect to the positions (torch autograd through the barycentric weights: the oracle of
    # the HIP position-backward / double-backward kernels) is the analytic slope of that affine function: sf_j * (B^T a)_j
    B = np.zeros((P + 1, P))
    B[0, :] = 1.0
    for i in range(1, P + 1):
        B[i, i - 1] = -float(i)
        B[i, i:] = 1.0
    for f in range(2):
        q = pts.clone().requires_grad_(True)
        po.encode(q, torch.from_numpy(lat), sl, sh[:1], torch.ones(1))[:, f].sum().backward()
        slope = (B.T @ a[f]) * sf[0].numpy().astype(np.float64)
        got = q.grad.numpy().astype(np.float64)
        assert np.abs(got - slope[None, :]).max() < 2e-4 * np.abs(slope).max()
