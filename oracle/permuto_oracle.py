"""CPU ORACLE for the permutohedral-lattice hash encoding.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module; it is the checker, never the product path.

PARITY UNPINNED.  The encoding arithmetic lives in the un-vendored third-party package
``github.com/RaduAlexandru/permutohedral_encoding`` (no version pin anywhere in the reference:
``README.md:40-49`` says ``git clone --recursive`` of the default branch; ``models.py:20`` imports it).
Its source is absent from ``/root/reference`` and there are no golden vectors for it, so this file
restates the published algorithm (Adams, Baek, Davis 2010, "Fast high-dimensional filtering using the
permutohedral lattice"; PermutoSDF, arXiv 2211.12562) with the conventions that the reference's call
sites force (``permuto_sdf_py/models/models.py:143-149,186,408-420``) and freezes the remaining ones
(SURVEY.md App. A.2-A.4).  Frozen conventions:

* ``scale_factor[l][i] = 1 / (sqrt((i+1)(i+2)) * scale_list[l])``             (no extra inv-std-dev term)
* ``lattice_values``: ``randn(T, L, F) * 1e-5`` permuted to ``[L, T, F]``; ``random_shift``: ``randn(L, P) * 10``
* rank tie-break: ``E_i - rem0_i <  E_j - rem0_j`` -> ``rank[i]++`` else ``rank[j]++``
* hash: ``h = 0; for i < P: h += (uint32) key[i]; h *= 2531011``; ``idx = h % T``
* output channel order ``[N, l*F + f]``; when ``concat_points`` the scaled points are appended as
  ``ceil(P/F)`` extra pseudo-levels, zero padded (so P=3,F=2 gives 4 extra channels, the last one 0) -- layout 1 --
  or as exactly P channels (SURVEY.md App. A.3 ``cat([sliced, scaling*points])``, 51 channels) -- layout 2.

What is NOT convention -- which simplex encloses a point, which lattice points are its vertices, the interpolation weights --
is checked against the definition of the lattice itself (vertices in A*_P, linear precision, Delaunay property by brute force,
exact interpolation of affine functions): ``tests/test_oracle_lattice_geometry.py``.

The VALUES of these conventions are not written in this file: they are parsed from the one header the HIP kernels
include, ``permuto_sdf_amd/csrc/encode_conventions.h`` (a data file: nothing of the product is imported or executed),
so that a change of convention there is followed by kernels, host code and oracle together
(``tests/test_encoding_conventions.py``).  ``CONV`` may be replaced by a test to evaluate a flipped convention.

All float arithmetic is fp32 with one rounding per operation (no FMA contraction) in the order written
below; the HIP kernels are compiled contraction-free for the same expressions, so forward parity is
expected to be bit-exact up to table-gather order.

Two implementations are provided and cross-checked in ``tests/test_oracle_encoding.py``:
``encode_scalar`` (pure-Python loop, the literal restatement, small N only) and ``encode``
(vectorised torch, differentiable: autograd through it is the oracle for the backward and
double-backward kernels, and it is what ``bench.py`` times as the CPU baseline).
"""
import math

import numpy as np
import torch

import os
import re

U32 = 0xFFFFFFFF
CONVENTIONS_HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "permuto_sdf_amd", "csrc",
                                  "encode_conventions.h")


def parse_conventions(path=CONVENTIONS_HEADER):
    out = {}
    for m in re.finditer(r"^#define\s+(PSDF_ENC_[A-Z0-9_]+)\s+([-+0-9.eE]+)\s*$", open(path).read(), re.M):
        v = m.group(2)
        out[m.group(1)] = float(v) if any(ch in v for ch in ".eE") else int(v)
    return out


CONV = parse_conventions()


def _hash_mult():
    return int(CONV["PSDF_ENC_HASH_MULTIPLIER"])


def _tie_later():
    return bool(CONV["PSDF_ENC_RANK_TIE_RAISES_LATER"])


def concat_layout(concat_points, layout=None):
    """-> 0 none, 1 padded pseudo-levels, 2 exactly P appended channels"""
    if not concat_points:
        return 0
    return int(CONV["PSDF_ENC_CONCAT_DEFAULT_LAYOUT"]) if layout is None else int(layout)


def scale_factors(scale_list, pos_dim):
    """[L, P] fp32.  Follows SURVEY.md App. A.2 (upstream ``Encoding`` constructor, not in tree)."""
    scale_list = np.asarray(scale_list, dtype=np.float64)
    sf = np.empty((len(scale_list), pos_dim), dtype=np.float64)
    for i in range(pos_dim):
        term = math.sqrt((i + 1) * (i + 2)) if CONV["PSDF_ENC_SCALE_SQRT_TERM"] else 1.0
        if CONV["PSDF_ENC_SCALE_INV_STDDEV"]:
            term /= (pos_dim + 1) * math.sqrt(2.0 / 3.0)
        sf[:, i] = 1.0 / (term * scale_list)
    return torch.from_numpy(sf.astype(np.float32))


def nr_extra_levels(pos_dim, nr_feat, concat_points):
    return int(math.ceil(pos_dim / nr_feat)) if concat_points else 0


def nr_point_channels(pos_dim, nr_feat, concat_points, layout=None):
    mode = concat_layout(concat_points, layout)
    return {0: 0, 1: nr_feat * nr_extra_levels(pos_dim, nr_feat, True), 2: pos_dim}[mode]


def output_dims(pos_dim, nr_levels, nr_feat, concat_points, layout=None):
    return nr_feat * nr_levels + nr_point_channels(pos_dim, nr_feat, concat_points, layout)


def coarse2fine_window(t, nr_levels):
    """Cosine-eased per-level window; same formula as the reference's own
    ``permuto_sdf_py/utils/common_utils.py:51-62`` (cosine_easing_window)."""
    alpha = t * nr_levels
    x = torch.clamp(alpha - torch.arange(nr_levels, dtype=torch.float32), 0.0, 1.0)
    return 0.5 * (1.0 + torch.cos(math.pi * x + math.pi))


# ----------------------------------------------------------------------------------------------
# literal scalar restatement (small N)
# ----------------------------------------------------------------------------------------------
def _f32(x):
    return np.float32(x)


def simplex_scalar(pos, shift, sf):
    """One (point, level): returns (rem0[P+1], rank[P+1], bary[P+2], elevated[P+1]).
    SURVEY.md App. A.3 lines 'elevate' .. 'bary'."""
    P = len(pos)
    E = [np.float32(0)] * (P + 1)
    sm = np.float32(0)
    for i in range(P, 0, -1):
        cf = _f32(_f32(pos[i - 1] + shift[i - 1]) * sf[i - 1])
        E[i] = _f32(sm - _f32(_f32(i) * cf))
        sm = _f32(sm + cf)
    E[0] = sm
    rem0 = [0] * (P + 1)
    rank = [0] * (P + 1)
    s = 0
    inv = 1.0 / (P + 1)  # double constant, as in the C expression `elevated[i] * (1.0 / (pos_dim + 1))`
    for i in range(P + 1):
        v = _f32(float(E[i]) * inv)
        up = _f32(np.ceil(v) * _f32(P + 1))
        down = _f32(np.floor(v) * _f32(P + 1))
        if _f32(up - E[i]) < _f32(E[i] - down):
            rem0[i] = int(up)
        else:
            rem0[i] = int(down)
        s += rem0[i]
    s = int(s / (P + 1))  # C integer division (exact: every rem0 is a multiple of P+1)
    for i in range(P):
        di = _f32(E[i] - _f32(rem0[i]))
        for j in range(i + 1, P + 1):
            dj = _f32(E[j] - _f32(rem0[j]))
            if (di < dj) if _tie_later() else (di <= dj):
                rank[i] += 1
            else:
                rank[j] += 1
    for i in range(P + 1):
        rank[i] += s
        if rank[i] < 0:
            rank[i] += P + 1
            rem0[i] += P + 1
        elif rank[i] > P:
            rank[i] -= P + 1
            rem0[i] -= P + 1
    bary = [np.float32(0)] * (P + 2)
    for i in range(P + 1):
        delta = _f32(float(_f32(E[i] - _f32(rem0[i]))) * inv)
        bary[P - rank[i]] = _f32(bary[P - rank[i]] + delta)
        bary[P + 1 - rank[i]] = _f32(bary[P + 1 - rank[i]] - delta)
    bary[0] = _f32(float(bary[0]) + (1.0 + float(bary[P + 1])))
    return rem0, rank, bary, E


def vertex_index_scalar(rem0, rank, remainder, P, capacity):
    h = 0
    for i in range(P):
        k = rem0[i] + remainder
        if rank[i] > P - remainder:
            k -= P + 1
        h = (h + (k & U32)) & U32
        h = (h * _hash_mult()) & U32
    return h % capacity


def encode_scalar(points, lattice_values, scale_list, shifts, window, concat_points=False, points_scaling=1.0, layout=None):
    """Pure-Python loop, literal restatement of SURVEY.md App. A.3.  O(N*L) Python: keep N small."""
    pts = points.detach().cpu().numpy().astype(np.float32)
    lat = lattice_values.detach().cpu().numpy().astype(np.float32)
    N, P = pts.shape
    L, T, F = lat.shape
    sf = scale_factors(scale_list, P).numpy()
    sh = shifts.detach().cpu().numpy().astype(np.float32)
    win = window.detach().cpu().numpy().astype(np.float32)
    extra = nr_extra_levels(P, F, concat_points)
    C = output_dims(P, L, F, concat_points, layout)
    out = np.zeros((N, C), dtype=np.float32)
    for n in range(N):
        for l in range(L):
            rem0, rank, bary, _ = simplex_scalar(pts[n], sh[l], sf[l])
            acc = [np.float32(0)] * F
            for r in range(P + 1):
                idx = vertex_index_scalar(rem0, rank, r, P, T)
                bw = _f32(bary[r] * win[l])
                for f in range(F):
                    acc[f] = _f32(acc[f] + _f32(lat[l, idx, f] * bw))
            for f in range(F):
                out[n, l * F + f] = acc[f]
        for e in range(extra):
            for f in range(F):
                d = e * F + f
                if L * F + d < C:       # channels beyond P exist only in the padded layout
                    out[n, L * F + d] = _f32(pts[n, d] * _f32(points_scaling)) if d < P else 0.0
    return torch.from_numpy(out)


# ----------------------------------------------------------------------------------------------
# vectorised, differentiable torch restatement
# ----------------------------------------------------------------------------------------------
def simplex(points, shift, sf):
    """Vectorised `simplex_scalar` for one level.  points [N,P] fp32 (may require grad).
    Returns rem0 [N,P+1] int64, rank [N,P+1] int64, bary [N,P+2] fp32 (differentiable wrt points)."""
    N, P = points.shape
    dev = points.device      # (tests may run the restatement on a GPU tensor: same IEEE operations, elementwise, no contraction)
    shift, sf = torch.as_tensor(shift).to(dev), torch.as_tensor(sf).to(dev)
    x = (points + shift) * sf
    cols = [None] * (P + 1)
    sm = torch.zeros(N, dtype=points.dtype, device=dev)
    for i in range(P, 0, -1):
        cf = x[:, i - 1]
        cols[i] = sm - float(i) * cf
        sm = sm + cf
    cols[0] = sm
    E = torch.stack(cols, dim=1)
    inv = 1.0 / (P + 1)
    with torch.no_grad():
        Ed = E.detach()
        v = (Ed.double() * inv).float()
        up = torch.ceil(v) * float(P + 1)
        down = torch.floor(v) * float(P + 1)
        rem0 = torch.where((up - Ed) < (Ed - down), up, down).to(torch.int64)
        s = torch.div(rem0.sum(1), P + 1, rounding_mode="trunc")
        d = Ed - rem0.float()
        rank = torch.zeros(N, P + 1, dtype=torch.int64, device=dev)
        for i in range(P):
            for j in range(i + 1, P + 1):
                lt = (d[:, i] < d[:, j]) if _tie_later() else (d[:, i] <= d[:, j])
                rank[:, i] += lt
                rank[:, j] += ~lt
        rank = rank + s[:, None]
        neg = rank < 0
        big = rank > P
        rank = rank + neg * (P + 1) - big * (P + 1)
        rem0 = rem0 + neg * (P + 1) - big * (P + 1)
    bary = torch.zeros(N, P + 2, dtype=points.dtype, device=dev)
    rows = torch.arange(N, device=dev)
    for i in range(P + 1):
        delta = ((E[:, i] - rem0[:, i].float()).double() * inv).to(points.dtype)
        plus = torch.zeros_like(bary)
        plus[rows, P - rank[:, i]] = delta
        minus = torch.zeros_like(bary)
        minus[rows, P + 1 - rank[:, i]] = delta
        bary = (bary + plus) - minus  # same order as `+= delta` then `-= delta` (distinct slots)
    b0 = (bary[:, 0].double() + (1.0 + bary[:, P + 1].double())).to(points.dtype)
    bary = torch.cat([b0[:, None], bary[:, 1:]], dim=1)
    return rem0, rank, bary


def vertex_indices(rem0, rank, capacity):
    """[N, P+1] int64 hashed table rows of the P+1 simplex vertices (remainder 0..P)."""
    N, P1 = rem0.shape
    P = P1 - 1
    idx = []
    for r in range(P + 1):
        h = torch.zeros(N, dtype=torch.int64, device=rem0.device)
        for i in range(P):
            k = rem0[:, i] + r - (rank[:, i] > (P - r)).to(torch.int64) * (P + 1)
            h = (h + (k & U32)) & U32
            h = (h * _hash_mult()) & U32
        idx.append(h % capacity)
    return torch.stack(idx, dim=1)


def encode(points, lattice_values, scale_list, shifts, window, concat_points=False, points_scaling=1.0, layout=None):
    """points [N,P], lattice_values [L,T,F], shifts [L,P], window [L]  ->  [N, F*(L+extra)].
    Differentiable wrt points and lattice_values (rank / rem0 are piecewise constant)."""
    N, P = points.shape
    L, T, F = lattice_values.shape
    sf = scale_factors(scale_list, P)
    outs = []
    for l in range(L):
        rem0, rank, bary = simplex(points, shifts[l], sf[l])
        idx = vertex_indices(rem0, rank, T)
        acc = torch.zeros(N, F, dtype=points.dtype, device=points.device)
        for r in range(P + 1):
            fv = lattice_values[l].index_select(0, idx[:, r])
            bw = bary[:, r] * window[l]
            acc = acc + fv * bw[:, None]
        outs.append(acc)
    npc = nr_point_channels(P, F, concat_points, layout)
    if npc:
        pad = torch.zeros(N, npc - P, dtype=points.dtype, device=points.device)
        outs.append(torch.cat([points * float(points_scaling), pad], dim=1))
    return torch.cat(outs, dim=1)


def make_params(pos_dim, capacity, nr_levels, nr_feat, seed=0, init_scale=None, random_shift=True):
    """Seeded parameter construction following SURVEY.md App. A.2."""
    if init_scale is None:
        init_scale = CONV["PSDF_ENC_LATTICE_INIT_SCALE"]
    g = torch.Generator().manual_seed(seed)
    lattice = (torch.randn(capacity, nr_levels, nr_feat, generator=g) * init_scale).permute(1, 0, 2).contiguous()
    if random_shift:
        shifts = torch.randn(nr_levels, pos_dim, generator=g) * CONV["PSDF_ENC_RANDOM_SHIFT_SCALE"]
    else:
        shifts = torch.zeros(nr_levels, pos_dim)
    return lattice, shifts
