"""Host side of csrc/fused.hip: encoding -> MLP in one launch (reference pair: models.py:186-192)."""
import torch

from . import _lib as L
from .mlp import _dims_array


def fused_supported(cfg, dims):
    return cfg.pos_dim == 3 and cfg.nr_feat == 2 and dims[0] == cfg.channels


def encode_mlp_forward_raw(cfg, positions, lattice, scale_factor, shifts, window, dims, packed, skip=None,
                           want_feat=False, out=None):
    """-> (Y [dims[-1], N] feature-major, feat [dims[0], N] or None).  `skip` [N] bool/uint8: masked samples may be
    left unevaluated (pass `out` to control what their entries hold)."""
    L.require_cuda(positions, lattice, packed)
    if not fused_supported(cfg, dims):
        raise ValueError("fused encode->MLP needs pos_dim=3, 2 features/level and dims[0]=2*(levels+pseudo-levels); got "
                         "pos_dim=%d nr_feat=%d dims[0]=%d" % (cfg.pos_dim, cfg.nr_feat, dims[0]))
    N = positions.shape[0]
    Y = out if out is not None else torch.empty((dims[-1], N), dtype=torch.float32, device=positions.device)
    feat = torch.empty((dims[0], N), dtype=torch.float32, device=positions.device) if want_feat else None
    L.call("psdf_encode_mlp_forward", L.c_l(N), L.c_i(cfg.nr_levels), L.c_i(cfg.capacity), L.ptr(positions), L.ptr(lattice),
           L.ptr(scale_factor), L.ptr(shifts), L.ptr(window), L.c_i(int(cfg.concat_mode)), L.c_f(cfg.points_scaling),
           L.c_i(len(dims) - 1), _dims_array(dims), L.ptr(packed), L.ptr(skip), L.ptr(feat), L.ptr(Y), L.stream())
    return Y, feat
