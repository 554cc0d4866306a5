"""GPU: fused encode->MLP launch (csrc/fused.hip) vs the unfused pair psdf_encode_forward + psdf_mlp_forward.  The
encoding by-product is BIT-identical (same expressions in the same order).  The network output is bit-identical where
both launches use the fp32 matrix pipe; nets that psdf_mlp_forward evaluates with split-bf16 operands (mlp_device.h)
agree to fp32 rounding level (different summation order).  Skip-mask and by-product feature tensor."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # (levels, concat_points, hidden..., out)
    (16, True, [64, 64, 64, 1]),
    (24, True, [64, 64, 64, 1]),
    (24, True, [32, 32, 32, 33]),
    (24, True, [32, 32, 32, 1]),
    (16, False, [64, 64, 64, 33]),
    (3, True, [32, 32, 32, 3]),      # odd number of levels+pseudo-levels: last pair half empty
    (1, False, [32, 32, 32, 1]),
]


def _setup(dev, levels, concat, net, seed=0):
    from permuto_sdf_amd import FusedMLP, PermutoEncoding
    from permuto_sdf_amd.mlp import pack_params
    torch.manual_seed(seed)
    enc = PermutoEncoding(3, 2 ** 14, levels, 2, np.geomspace(1.0, 1e-3, levels), concat_points=concat,
                          concat_points_scaling=0.7, init_scale=1.0).to(dev)
    mlp = FusedMLP([enc.output_dims()] + net).to(dev)
    packed = pack_params(mlp.dims, [l.weight for l in mlp.layers], [l.bias for l in mlp.layers])
    win = torch.rand(levels, device=dev)
    return enc, mlp, packed, win


def _same_output(net, yf, y):
    if net == [64, 64, 64, 33]:      # too wide for the split-bf16 image: both launches are the fp32 MFMA evaluator
        assert torch.equal(yf, y)
    else:
        assert (yf - y).abs().max() <= 3e-6 * max(1.0, y.abs().max().item())


@pytest.mark.parametrize("levels,concat,net", CASES)
@pytest.mark.parametrize("N", [1, 33, 10007])
def test_fused_equals_unfused_bitwise(dev, levels, concat, net, N):
    from permuto_sdf_amd.encoding import encode_forward_raw
    from permuto_sdf_amd.fused import encode_mlp_forward_raw
    from permuto_sdf_amd.mlp import mlp_forward_raw
    enc, mlp, packed, win = _setup(dev, levels, concat, net)
    x = torch.rand(N, 3, device=dev) - 0.5
    args = (enc.cfg, x, enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), win)
    feat = encode_forward_raw(*args)
    y = mlp_forward_raw(mlp.dims, feat, packed)
    yf, ff = encode_mlp_forward_raw(*args, mlp.dims, packed, want_feat=True)
    assert torch.equal(ff, feat)
    _same_output(net, yf, y)
    yf2, none = encode_mlp_forward_raw(*args, mlp.dims, packed)
    assert none is None and torch.equal(yf2, yf)


def test_skip_mask_leaves_fully_masked_tiles_untouched(dev):
    from permuto_sdf_amd.fused import encode_mlp_forward_raw
    enc, mlp, packed, win = _setup(dev, 16, True, [64, 64, 64, 1])
    N = 4096 + 5
    x = torch.rand(N, 3, device=dev) - 0.5
    args = (enc.cfg, x, enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), win)
    y, _ = encode_mlp_forward_raw(*args, mlp.dims, packed)
    skip = torch.zeros(N, dtype=torch.bool, device=dev)
    skip[64:1024] = True           # whole tiles
    skip[2000:2010] = True         # part of a tile: still evaluated
    out = torch.full((1, N), -7.0, device=dev)
    ys, _ = encode_mlp_forward_raw(*args, mlp.dims, packed, skip=skip, out=out)
    assert torch.equal(ys[0, :64], y[0, :64]) and torch.equal(ys[0, 1024:], y[0, 1024:])
    assert bool((ys[0, 64:1024] == -7.0).all())


def test_argument_errors(dev):
    from permuto_sdf_amd._lib import PsdfError
    from permuto_sdf_amd.fused import encode_mlp_forward_raw
    enc, mlp, packed, win = _setup(dev, 16, True, [64, 64, 64, 1])
    x = torch.rand(10, 3, device=dev)
    with pytest.raises(ValueError):
        encode_mlp_forward_raw(enc.cfg, x, enc.lattice_values.detach(), enc.scale_factor,
                               enc.random_shift_per_level.detach(), win, [35, 64, 64, 64, 1], packed)
    y, _ = encode_mlp_forward_raw(enc.cfg, x[:0], enc.lattice_values.detach(), enc.scale_factor,
                                  enc.random_shift_per_level.detach(), win, mlp.dims, packed)
    assert y.shape == (1, 0)
