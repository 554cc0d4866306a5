"""GPU: empty and degenerate inputs of the entry points added late in round 1 (the reference's own tests cover none of
this; the C ABI must return PSDF_OK and touch nothing)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(dev, dims):
    from permuto_sdf_amd import FusedMLP
    torch.manual_seed(0)
    m = FusedMLP(dims).to(dev)
    return m, [l.weight for l in m.layers], [l.bias for l in m.layers]


def test_masked_forward_all_masked_and_none_masked(dev):
    from permuto_sdf_amd import PermutoEncoding
    from permuto_sdf_amd.encoding import encode_forward_raw
    from permuto_sdf_amd.mlp import mlp_forward_raw, pack_params
    enc = PermutoEncoding(3, 2 ** 12, 4, 2, np.geomspace(1.0, 1e-2, 4), concat_points=True, init_scale=1.0).to(dev)
    m, ws, bs = _net(dev, [enc.output_dims(), 32, 32, 32, 1])
    packed = pack_params(m.dims, ws, bs)
    x = torch.rand(1000, 3, device=dev) - 0.5
    a = (enc.cfg, x, enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), torch.ones(4, device=dev))
    ref_f = encode_forward_raw(*a)
    ref_y = mlp_forward_raw(m.dims, ref_f, packed)
    none = torch.zeros(1000, dtype=torch.bool, device=dev)
    allm = torch.ones(1000, dtype=torch.bool, device=dev)
    f = torch.full_like(ref_f, -3.0)
    y = torch.full_like(ref_y, -5.0)
    encode_forward_raw(*a, skip=allm, out=f)
    mlp_forward_raw(m.dims, ref_f, packed, skip=allm, out=y)
    assert bool((f == -3.0).all()) and bool((y == -5.0).all())
    encode_forward_raw(*a, skip=none, out=f)
    mlp_forward_raw(m.dims, f, packed, skip=none, out=y)
    assert torch.equal(f, ref_f) and torch.equal(y, ref_y)


def test_empty_batches(dev):
    from permuto_sdf_amd.mlp import mlp_backward_raw, mlp_double_backward
    dims = [52, 32, 32, 32, 33]
    m, ws, bs = _net(dev, dims)
    x = torch.zeros(52, 0, device=dev)
    gy = torch.zeros(33, 0, device=dev)
    dx, dWs, dbs = mlp_backward_raw(dims, x, ws, bs, gy)
    assert dx.shape == (52, 0) and all(float(w.abs().sum()) == 0 for w in dWs)
    dx2, dW2, db2 = mlp_double_backward(dims, x, ws, bs, gy, torch.zeros(52, 0, device=dev))
    assert dx2.shape == (52, 0) and all(float(w.abs().sum()) == 0 for w in dW2)
    dx_only, _, _ = mlp_backward_raw(dims, x, ws, bs, gy, need_dw=False)
    assert dx_only.shape == (52, 0)


def test_single_sample_backward_and_double_backward(dev):
    """N = 1: one partially filled 16-sample tile, seven idle waves per workgroup"""
    from permuto_sdf_amd import FusedMLP
    dims = [52, 32, 32, 32, 33]
    torch.manual_seed(2)
    m = FusedMLP(dims).to(dev)
    ref = torch.nn.Sequential(*[mod for i, l in enumerate(m.layers) for mod in ((torch.nn.Linear(l.in_features, l.out_features),) + ((torch.nn.GELU(),) if i < 3 else ()))]).to(dev).double()
    for a, b in zip([mod for mod in ref if isinstance(mod, torch.nn.Linear)], m.layers):
        a.weight.data.copy_(b.weight.data.double())
        a.bias.data.copy_(b.bias.data.double())

    def loss_of(net, xin):
        y = net(xin)
        (g,) = torch.autograd.grad(y[:, 0:1], xin, torch.ones_like(y[:, 0:1]), create_graph=True)
        return (g ** 2).sum() + y.sum()
    x = torch.randn(1, 52, device=dev)
    xa = x.clone().requires_grad_(True)
    xb = x.double().requires_grad_(True)
    loss_of(m, xa).backward()
    loss_of(ref, xb).backward()
    assert (xa.grad.double() - xb.grad).abs().max() <= 2e-4 * xb.grad.abs().max()
    lin = [mod for mod in ref if isinstance(mod, torch.nn.Linear)]
    for a, b in zip(m.layers, lin):
        assert (a.weight.grad.double() - b.weight.grad).abs().max() <= 2e-4 * max(1e-9, float(b.weight.grad.abs().max()))


def test_coarse2fine_window_host_evaluation_equals_the_device_formula():
    """encoding.Coarse2Fine evaluates its 24-float cosine window on the HOST (one upload instead of five launches); the reference
    (common_utils.py:51-62) and earlier rounds evaluate the same formula on the GPU.  Host libm and the device's cos may differ
    in the last place: measured here over a sweep of t, bound 1.2e-7 absolute (one ulp of a value in [0.5, 1)), which is the
    tolerance the bit-parity statements of the reference-step tests carry for the window."""
    import math
    import torch
    from permuto_sdf_amd.encoding import Coarse2Fine
    dev = torch.device("cuda:0")
    worst, differing, total = 0.0, 0, 0
    for L in (16, 24):
        c2f = Coarse2Fine(L).to(dev)
        for t in [i / 997.0 for i in range(0, 998, 7)] + [0.3, 1.0, 0.0]:
            host = c2f(t)
            alpha = torch.tensor(float(t) * L, dtype=torch.float32, device=dev)
            x = torch.clamp(alpha - torch.arange(L, dtype=torch.float32, device=dev), 0.0, 1.0)
            devw = 0.5 * (1.0 + torch.cos(math.pi * x + math.pi))
            d = (host - devw).abs()
            worst = max(worst, float(d.max()))
            differing += int((d > 0).sum())
            total += L
    print("Coarse2Fine host vs device: %d of %d window values differ, worst %.2e" % (differing, total, worst))
    assert worst <= 1.2e-7
