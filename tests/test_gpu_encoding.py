"""GPU parity: HIP permutohedral encoding (through the C ABI) vs the CPU oracle on identical seeded inputs.
Tolerances: forward 1e-6 abs on O(1) lattice values (same op order, fp32); gradients 1e-5 relative
(atomic accumulation order differs)."""
import numpy as np
import pytest
import torch

from oracle import permuto_oracle as po

pytestmark = pytest.mark.gpu


def _make(P, L, T, F, N, seed, concat=True, scaling=1e-3, init_scale=1.0, coarsest=1.0, finest=1e-4):
    from permuto_sdf_amd import PermutoEncoding
    torch.manual_seed(seed)
    sl = np.geomspace(coarsest, finest, L)
    enc = PermutoEncoding(P, T, L, F, sl, appply_random_shift_per_level=True, concat_points=concat,
                          concat_points_scaling=scaling, init_scale=init_scale)
    pts = (torch.rand(N, P) - 0.5)
    win = po.coarse2fine_window(0.8, L)
    return enc, sl, pts, win


@pytest.mark.parametrize("P,L,T,N", [(3, 1, 2 ** 18, 65536), (3, 24, 2 ** 18, 20000), (4, 24, 2 ** 18, 8192),
                                      (3, 16, 2 ** 12, 5000), (2, 4, 2 ** 10, 1000)])
def test_forward_matches_oracle(dev, P, L, T, N):
    enc, sl, pts, win = _make(P, L, T, 2, N, seed=P * 100 + L)
    ref = po.encode(pts, enc.lattice_values.detach(), sl, enc.random_shift_per_level.detach(), win, True, 1e-3)
    enc = enc.to(dev)
    out = enc(pts.to(dev), win.to(dev))
    assert out.shape == (N, enc.output_dims()) == ref.shape
    err = (out.cpu() - ref).abs().max().item()
    assert err <= 1e-6 * max(1.0, ref.abs().max().item()), err


def test_forward_edge_cases(dev):
    enc, sl, pts, win = _make(3, 8, 2 ** 14, 2, 1, seed=5, concat=False)
    enc_d = enc.to(dev)
    # empty input
    out = enc_d(torch.zeros(0, 3, device=dev), win.to(dev))
    assert out.shape == (0, 16)
    # single point, ragged block (N not a multiple of the block), far-away and lattice-aligned coordinates
    pts = torch.tensor([[0.0, 0.0, 0.0], [1e3, -1e3, 5e2], [0.5, 0.5, 0.5], [-0.25, 0.125, 1.0]])
    pts = torch.cat([pts, torch.rand(300, 3) * 4 - 2])
    ref = po.encode(pts, enc.lattice_values.detach().cpu(), sl, enc.random_shift_per_level.detach().cpu(), win)
    out = enc_d(pts.to(dev), win.to(dev))
    assert torch.allclose(out.cpu(), ref, rtol=0, atol=1e-5)  # |coords| up to 1e3 at scale 1e-4: elevated ~1e7


@pytest.mark.parametrize("P", [3, 4])
def test_backward_matches_oracle_autograd(dev, P):
    L_, T, N = 12, 2 ** 14, 6000
    enc, sl, pts, win = _make(P, L_, T, 2, N, seed=7 + P)
    g = torch.randn(N, enc.output_dims())
    lat = enc.lattice_values.detach().clone().requires_grad_(True)
    p_ref = pts.clone().requires_grad_(True)
    ref = po.encode(p_ref, lat, sl, enc.random_shift_per_level.detach(), win, True, 1e-3)
    ref.backward(g)
    enc = enc.to(dev)
    p = pts.to(dev).requires_grad_(True)
    out = enc(p, win.to(dev))
    out.backward(g.to(dev))
    gl, gl_ref = enc.lattice_values.grad.cpu(), lat.grad
    assert (gl - gl_ref).abs().max() <= 1e-5 * gl_ref.abs().max() + 1e-7
    gp, gp_ref = p.grad.cpu(), p_ref.grad
    assert (gp - gp_ref).abs().max() <= 2e-5 * gp_ref.abs().max()


def test_double_backward_matches_oracle_autograd(dev):
    P, L_, T, N = 3, 10, 2 ** 13, 4000
    enc, sl, pts, win = _make(P, L_, T, 2, N, seed=21)
    C = enc.output_dims()
    torch.manual_seed(3)
    w1 = torch.randn(C, 5)            # stand-in for the MLP: sdf = tanh(feat @ w1).sum
    u = torch.randn(N, P)

    def run(encode_fn, lat, pts_, w1_, u_):
        pts_ = pts_.clone().requires_grad_(True)
        feat = encode_fn(pts_)
        sdf = torch.tanh(feat @ w1_).sum(1, keepdim=True)
        grad = torch.autograd.grad(sdf, pts_, torch.ones_like(sdf), create_graph=True)[0]
        loss = ((grad * u_).sum(1) ** 2).mean() + sdf.mean()
        return torch.autograd.grad(loss, [lat], retain_graph=False)[0], grad.detach()

    lat_ref = enc.lattice_values.detach().clone().requires_grad_(True)
    shifts = enc.random_shift_per_level.detach().clone()
    g_ref, grad_ref = run(lambda x: po.encode(x, lat_ref, sl, shifts, win, True, 1e-3), lat_ref, pts, w1, u)
    enc = enc.to(dev)
    wd = win.to(dev)
    g_hip, grad_hip = run(lambda x: enc(x, wd), enc.lattice_values, pts.to(dev), w1.to(dev), u.to(dev))
    assert (grad_hip.cpu() - grad_ref).abs().max() <= 2e-5 * grad_ref.abs().max()
    assert (g_hip.cpu() - g_ref).abs().max() <= 2e-5 * g_ref.abs().max()


def test_lattice_grad_linearity_full_size(dev):
    """Size-independent property at the BASELINE size (2M points, 16 levels): the lattice gradient is
    linear in the upstream gradient and its total mass equals sum(window_l * sum_n g) per level and feature
    (barycentric weights sum to one)."""
    P, L_, T, N = 3, 16, 2 ** 18, 2 * 1024 * 1024
    enc, sl, _, win = _make(P, L_, T, 2, 8, seed=33, concat=False)
    enc = enc.to(dev)
    torch.manual_seed(0)
    pts = torch.rand(N, P, device=dev) - 0.5
    g = torch.randn(N, enc.output_dims(), device=dev)
    out = enc(pts, win.to(dev))
    (gl,) = torch.autograd.grad(out, enc.lattice_values, g)
    mass = gl.double().sum(1)                                       # [L, F]
    expect = (g.double().sum(0).view(L_, 2)) * win.to(dev).double()[:, None]
    assert torch.allclose(mass, expect, rtol=1e-3, atol=1e-2)
    (gl2,) = torch.autograd.grad(enc(pts, win.to(dev)), enc.lattice_values, 2 * g)
    assert (gl2 - 2 * gl).abs().max() <= 2e-5 * gl2.abs().max()      # atomic accumulation order differs run to run


@pytest.mark.parametrize("P,F,L_,T", [(3, 2, 16, 2 ** 18), (4, 2, 8, 2 ** 16), (2, 2, 8, 2 ** 14), (3, 4, 8, 2 ** 16)])
def test_binned_backward_equals_atomic_backward(dev, P, F, L_, T):
    """Large batches take the queue + LDS-reduction path (N >= 2^18); it must agree with the plain atomic path
    (same kernels, workspace withheld) and with the oracle on a subset.  Every (pos_dim, features) instantiation: the
    queue-mode kernels have their own LDS cache size and register budget."""
    from permuto_sdf_amd import _lib as L
    from permuto_sdf_amd.encoding import _head, _tail, encode_backward_raw
    N = 600_001
    enc, sl, _, win = _make(P, L_, T, F, 8, seed=44, concat=True)
    enc = enc.to(dev)
    torch.manual_seed(1)
    pts = (torch.rand(N, P, device=dev) - 0.5)
    g = torch.randn(enc.output_dims(), N, device=dev)
    w = win.to(dev)
    cfg = enc.cfg
    gl_q, gp_q = torch.zeros_like(enc.lattice_values), torch.zeros_like(pts)
    encode_backward_raw(cfg, pts, enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), w, g, gl_q, gp_q)
    gl_a, gp_a = torch.zeros_like(enc.lattice_values), torch.zeros_like(pts)
    L.call("psdf_encode_backward", *_head(cfg, N), L.ptr(pts), L.ptr(enc.lattice_values.detach()), L.ptr(enc.scale_factor),
           L.ptr(enc.random_shift_per_level.detach()), L.ptr(w), *_tail(cfg), L.ptr(g), L.ptr(gl_a), L.ptr(gp_a), L.stream())
    scale = gl_a.abs().max().item()
    assert (gl_q - gl_a).abs().max().item() <= 2e-5 * scale
    assert (gp_q - gp_a).abs().max().item() <= 2e-5 * gp_a.abs().max().item()
    # oracle on a slice of the batch: gradient restricted to the first 3000 points
    sub = 3000
    lat = enc.lattice_values.detach().cpu().clone().requires_grad_(True)
    ref = po.encode(pts[:sub].cpu(), lat, sl, enc.random_shift_per_level.detach().cpu(), win, True, 1e-3)
    ref.backward(g[:, :sub].t().cpu())
    gl_s = torch.zeros_like(enc.lattice_values)
    encode_backward_raw(cfg, pts[:sub].contiguous(), enc.lattice_values.detach(), enc.scale_factor,
                        enc.random_shift_per_level.detach(), w, g[:, :sub].contiguous(), gl_s, None)
    assert (gl_s.cpu() - lat.grad).abs().max() <= 1e-5 * lat.grad.abs().max() + 1e-7


@pytest.mark.parametrize("P,F,L_,T,N", [(3, 2, 24, 2 ** 18, 49_152), (3, 2, 16, 2 ** 18, 300_001), (4, 2, 8, 2 ** 16, 40_000),
                                       (2, 2, 8, 2 ** 14, 20_000), (3, 4, 8, 2 ** 16, 30_000)])
def test_binned_double_backward_equals_atomic_double_backward(dev, P, F, L_, T, N):
    """From 2^13 points on, the double backward's lattice scatter goes through the backward's binning + reduce kernels
    (encode_bwd_kernel<.., DBL>); it must agree with the plain kernel (same operator, workspace withheld: LDS cache + float
    atomics, the path the oracle-parity test above exercises) in both outputs -- the lattice gradient up to the order of the float
    additions, the gradient w.r.t. the feature gradient bit for bit (a gather).  Ray-like batch: consecutive points are consecutive
    samples of a ray, as in a training step (that is what the run combine and the LDS cache of the binning kernel see)."""
    from permuto_sdf_amd import _lib as L
    from permuto_sdf_amd.encoding import _head, _tail, encode_double_backward_raw
    enc, sl, _, win = _make(P, L_, T, F, 8, seed=45, concat=True)
    enc = enc.to(dev)
    torch.manual_seed(2)
    R = (N + 95) // 96
    o = torch.rand(R, P, device=dev) - 0.5
    d = torch.nn.functional.normalize(torch.randn(R, P, device=dev), dim=1)
    t = torch.linspace(-0.3, 0.3, 96, device=dev)
    pts = (o[:, None, :] + t[None, :, None] * d[:, None, :]).reshape(-1, P)[:N].contiguous()
    g = torch.randn(enc.output_dims(), N, device=dev)
    u = torch.randn(N, P, device=dev)
    w = win.to(dev)
    cfg = enc.cfg
    lat, sf, sh = enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach()
    gl_q, gg_q = torch.zeros_like(lat), torch.full_like(g, float("nan"))
    encode_double_backward_raw(cfg, pts, lat, sf, sh, w, u, g, gl_q, gg_q)
    gl_a, gg_a = torch.zeros_like(lat), torch.full_like(g, float("nan"))
    L.call("psdf_encode_double_backward", *_head(cfg, N), L.ptr(pts), L.ptr(lat), L.ptr(sf), L.ptr(sh), L.ptr(w), *_tail(cfg),
           L.ptr(u), L.ptr(g), L.ptr(gl_a), L.ptr(gg_a), L.stream())
    assert torch.equal(gg_q, gg_a) and bool(torch.isfinite(gg_q).all())
    scale = gl_a.abs().max().item()
    assert scale > 0 and (gl_q - gl_a).abs().max().item() <= 2e-5 * scale
    # without a lattice gradient buffer (gradient w.r.t. the feature gradient only) the plain kernel runs: same numbers
    gg_n = torch.full_like(g, float("nan"))
    encode_double_backward_raw(cfg, pts, lat, sf, sh, w, u, g, None, gg_n)
    assert torch.equal(gg_n, gg_a)
    # the merged scatter (the training step's form): double-backward scatter + plain backward of a second upstream gradient in
    # one pass, no gathered output == the two operators run one after the other
    from permuto_sdf_amd.encoding import encode_backward_raw
    g2 = torch.randn_like(g)
    gl_m = torch.zeros_like(lat)
    encode_double_backward_raw(cfg, pts, lat, sf, sh, w, u, g, gl_m, None, g2)
    gl_s = gl_a.clone()
    L.call("psdf_encode_backward", *_head(cfg, N), L.ptr(pts), L.ptr(lat), L.ptr(sf), L.ptr(sh), L.ptr(w), *_tail(cfg), L.ptr(g2),
           L.ptr(gl_s), None, L.stream())
    scale2 = gl_s.abs().max().item()
    assert (gl_m - gl_s).abs().max().item() <= 2e-5 * scale2
    # ... and below the binned plan's threshold (plain kernels, scratch for the unwanted gathered output)
    n_small = 3000
    gl_m2, gl_s2 = torch.zeros_like(lat), torch.zeros_like(lat)
    ps, gs, g2s, us = pts[:n_small].contiguous(), g[:, :n_small].contiguous(), g2[:, :n_small].contiguous(), u[:n_small].contiguous()
    encode_double_backward_raw(cfg, ps, lat, sf, sh, w, us, gs, gl_m2, None, g2s)
    gg_tmp = torch.empty_like(gs)
    L.call("psdf_encode_double_backward", *_head(cfg, n_small), L.ptr(ps), L.ptr(lat), L.ptr(sf), L.ptr(sh), L.ptr(w), *_tail(cfg),
           L.ptr(us), L.ptr(gs), L.ptr(gl_s2), L.ptr(gg_tmp), L.stream())
    L.call("psdf_encode_backward", *_head(cfg, n_small), L.ptr(ps), L.ptr(lat), L.ptr(sf), L.ptr(sh), L.ptr(w), *_tail(cfg), L.ptr(g2s),
           L.ptr(gl_s2), None, L.stream())
    assert (gl_m2 - gl_s2).abs().max().item() <= 2e-5 * gl_s2.abs().max().item()
