"""csrc/neus.hip (fused NeuS section-point opacity fwd/bwd, L1 and eikonal loss tails) against the CPU oracle
oracle/neus_oracle.py = the reference's own torch expressions (volume_rendering_modules.py:129-172,
permuto_sdf_utils.py:43-51) evaluated on the CPU, autograd for the backward.  Tolerances: forward 2e-6 abs on values in
[0,1]; gradients 2e-5 relative to the largest entry (expf of the device libm differs from the host's in the last bit)."""
import numpy as np
import pytest
import torch

from oracle import neus_oracle as no

pytestmark = pytest.mark.gpu


def _inputs(N, seed, scale=0.02):
    g = torch.Generator().manual_seed(seed)
    sdf = torch.randn(N, 1, generator=g) * scale
    dirs = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=1)
    grad = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=1) * (0.5 + torch.rand(N, 1, generator=g))
    dt = torch.rand(N, 1, generator=g) * 0.01 + 1e-4
    return sdf, dirs, grad, dt


@pytest.mark.parametrize("ratio", [0.0, 0.37, 1.0])
@pytest.mark.parametrize("variance", [0.3, 0.55, 0.8])
def test_neus_alpha_forward_backward_match_reference_expressions(dev, ratio, variance):
    from permuto_sdf_amd.neus import neus_alpha
    N = 50_000
    sdf, dirs, grad, dt = _inputs(N, 7)
    var = torch.tensor(variance, requires_grad=True)
    sdf_r, grad_r = sdf.clone().requires_grad_(True), grad.clone().requires_grad_(True)
    a_ref, om_ref = no.neus_alpha(sdf_r, dirs, grad_r, dt, no.inv_s(var), ratio)
    up_a, up_om = torch.randn(N, 1), torch.randn(N, 1)
    (a_ref * up_a + om_ref * up_om).sum().backward()
    var_d = torch.tensor(variance, device=dev, requires_grad=True)
    sdf_d, grad_d = sdf.to(dev).requires_grad_(True), grad.to(dev).requires_grad_(True)
    inv_s_d = torch.exp(var_d * 10.0).clip(1e-6, 1e6)
    a, om = neus_alpha(sdf_d, dirs.to(dev), grad_d, dt.to(dev), inv_s_d, ratio)
    assert a.shape == (N, 1) and float(a.min()) >= 0.0 and float(a.max()) <= 1.0
    assert (a.detach().cpu() - a_ref.detach()).abs().max() <= 2e-6
    assert (om.detach().cpu() - om_ref.detach()).abs().max() <= 2e-6
    (a * up_a.to(dev) + om * up_om.to(dev)).sum().backward()
    for got, ref, name in ((sdf_d.grad, sdf_r.grad, "sdf"), (grad_d.grad, grad_r.grad, "gradients")):
        assert (got.cpu() - ref).abs().max() <= 2e-5 * ref.abs().max(), name
    # d/d variance is a SIGNED sum over all samples (wave sums + one float atomic per wave, order not fixed): the tolerance is
    # relative to the sum of the magnitudes of the per-sample terms (from a per-sample inv_s on the CPU), not to the sum itself
    inv_vec = no.inv_s(var).detach().expand(N, 1).clone().requires_grad_(True)
    a2, om2 = no.neus_alpha(sdf, dirs, grad, dt, inv_vec, ratio)
    (a2 * up_a + om2 * up_om).sum().backward()
    chain = 10.0 * float(no.inv_s(var))                        # d inv_s / d variance (inside the clip)
    magnitude = float(inv_vec.grad.abs().sum()) * chain
    assert abs(float(var_d.grad) - float(var.grad)) <= 2e-6 * magnitude + 1e-4 * abs(float(var.grad)), \
        (float(var_d.grad), float(var.grad), magnitude)
    # interior of the clip: a share of the samples is clipped at 0 or 1 in this regime -- both branches are exercised
    q = ((a_ref.detach() == 0) | (a_ref.detach() == 1)).float().mean()
    assert 0.0 <= float(q) < 1.0


def test_neus_alpha_edge_cases(dev):
    from permuto_sdf_amd.neus import neus_alpha_backward_raw, neus_alpha_forward_raw
    inv_s = torch.tensor([64.0], device=dev)
    # empty batch
    z1, z3 = torch.zeros(0, 1, device=dev), torch.zeros(0, 3, device=dev)
    a, om = neus_alpha_forward_raw(z1, z3, z3, z1, inv_s, 1.0)
    assert a.shape == (0, 1) and om.shape == (0, 1)
    # saturated samples: far inside (sdf << 0): c -> 0, p -> 0: alpha = 1e-5 / 1e-5 = 1; far outside: alpha -> 0
    sdf = torch.tensor([[-10.0], [10.0], [0.0]], device=dev)
    d = torch.tensor([[0.0, 0.0, 1.0]], device=dev).repeat(3, 1)
    n = -d
    dt = torch.full((3, 1), 0.01, device=dev)
    a, om = neus_alpha_forward_raw(sdf, d, n, dt, inv_s, 1.0)
    ref, _ = no.neus_alpha(sdf.cpu(), d.cpu(), n.cpu(), dt.cpu(), inv_s.cpu(), 1.0)
    assert torch.allclose(a.cpu(), ref, atol=2e-6)
    assert float(a[0]) == 1.0 and float(a[1]) < 1e-4
    g_sdf, g_n, g_s = neus_alpha_backward_raw(torch.ones(3, 1, device=dev), sdf, d, n, dt, inv_s, 1.0)
    assert torch.isfinite(g_sdf).all() and torch.isfinite(g_n).all() and torch.isfinite(g_s).all()


def test_loss_tails(dev):
    from permuto_sdf_amd.neus import eikonal_loss_raw, l1_loss, l1_loss_raw
    g = torch.Generator().manual_seed(1)
    R = 3001
    pred, gt = torch.rand(R, 3, generator=g), torch.rand(R, 3, generator=g)
    hit = torch.rand(R, 1, generator=g) > 0.2
    p = pred.clone().requires_grad_(True)
    ref = no.rgb_loss(gt, p, hit)
    ref.backward()
    loss, gp = l1_loss_raw(pred.to(dev), gt.to(dev), hit.to(dev))
    assert abs(float(loss) - float(ref)) <= 1e-5 * float(ref)
    assert torch.equal(gp.cpu(), p.grad)                       # +-1/(3R) or 0: exact
    pd = pred.to(dev).requires_grad_(True)
    (l1_loss(pd, gt.to(dev), hit.to(dev)) * 2.0).backward()
    assert torch.allclose(pd.grad.cpu(), 2.0 * p.grad, rtol=1e-6, atol=0)
    grads = torch.randn(5000, 3, generator=g) * 1.3
    gr = grads.clone().requires_grad_(True)
    e_ref = no.eikonal_loss(gr)
    e_ref.backward()
    e, ge = eikonal_loss_raw(grads.to(dev))
    assert abs(float(e) - float(e_ref)) <= 1e-5 * float(e_ref)
    assert (ge.cpu() - gr.grad).abs().max() <= 1e-6 * gr.grad.abs().max()


def test_hot_path_true_gradient_matches_oracle_autograd(dev):
    """the whole first-order step (what bench.py times, at a size the CPU oracle finishes in seconds): gradients of the
    L1 radiance loss w.r.t. lattice and every MLP parameter against torch autograd through the oracles"""
    import __graft_entry__ as ge
    assert ge.smoke()


# ---- second block of csrc/neus.hip: the remaining elementwise chains of the training step, against the torch expressions the
# reference writes (float64 on the CPU, autograd for the gradients)
def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_normalize3_and_curvature_terms_match_torch(dev):
    import math
    import torch.nn.functional as F
    from permuto_sdf_amd.neus import curvature_loss, curvature_shift, normalize3
    g = torch.Generator().manual_seed(5)
    N = 5001
    x = torch.randn(N, 3, generator=g) * torch.rand(N, 1, generator=g) * 3
    x[7] = 0.0                                   # |x| below eps: the clamp of F.normalize is active
    g2 = x + 0.3 * torch.randn(N, 3, generator=g)
    g2[11] = x[11]                               # dot = 1: clamped, zero gradient
    g2[12] = -x[12]                              # dot = -1
    pts = torch.rand(N, 3, generator=g) - 0.5
    rnd = torch.randn(N, 3, generator=g)
    up = torch.randn(N, 3, generator=g)

    # reference expressions (models.py:266-289), float64
    xd, g2d = x.double().requires_grad_(True), g2.double().requires_grad_(True)
    n_ref = F.normalize(xd, dim=-1)
    (gx_ref,) = torch.autograd.grad(n_ref, xd, up.double(), retain_graph=True)
    shifted_ref = pts.double() + torch.cross(n_ref, F.normalize(rnd.double(), dim=-1), dim=-1) * 1e-4
    (gs_ref,) = torch.autograd.grad(shifted_ref, xd, up.double(), retain_graph=True)
    dot = (n_ref * F.normalize(g2d, dim=-1)).sum(-1, keepdim=True)
    curv_ref = (torch.acos(torch.clamp(dot, -1.0 + 1e-6, 1.0 - 1e-6)) / math.pi).mean()
    ga_ref, gb_ref = torch.autograd.grad(curv_ref, [xd, g2d])

    xg, g2g = x.to(dev).requires_grad_(True), g2.to(dev).requires_grad_(True)
    n = normalize3(xg)
    assert _rel(n, n_ref) < 1e-6
    (gx,) = torch.autograd.grad(n, xg, up.to(dev))
    ok = torch.ones(N, dtype=torch.bool); ok[7] = False        # the zero vector: gradient 1/eps = 1e12 either way
    assert _rel(gx.cpu()[ok], gx_ref[ok]) < 1e-5
    sh = curvature_shift(pts.to(dev), xg, rnd.to(dev), 1e-4)
    assert float((sh.cpu().double() - shifted_ref.detach()).abs().max()) < 1e-7
    (gs,) = torch.autograd.grad(sh, xg, up.to(dev))
    assert _rel(gs.cpu()[ok], gs_ref[ok]) < 1e-5
    curv = curvature_loss(xg, g2g)
    # acos is ill-conditioned near +-1 (the clamped rows): compare the mean with an absolute tolerance
    assert abs(float(curv) - float(curv_ref)) < 2e-6
    ga, gb = torch.autograd.grad(curv, [xg, g2g])
    keep = ok.clone(); keep[11] = keep[12] = False
    d = dot.detach().view(-1).abs()
    keep &= d < 0.999                            # fp32 cancellation in 1 - u^2 next to the clamp; the rest to 1e-4
    assert _rel(ga.cpu()[keep], ga_ref[keep]) < 1e-4 and _rel(gb.cpu()[keep], gb_ref[keep]) < 1e-4
    assert float(ga[11].abs().max()) == 0.0 and float(gb[12].abs().max()) == 0.0   # clamped: no gradient


def test_offsurface_and_nerf_alpha_match_torch(dev):
    import torch.nn.functional as F
    from permuto_sdf_amd.neus import eikonal_loss, nerf_alpha, offsurface_loss
    g = torch.Generator().manual_seed(6)
    N = 4097
    s = torch.randn(N, 1, generator=g) * 0.02
    s[3] = 0.0
    sd = s.double().requires_grad_(True)
    ref = torch.exp(-1e2 * sd.abs()).mean()                      # train_permuto_sdf.py:372-375
    (gref,) = torch.autograd.grad(ref, sd)
    sg = s.to(dev).requires_grad_(True)
    out = offsurface_loss(sg, 1e2)
    assert abs(float(out) - float(ref)) < 1e-6
    (gg,) = torch.autograd.grad(out, sg)
    assert gg.shape == sg.shape and _rel(gg, gref) < 1e-5

    raw = torch.randn(N, 1, generator=g) * 6
    raw[5] = 25.0                                                # beyond softplus' linear threshold
    dt = torch.rand(N, 1, generator=g) * 0.05
    dt[9] = 1e10                                                 # the last background sample (RaySamplerGPU.cuh:150)
    rd = raw.double().requires_grad_(True)
    a_ref = 1.0 - torch.exp(-F.softplus(rd) * dt.double())       # models.py:520, volume_rendering_modules.py:72-86
    om_ref = 1 - a_ref + 1e-7
    ua, uo = torch.randn(N, 1, generator=g).double(), torch.randn(N, 1, generator=g).double()
    (gr_ref,) = torch.autograd.grad((a_ref * ua).sum() + (om_ref * uo).sum(), rd)
    rg = raw.to(dev).requires_grad_(True)
    a, om = nerf_alpha(rg, dt.to(dev))
    assert a.shape == (N, 1) and _rel(a, a_ref) < 1e-6 and _rel(om, om_ref) < 1e-6
    (gr,) = torch.autograd.grad((a * ua.float().to(dev)).sum() + (om * uo.float().to(dev)).sum(), rg)
    assert gr.shape == rg.shape and _rel(gr, gr_ref) < 1e-5

    gr3 = torch.randn(N, 3, generator=g)
    g3 = gr3.double().requires_grad_(True)
    e_ref = ((g3.norm(dim=-1) - 1.0) ** 2).mean()                # permuto_sdf_utils.py:49-51
    (ge_ref,) = torch.autograd.grad(e_ref, g3)
    gd = gr3.to(dev).requires_grad_(True)
    e = eikonal_loss(gd)
    (ge,) = torch.autograd.grad(e * 0.04, gd)
    assert abs(float(e) - float(e_ref)) < 1e-5 and _rel(ge, ge_ref * 0.04) < 1e-5


@pytest.mark.parametrize("mode", ["equal128", "equal48", "ragged", "holes"])
def test_fused_compositing_equals_the_operator_chain_and_the_oracle(dev, mode):
    """csrc/composite_fused.hip: opacity -> transmittance -> weights -> radiance in one launch and its backward in one launch,
    against (1) the chain of per-operator kernels it fuses (neus_alpha -> cumprod -> alpha T -> integrate, and their backward
    kernels as hotpath.py / train_step.py string them), bit for bit where the summation orders coincide (ray lengths that are
    multiples of 64) and within 2e-6 elsewhere, and (2) the reference's own torch expressions (oracle/neus_oracle.py, autograd)
    for equal-count rays.  Containers: equal counts, ragged counts up to 200 samples, and rays the reference skips (empty)."""
    from permuto_sdf import RaySamplesPacked, VolumeRendering as VR
    from permuto_sdf_amd.neus import (neus_alpha_backward_raw, neus_alpha_forward_raw, neus_composite_backward_raw,
                                      neus_composite_forward_raw)
    g = torch.Generator().manual_seed({"equal128": 1, "equal48": 2, "ragged": 3, "holes": 4}[mode])
    R = 300
    if mode.startswith("equal"):
        per = int(mode[5:])
        counts = torch.full((R,), per, dtype=torch.int64)
    else:
        counts = torch.randint(1, 200, (R,), generator=g)
        if mode == "holes":
            counts[torch.randperm(R, generator=g)[:40]] = 0
    ends = torch.cumsum(counts, 0)
    starts = ends - counts
    N = int(ends[-1])
    sdf, dirs, grad, dt = _inputs(N, 11, scale=0.01)
    rgb = torch.rand(N, 3, generator=g)
    rs = RaySamplesPacked(R, N, device=dev)
    if mode.startswith("equal"):
        rs.rays_have_equal_nr_of_samples, rs.fixed_nr_of_samples_per_ray = True, per
    rs.ray_start_end_idx = torch.stack([starts, ends], 1).to(torch.int32).to(dev)
    rs.samples_dirs, rs.samples_dt = dirs.to(dev), dt.to(dev)
    rs.cur_nr_samples.fill_(N)
    sdf_d, grad_d, rgb_d = sdf.to(dev), grad.to(dev), rgb.to(dev)
    inv_s = torch.tensor([300.0], device=dev)
    ratio = 0.6
    g_pred = torch.randn(R, 3, generator=g).to(dev)
    g_bg = torch.randn(R, 1, generator=g).to(dev)
    # ---- fused
    pred, bg, w = neus_composite_forward_raw(rs, sdf_d, grad_d, rgb_d, inv_s, ratio, want_weights=True)
    gs, gg, gr, gi = neus_composite_backward_raw(rs, 200 if not mode.startswith("equal") else per, g_pred, g_bg, sdf_d, grad_d, rgb_d,
                                                 inv_s, ratio)
    # ---- the operator chain
    alpha, om = neus_alpha_forward_raw(sdf_d, rs.samples_dirs, grad_d, rs.samples_dt, inv_s, ratio)
    T, bg2 = VR.cumprod_alpha2transmittance(rs, om)
    w2 = alpha * T
    pred2 = VR.integrate_with_weights(rs, rgb_d, w2)
    g_rgb2, g_w2 = VR.integrate_with_weights_backward(g_pred, rs, rgb_d, w2, None)
    g_T = g_w2 * alpha
    cs = VR.cumsum_over_each_ray(rs, g_T * T, True)
    g_om = VR.cumprod_alpha2transmittance_backward(g_T, g_bg, rs, om, T, bg2, cs)
    g_alpha = torch.addcmul(-g_om, g_w2, T)
    gs2, gg2, gi2 = neus_alpha_backward_raw(g_alpha, sdf_d, rs.samples_dirs, grad_d, rs.samples_dt, inv_s, ratio)
    live = (counts > 0).to(dev)
    assert torch.equal(pred[live], pred2[live]) and torch.equal(bg[live], bg2[live])          # forward: identical order
    assert torch.equal(w, w2) and torch.equal(gr, g_rgb2)
    tol = 0.0 if mode == "equal128" else 2e-6
    for a, b, name in ((gs, gs2, "g_sdf"), (gg, gg2, "g_gradients")):
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()) + (0 if tol == 0 else 1e-12), (name, float((a - b).abs().max()))
    assert abs(float(gi) - float(gi2)) <= 2e-5 * float(g_alpha.abs().sum()) + 1e-6 * abs(float(gi2))   # an atomic sum: order free
    # ---- the reference's expressions (equal counts: oracle/neus_oracle.composite_equal)
    if mode.startswith("equal"):
        sdf_r, grad_r, rgb_r = sdf.clone().requires_grad_(True), grad.clone().requires_grad_(True), rgb.clone().requires_grad_(True)
        a_ref, om_ref = no.neus_alpha(sdf_r, dirs, grad_r, dt, torch.tensor(300.0), ratio)
        pred_ref, w_ref, T_ref = no.composite_equal(a_ref, om_ref, rgb_r, R, per, reference_compat=VR.reference_compat)
        bg_ref = (T_ref.view(R, per)[:, -1:])                               # bg transmittance == T of the last sample
        ((pred_ref * g_pred.cpu()).sum() + (bg_ref * g_bg.cpu()).sum()).backward()
        assert (pred.cpu() - pred_ref.detach()).abs().max() <= 2e-6 * max(1.0, float(pred_ref.abs().max()))
        assert (gs.cpu() - sdf_r.grad).abs().max() <= 3e-5 * sdf_r.grad.abs().max()
        assert (gg.cpu() - grad_r.grad).abs().max() <= 3e-5 * grad_r.grad.abs().max()
        assert (gr.cpu() - rgb_r.grad).abs().max() <= 3e-5 * rgb_r.grad.abs().max()
    # more than 256 samples per ray: the fused backward says so (-2) and the caller uses the operator chain
    from permuto_sdf_amd._lib import PsdfError
    with pytest.raises(PsdfError):
        neus_composite_backward_raw(rs, 300, g_pred, g_bg, sdf_d, grad_d, rgb_d, inv_s, ratio)


@pytest.mark.parametrize("N,C", [(1, 3), (5000, 3), (70001, 4)])
def test_sigmoid_rows_and_backward(dev, N, C):
    """psdf_sigmoid_rows / _backward (the colour heads' sigmoid fused with the [C, N] <-> [N, C] layout change) against torch"""
    from permuto_sdf_amd.neus import sigmoid_rows_backward_raw, sigmoid_rows_raw
    torch.manual_seed(N)
    x = torch.randn(C, N, device=dev) * 4
    y = sigmoid_rows_raw(x)
    ref = torch.sigmoid(x.t())
    assert y.shape == (N, C) and float((y - ref).abs().max()) <= 2e-7
    g = torch.randn(N, C, device=dev)
    gx = sigmoid_rows_backward_raw(g, y)
    assert gx.shape == (C, N) and float((gx - (g * y * (1 - y)).t()).abs().max()) <= 1e-7 * float(g.abs().max())


@pytest.mark.parametrize("layout", ["equal32", "packed"])
def test_nerf_composite_equals_the_operator_chain(dev, layout):
    """psdf_nerf_composite_forward / _backward (background NeRF rendering + composition with the foreground, one launch per
    direction) against the chain of operators they replace: nerf_alpha -> cumprod_alpha2transmittance -> alpha * T ->
    integrate_with_weights -> pred_fg + bgT * pred_bg, and its backward through the reference's autograd Functions' kernels."""
    from permuto_sdf import RaySamplesPacked, VolumeRendering as VR
    from permuto_sdf_amd import _lib as L
    from permuto_sdf_amd.neus import nerf_composite_backward_raw, nerf_composite_forward_raw
    torch.manual_seed(5)
    R = 300
    if layout == "equal32":
        counts = torch.full((R,), 32, dtype=torch.int64)
    else:
        counts = torch.randint(0, 150, (R,))
        counts[:3] = torch.tensor([0, 1, 2])
    start = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)])
    M = int(start[-1])
    rs = RaySamplesPacked(R, M)
    rs.ray_start_end_idx = torch.stack([start[:-1], start[1:]], 1).to(torch.int32).to(dev)
    if layout == "equal32":
        rs.rays_have_equal_nr_of_samples, rs.fixed_nr_of_samples_per_ray = True, 32
    rs.samples_dt = (torch.rand(M, 1, device=dev) * 0.05 + 1e-3)
    rs.cur_nr_samples.fill_(M)
    rs._exact = True
    rs._dense = True
    raw = torch.randn(M, device=dev) * 3
    rgb = torch.rand(M, 3, device=dev)
    fg_pred, fg_bg = torch.rand(R, 3, device=dev), torch.rand(R, 1, device=dev)
    g_pred = torch.randn(R, 3, device=dev)

    def chain():
        dtv = rs.samples_dt.reshape(-1).contiguous()
        alpha, om = torch.empty_like(raw), torch.empty_like(raw)
        L.call("psdf_nerf_alpha_forward", L.c_l(M), L.ptr(raw), L.ptr(dtv), L.ptr(alpha), L.ptr(om), L.stream())
        T, bgT = VR.cumprod_alpha2transmittance(rs, om.view(-1, 1))
        w = alpha.view(-1, 1) * T
        pred_bg = VR.integrate_with_weights(rs, rgb, w)
        pred = fg_pred + fg_bg * pred_bg
        g_fg = (g_pred * pred_bg).sum(1, keepdim=True)
        g_rgb, g_w = VR.integrate_with_weights_backward((fg_bg * g_pred).contiguous(), rs, rgb, w, None)
        g_T = g_w * alpha.view(-1, 1)
        cs = VR.cumsum_over_each_ray(rs, g_T * T, True)
        g_om = VR.cumprod_alpha2transmittance_backward(g_T, torch.zeros_like(bgT), rs, om.view(-1, 1), T, bgT, cs)
        g_raw = torch.empty_like(raw)
        L.call("psdf_nerf_alpha_backward", L.c_l(M), L.ptr(raw), L.ptr(dtv), L.ptr((g_w * T).reshape(-1).contiguous()),
               L.ptr(g_om.reshape(-1).contiguous()), L.ptr(g_raw), L.stream())
        return pred_bg, pred, g_raw, g_rgb, g_fg
    pb, p, gr, gc, gf = chain()
    pb2, p2 = nerf_composite_forward_raw(rs, raw, rgb, fg_pred, fg_bg)
    gr2, gc2, gf2 = nerf_composite_backward_raw(rs, int(counts.max()), g_pred, raw, rgb, fg_bg)

    def close(a, b, tol=2e-6):
        assert a.shape == b.shape and float((a - b).abs().max()) <= tol * max(1.0, float(a.abs().max())), float((a - b).abs().max())
    close(pb, pb2)
    close(p, p2)
    close(gf, gf2, 5e-6)
    close(gc, gc2)
    close(gr, gr2, 1e-5)
    # alone (no foreground): gradient of the background radiance itself
    pb3, none = nerf_composite_forward_raw(rs, raw, rgb)
    assert none is None
    close(pb, pb3)


def test_fused_backward_poisons_rays_longer_than_declared(dev):
    """`max_per_ray` is the caller's promise.  A ray that holds more samples than declared used to lose its tail
    silently (gradients of the dropped samples never written); now every gradient of such a ray is NaN -- loud -- and the rays
    that keep the promise are untouched."""
    from permuto_sdf import RaySamplesPacked
    from permuto_sdf_amd.neus import (nerf_composite_backward_raw, nerf_composite_forward_raw, neus_composite_backward_raw,
                                      neus_composite_forward_raw)
    g = torch.Generator().manual_seed(9)
    counts = torch.tensor([40, 100, 64, 7])              # declared: at most 64 per ray -> ray 1 breaks the promise
    ends = torch.cumsum(counts, 0)
    starts = ends - counts
    R, N = 4, int(ends[-1])
    sdf, dirs, grad, dt = _inputs(N, 5, scale=0.01)
    rgb = torch.rand(N, 3, generator=g).to(dev)
    rs = RaySamplesPacked(R, N, device=dev)
    rs.ray_start_end_idx = torch.stack([starts, ends], 1).to(torch.int32).to(dev)
    rs.samples_dirs, rs.samples_dt = dirs.to(dev), dt.to(dev)
    rs.cur_nr_samples.fill_(N)
    inv_s = torch.tensor([300.0], device=dev)
    sdf_d, grad_d = sdf.to(dev), grad.to(dev)
    g_pred, g_bg = torch.randn(R, 3, generator=g).to(dev), torch.randn(R, 1, generator=g).to(dev)
    ok = neus_composite_backward_raw(rs, 128, g_pred, g_bg, sdf_d, grad_d, rgb, inv_s, 0.6, need_inv_s=False)      # honest bound
    bad = neus_composite_backward_raw(rs, 64, g_pred, g_bg, sdf_d, grad_d, rgb, inv_s, 0.6, need_inv_s=False)
    lo, hi = int(starts[1]), int(ends[1])
    for a, b in zip(ok[:3], bad[:3]):
        assert torch.isfinite(a).all()
        assert torch.isnan(b[lo:hi]).all()
        keep = torch.ones(N, dtype=torch.bool, device=dev)
        keep[lo:hi] = False
        assert torch.equal(a[keep], b[keep])
    raw = torch.randn(N, generator=g).to(dev)
    fg_pred, fg_bg = torch.rand(R, 3, generator=g).to(dev), torch.rand(R, 1, generator=g).to(dev)
    ok = nerf_composite_backward_raw(rs, 128, g_pred, raw, rgb, fg_bg)
    bad = nerf_composite_backward_raw(rs, 64, g_pred, raw, rgb, fg_bg)
    assert torch.isfinite(ok[0]).all() and torch.isnan(bad[0][lo:hi]).all() and torch.isnan(bad[2][1]).all()
    assert torch.equal(ok[0][:lo], bad[0][:lo]) and torch.equal(ok[2][[0, 2, 3]], bad[2][[0, 2, 3]])
