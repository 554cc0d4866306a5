"""Host side of csrc/neus.hip: the NeuS section-point opacity as ONE differentiable operator, and the fused loss tails.

`neus_alpha(sdf, dirs, gradients, dt, inv_s, cos_anneal_ratio)` computes what `VolumeRenderingNeus.compute_weights` of the
reference (permuto_sdf_py/volume_rendering/volume_rendering_modules.py:129-172) computes between `inv_s` and `alpha` --
about 30 torch elementwise launches per direction there, one kernel per direction here -- and returns
(alpha [N,1], 1 - alpha + 1e-7 [N,1]); gradients flow to `sdf`, `gradients` and `inv_s`."""
import torch

from . import _lib as L


def neus_alpha_forward_raw(sdf, dirs, gradients, dt, inv_s, cos_anneal_ratio):
    N = sdf.shape[0]
    L.require_cuda(sdf, dirs, gradients, dt, inv_s)
    alpha = torch.empty((N, 1), dtype=torch.float32, device=sdf.device)
    one_minus = torch.empty((N, 1), dtype=torch.float32, device=sdf.device)
    L.call("psdf_neus_alpha_forward", L.c_l(N), L.ptr(sdf), L.ptr(dirs), L.ptr(gradients), L.ptr(dt), L.ptr(inv_s),
           L.c_f(float(cos_anneal_ratio)), L.ptr(alpha), L.ptr(one_minus), L.stream())
    return alpha, one_minus


def neus_alpha_backward_raw(g_alpha, sdf, dirs, gradients, dt, inv_s, cos_anneal_ratio, need_grad=True, need_inv_s=True):
    N = sdf.shape[0]
    g_sdf = torch.empty((N, 1), dtype=torch.float32, device=sdf.device)
    g_grad = torch.empty((N, 3), dtype=torch.float32, device=sdf.device) if need_grad else None
    g_inv_s = L.zeroed_scalar(sdf.device) if need_inv_s else None
    L.call("psdf_neus_alpha_backward", L.c_l(N), L.ptr(g_alpha), L.ptr(sdf), L.ptr(dirs), L.ptr(gradients), L.ptr(dt),
           L.ptr(inv_s), L.c_f(float(cos_anneal_ratio)), L.ptr(g_sdf), L.ptr(g_grad), L.ptr(g_inv_s), L.stream())
    return g_sdf, g_grad, g_inv_s


def _c(t):
    return t.detach().to(torch.float32).contiguous()


class NeusAlphaFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sdf, dirs, gradients, dt, inv_s, cos_anneal_ratio):
        sdf_c, dirs_c, grad_c, dt_c = _c(sdf).view(-1, 1), _c(dirs), _c(gradients), _c(dt).view(-1, 1)
        inv_s_c = _c(inv_s).view(1)
        alpha, one_minus = neus_alpha_forward_raw(sdf_c, dirs_c, grad_c, dt_c, inv_s_c, cos_anneal_ratio)
        ctx.save_for_backward(sdf_c, dirs_c, grad_c, dt_c, inv_s_c)
        ctx.r = float(cos_anneal_ratio)
        ctx.inv_s_shape = inv_s.shape
        return alpha, one_minus

    @staticmethod
    def backward(ctx, g_alpha, g_one_minus):
        sdf, dirs, grad, dt, inv_s = ctx.saved_tensors
        g = g_alpha if g_one_minus is None else (g_alpha - g_one_minus if g_alpha is not None else -g_one_minus)
        g_sdf, g_grad, g_inv_s = neus_alpha_backward_raw(g.contiguous(), sdf, dirs, grad, dt, inv_s, ctx.r,
                                                          need_grad=ctx.needs_input_grad[2],
                                                          need_inv_s=ctx.needs_input_grad[4])
        return (g_sdf if ctx.needs_input_grad[0] else None, None, g_grad, None,
                g_inv_s.view(ctx.inv_s_shape) if g_inv_s is not None else None, None)


def neus_alpha(sdf, dirs, gradients, dt, inv_s, cos_anneal_ratio):
    return NeusAlphaFunc.apply(sdf, dirs, gradients, dt, inv_s, cos_anneal_ratio)


# ---------------------------------------------------------------------------------------------- fused NeuS rendering
def neus_composite_forward_raw(rs, sdf, gradients, rgb, inv_s, cos_anneal_ratio, want_weights=False):
    """csrc/composite_fused.hip: opacity -> transmittance -> weights -> radiance of a packed container in one launch.
    -> pred [R,3], bg transmittance [R,1], weights [N,1] or None"""
    R, N, dev = rs.ray_start_end_idx.shape[0], sdf.shape[0], sdf.device
    L.require_cuda(sdf, gradients, rgb, inv_s)
    pred = torch.empty((R, 3), dtype=torch.float32, device=dev)       # the kernel writes every ray (0 / 1 for empty ones)
    bg = torch.empty((R, 1), dtype=torch.float32, device=dev)
    w = torch.zeros((N, 1), dtype=torch.float32, device=dev) if want_weights else None
    L.call("psdf_neus_composite_forward", *rs._ri(), L.ptr(sdf), L.ptr(rs.samples_dirs), L.ptr(gradients), L.ptr(rs.samples_dt),
           L.ptr(rgb), L.ptr(inv_s), L.c_f(float(cos_anneal_ratio)), L.ptr(pred), L.ptr(bg), L.ptr(w), L.stream())
    return pred, bg, w


def neus_composite_backward_raw(rs, max_per_ray, g_pred, g_bg, sdf, gradients, rgb, inv_s, cos_anneal_ratio, need_grad=True,
                                need_rgb=True, need_inv_s=True):
    """-> g_sdf [N,1], g_gradients [N,3] | None, g_rgb [N,3] | None, g_inv_s [1] | None; PsdfError(-2) when a ray may hold more
    than 256 samples (callers then use the per-operator chain)"""
    from .bridge import VolumeRendering, _per_sample
    N, dev = sdf.shape[0], sdf.device
    g_sdf = _per_sample(rs, (N, 1), dev)
    g_grad = _per_sample(rs, (N, 3), dev) if need_grad else None
    g_rgb = _per_sample(rs, (N, 3), dev) if need_rgb else None
    g_inv_s = L.zeroed_scalar(dev) if need_inv_s else None
    L.call("psdf_neus_composite_backward", *rs._ri(), L.c_i(int(max_per_ray)), L.ptr(g_pred), L.ptr(g_bg), L.ptr(sdf),
           L.ptr(rs.samples_dirs), L.ptr(gradients), L.ptr(rs.samples_dt), L.ptr(rgb), L.ptr(inv_s),
           L.c_f(float(cos_anneal_ratio)), L.c_i(int(VolumeRendering.reference_compat)), L.ptr(g_sdf), L.ptr(g_grad),
           L.ptr(g_rgb), L.ptr(g_inv_s), L.stream())
    return g_sdf, g_grad, g_rgb, g_inv_s


FUSED_MAX_PER_RAY = 256


class NeusCompositeFunc(torch.autograd.Function):
    """(sdf, gradients, rgb, inv_s) -> (pred [R,3], bg transmittance [R,1]): VolumeRenderingNeus.compute_weights + integrate
    (volume_rendering_modules.py:129-190) as one launch per direction; gradients flow to all four inputs."""

    @staticmethod
    def forward(ctx, rs, max_per_ray, sdf, gradients, rgb, inv_s, cos_anneal_ratio):
        sdf_c, grad_c, rgb_c, inv_s_c = _c(sdf).view(-1, 1), _c(gradients), _c(rgb), _c(inv_s).view(1)
        pred, bg, _ = neus_composite_forward_raw(rs, sdf_c, grad_c, rgb_c, inv_s_c, cos_anneal_ratio)
        ctx.save_for_backward(sdf_c, grad_c, rgb_c, inv_s_c)
        ctx.rs, ctx.max_per_ray, ctx.r, ctx.inv_s_shape = rs, int(max_per_ray), float(cos_anneal_ratio), inv_s.shape
        return pred, bg

    @staticmethod
    def backward(ctx, g_pred, g_bg):
        sdf, grad, rgb, inv_s = ctx.saved_tensors
        need = ctx.needs_input_grad
        g_sdf, g_grad, g_rgb, g_inv_s = neus_composite_backward_raw(
            ctx.rs, ctx.max_per_ray, g_pred.contiguous(), None if g_bg is None else g_bg.contiguous(), sdf, grad, rgb, inv_s, ctx.r,
            need_grad=need[3], need_rgb=need[4], need_inv_s=need[5])
        return (None, None, g_sdf if need[2] else None, g_grad, g_rgb,
                g_inv_s.view(ctx.inv_s_shape) if g_inv_s is not None else None, None)


def neus_composite(rs, max_per_ray, sdf, gradients, rgb, inv_s, cos_anneal_ratio):
    """fused NeuS rendering of a packed container whose rays hold at most `max_per_ray` (<= 256) samples"""
    return NeusCompositeFunc.apply(rs, max_per_ray, sdf, gradients, rgb, inv_s, cos_anneal_ratio)


def l1_loss_raw(pred, gt, mask=None, scale=None, want_grad=True, loss=None):
    """-> (loss [1], g_pred or None): loss = scale * sum |gt - pred| * mask; scale defaults to 1/numel (the reference's mean).
    `loss`: an accumulator to ADD to (the kernels add with one atomic per workgroup) instead of a fresh zero"""
    L.require_cuda(pred, gt)
    R, C = pred.shape
    scale = 1.0 / max(1, R * C) if scale is None else float(scale)
    loss = L.zeroed_scalar(pred.device) if loss is None else loss
    g = torch.empty_like(pred) if want_grad else None
    m = None if mask is None else mask.reshape(-1).to(torch.uint8).contiguous()
    L.call("psdf_l1_loss", L.c_l(R), L.c_i(C), L.ptr(pred.contiguous()), L.ptr(gt.contiguous()), L.ptr(m), L.c_f(scale),
           L.ptr(loss), L.ptr(g), L.stream())
    return loss, g


def eikonal_loss_raw(gradients, scale=None, want_grad=True, loss=None):
    N = gradients.shape[0]
    scale = 1.0 / max(1, N) if scale is None else float(scale)
    loss = L.zeroed_scalar(gradients.device) if loss is None else loss
    g = torch.empty_like(gradients) if want_grad else None
    L.call("psdf_eikonal_loss", L.c_l(N), L.ptr(gradients.contiguous()), L.c_f(scale), L.ptr(loss), L.ptr(g), L.stream())
    return loss, g


class _L1Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, mask):
        loss, g = l1_loss_raw(pred.detach(), gt.detach(), mask)
        ctx.save_for_backward(g)
        return loss.view(())

    @staticmethod
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        return g * go, None, None


class _EikonalLoss(torch.autograd.Function):
    """first-order only: the loss value is exact, its gradient w.r.t. `gradients` is returned as a constant (use the torch
    expression when the eikonal term itself must be differentiated twice)"""

    @staticmethod
    def forward(ctx, gradients):
        loss, g = eikonal_loss_raw(gradients.detach())
        ctx.save_for_backward(g)
        return loss.view(())

    @staticmethod
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        return g * go


def l1_loss(pred, gt, mask=None):
    """((gt - pred).abs() * mask).mean() -- rgb_loss of permuto_sdf_py/utils/permuto_sdf_utils.py:43-47"""
    return _L1Loss.apply(pred, gt, mask)


def eikonal_loss(gradients):
    """((gradients.norm(dim=-1) - 1) ** 2).mean() -- eikonal_loss of permuto_sdf_py/utils/permuto_sdf_utils.py:49-51 (first-order
    autograd: one launch forward, one multiply backward)"""
    return _EikonalLoss.apply(gradients)


# ---- more fused elementwise chains of the training step (csrc/neus.hip, second block); all first-order autograd
def _c3(t):
    return _c(t).view(-1, 3)


class _Normalize3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c3(x)
        y = torch.empty_like(x)
        L.call("psdf_normalize3", L.c_l(x.shape[0]), L.ptr(x), None, L.ptr(y), L.stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gx = torch.empty_like(x)
        L.call("psdf_normalize3", L.c_l(x.shape[0]), L.ptr(x), L.ptr(_c3(gy)), L.ptr(gx), L.stream())
        return gx


def normalize3(x):
    """F.normalize(x, dim=-1) for [N,3] (models.py:272,280,367)"""
    return _Normalize3.apply(x)


class _CurvatureShift(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, gradients, rand_directions, eps):
        p, g, r = _c3(points), _c3(gradients), _c3(rand_directions)
        out = torch.empty_like(p)
        L.call("psdf_curvature_shift", L.c_l(p.shape[0]), L.ptr(p), L.ptr(g), L.ptr(r), L.c_f(eps), None, L.ptr(out), L.stream())
        ctx.save_for_backward(g, r)
        ctx.eps = float(eps)
        return out

    @staticmethod
    def backward(ctx, gs):
        g, r = ctx.saved_tensors
        gg = torch.empty_like(g)
        L.call("psdf_curvature_shift", L.c_l(g.shape[0]), None, L.ptr(g), L.ptr(r), L.c_f(ctx.eps), L.ptr(_c3(gs)), L.ptr(gg),
               L.stream())
        return None, gg, None, None


def curvature_shift(points, gradients, rand_directions, eps=1e-4):
    """points + eps * cross(normalize(gradients), normalize(rand_directions)) -- models.py:266-277; the gradient flows to
    `gradients` only (as in the reference, where points and the random directions are constants)"""
    return _CurvatureShift.apply(points, gradients, rand_directions, eps)


class _CurvatureLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c3(a), _c3(b)
        N = a.shape[0]
        loss = L.zeroed_scalar(a.device)
        ga, gb = torch.empty_like(a), torch.empty_like(b)
        L.call("psdf_curvature_loss", L.c_l(N), L.ptr(a), L.ptr(b), L.c_f(1.0 / max(1, N)), L.ptr(loss), L.ptr(ga), L.ptr(gb),
               L.stream())
        ctx.save_for_backward(ga, gb)
        return loss.view(())

    @staticmethod
    def backward(ctx, go):
        ga, gb = ctx.saved_tensors
        return ga * go, gb * go


def curvature_loss(gradients, gradients_shifted):
    """mean of acos(clamp(n(g) . n(g_shifted), -1+1e-6, 1-1e-6)) / pi -- models.py:280-289 and the .mean() of
    train_permuto_sdf.py:363"""
    return _CurvatureLoss.apply(gradients, gradients_shifted)


class _OffsurfaceLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sdf, sharpness):
        s = _c(sdf).reshape(-1)
        N = s.shape[0]
        loss = L.zeroed_scalar(s.device)
        g = torch.empty_like(s)
        L.call("psdf_offsurface_loss", L.c_l(N), L.ptr(s), L.c_f(sharpness), L.c_f(1.0 / max(1, N)), L.ptr(loss), L.ptr(g),
               L.stream())
        ctx.save_for_backward(g)
        ctx.shape = sdf.shape
        return loss.view(())

    @staticmethod
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        return (g * go).view(ctx.shape), None


def offsurface_loss(sdf, sharpness=1e2):
    """torch.exp(-sharpness * sdf.abs()).mean() -- train_permuto_sdf.py:372-375"""
    return _OffsurfaceLoss.apply(sdf, float(sharpness))


class _NerfAlpha(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw_density, dt):
        x, d = _c(raw_density).reshape(-1), _c(dt).reshape(-1)
        alpha, om = torch.empty_like(x), torch.empty_like(x)
        L.call("psdf_nerf_alpha_forward", L.c_l(x.shape[0]), L.ptr(x), L.ptr(d), L.ptr(alpha), L.ptr(om), L.stream())
        ctx.save_for_backward(x, d)
        ctx.shape = raw_density.shape
        return alpha.view(ctx.shape), om.view(ctx.shape)

    @staticmethod
    def backward(ctx, ga, gom):
        x, d = ctx.saved_tensors
        g = torch.empty_like(x)
        L.call("psdf_nerf_alpha_backward", L.c_l(x.shape[0]), L.ptr(x), L.ptr(d), L.ptr(None if ga is None else _c(ga).reshape(-1)),
               L.ptr(None if gom is None else _c(gom).reshape(-1)), L.ptr(g), L.stream())
        return g.view(ctx.shape), None


def nerf_alpha(raw_density, dt):
    """alpha = 1 - exp(-softplus(raw_density) * dt) and 1 - alpha + 1e-7 (shaped like raw_density) -- NerfHash's density activation
    (models.py:520) followed by VolumeRenderingNerf.compute_weights' first lines (volume_rendering_modules.py:72-86)"""
    return _NerfAlpha.apply(raw_density, dt)


def sigmoid_rows_raw(x_fm):
    """[C, N] feature-major MLP output -> sigmoid as [N, C] (one launch: transpose + sigmoid)"""
    C, N = x_fm.shape
    y = torch.empty((N, C), dtype=torch.float32, device=x_fm.device)
    L.call("psdf_sigmoid_rows", L.c_l(N), L.c_i(C), L.ptr(x_fm), L.ptr(y), L.stream())
    return y


def sigmoid_rows_backward_raw(g_y, y):
    """gradient w.r.t. the feature-major pre-activation [C, N] from dL/dy [N, C] and y = sigmoid(.) [N, C]"""
    N, C = y.shape
    g = torch.empty((C, N), dtype=torch.float32, device=y.device)
    L.call("psdf_sigmoid_rows_backward", L.c_l(N), L.c_i(C), L.ptr(g_y.contiguous()), L.ptr(y), L.ptr(g), L.stream())
    return g


def nerf_composite_forward_raw(rs, raw_density, rgb, fg_pred=None, fg_bg=None):
    """csrc/composite_fused.hip: softplus density -> opacity -> transmittance -> weights -> radiance of the background container,
    and (given the foreground's radiance [R,3] and bg transmittance [R,1]) the composed radiance, in one launch.
    -> (pred_bg [R,3], pred [R,3] or None)"""
    R, dev = rs.ray_start_end_idx.shape[0], raw_density.device
    pred_bg = torch.empty((R, 3), dtype=torch.float32, device=dev)
    pred = torch.empty((R, 3), dtype=torch.float32, device=dev) if fg_pred is not None else None
    L.call("psdf_nerf_composite_forward", *rs._ri(), L.ptr(raw_density), L.ptr(rs.samples_dt), L.ptr(rgb), L.ptr(fg_pred), L.ptr(fg_bg),
           L.ptr(pred_bg), L.ptr(pred), L.stream())
    return pred_bg, pred


def nerf_composite_backward_raw(rs, max_per_ray, g_pred, raw_density, rgb, fg_bg=None):
    """-> (g_raw_density [M], g_rgb [M,3], g_fg_bg [R,1] or None) for dL/d pred [R,3] of the composed radiance (fg_bg given) or of
    the background radiance alone"""
    from .bridge import VolumeRendering, _per_sample
    M, R, dev = raw_density.shape[0], rs.ray_start_end_idx.shape[0], raw_density.device
    g_raw = _per_sample(rs, (M,), dev)
    g_rgb = _per_sample(rs, (M, 3), dev)
    g_fg = torch.empty((R, 1), dtype=torch.float32, device=dev) if fg_bg is not None else None
    L.call("psdf_nerf_composite_backward", *rs._ri(), L.c_i(int(max_per_ray)), L.ptr(g_pred), L.ptr(fg_bg), L.ptr(raw_density),
           L.ptr(rs.samples_dt), L.ptr(rgb), L.c_i(int(VolumeRendering.reference_compat)), L.ptr(g_raw), L.ptr(g_rgb), L.ptr(g_fg),
           L.stream())
    return g_raw, g_rgb, g_fg


class NerfCompositeFunc(torch.autograd.Function):
    """(raw_density, rgb, fg_pred, fg_bg) -> pred = fg_pred + fg_bg * render(background): NerfHash's density activation,
    VolumeRenderingNerf.compute_weights + integrate and the composition of train_permuto_sdf.py:160-165 as one launch per
    direction; gradients flow to all four inputs."""

    @staticmethod
    def forward(ctx, rs, max_per_ray, raw_density, rgb, fg_pred, fg_bg):
        raw, c, fp, fb = _c(raw_density).reshape(-1), _c(rgb), _c(fg_pred), _c(fg_bg).view(-1, 1)
        _, pred = nerf_composite_forward_raw(rs, raw, c, fp, fb)
        ctx.save_for_backward(raw, c, fb)
        ctx.rs, ctx.max_per_ray, ctx.raw_shape, ctx.bg_shape = rs, int(max_per_ray), raw_density.shape, fg_bg.shape
        return pred

    @staticmethod
    def backward(ctx, g_pred):
        raw, c, fb = ctx.saved_tensors
        g_pred = g_pred.contiguous()
        g_raw, g_rgb, g_fb = nerf_composite_backward_raw(ctx.rs, ctx.max_per_ray, g_pred, raw, c, fb)
        return None, None, g_raw.view(ctx.raw_shape), g_rgb, g_pred, g_fb.view(ctx.bg_shape)


def nerf_composite(rs, max_per_ray, raw_density, rgb, fg_pred, fg_bg):
    return NerfCompositeFunc.apply(rs, max_per_ray, raw_density, rgb, fg_pred, fg_bg)
