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
    g_inv_s = torch.zeros(1, dtype=torch.float32, device=sdf.device) if need_inv_s else None
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


def l1_loss_raw(pred, gt, mask=None, scale=None, want_grad=True):
    """-> (loss [1], g_pred or None): loss = scale * sum |gt - pred| * mask; scale defaults to 1/numel (the reference's mean)"""
    L.require_cuda(pred, gt)
    R, C = pred.shape
    scale = 1.0 / max(1, R * C) if scale is None else float(scale)
    loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
    g = torch.empty_like(pred) if want_grad else None
    m = None if mask is None else mask.reshape(-1).to(torch.uint8).contiguous()
    L.call("psdf_l1_loss", L.c_l(R), L.c_i(C), L.ptr(pred.contiguous()), L.ptr(gt.contiguous()), L.ptr(m), L.c_f(scale),
           L.ptr(loss), L.ptr(g), L.stream())
    return loss, g


def eikonal_loss_raw(gradients, scale=None, want_grad=True):
    N = gradients.shape[0]
    scale = 1.0 / max(1, N) if scale is None else float(scale)
    loss = torch.zeros(1, dtype=torch.float32, device=gradients.device)
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
