"""CPU ORACLE for csrc/neus.hip.  TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline).

A restatement, expression by expression, of the reference's PyTorch code -- the reference evaluates this part with torch
itself, so plain torch on the CPU (fp32, autograd for the backward) IS the reference arithmetic:
  * `neus_alpha`  : VolumeRenderingNeus.compute_weights, permuto_sdf_py/volume_rendering/volume_rendering_modules.py:129-163
                    (from `true_cos` to `alpha`), plus the `1 - alpha + 1e-7` fed to the transmittance product (:166)
  * `inv_s`       : SingleVarianceNetwork.forward + clip, :96-108,137-138
  * `rgb_loss`, `eikonal_loss` : permuto_sdf_py/utils/permuto_sdf_utils.py:43-51
  * `composite_equal` : the weights / integration that follow (:166-172, VolumeRenderingGPU.cuh:401-417,425-481) for rays
                    with an equal sample count, in differentiable torch (exclusive cumulative product, weighted sum)
"""
import torch
import torch.nn.functional as F


def inv_s(variance, forced_variance=None):
    v = variance if forced_variance is None else torch.tensor(1.0) * forced_variance
    return torch.exp(v * 10.0).clip(1e-6, 1e6)


def neus_alpha(sdf, dirs, gradients, dists, inv_s_value, cos_anneal_ratio):
    true_cos = (dirs * gradients).sum(-1, keepdim=True)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) + F.relu(-true_cos) * cos_anneal_ratio)
    estimated_next_sdf = sdf + iter_cos * dists.reshape(-1, 1) * 0.5
    estimated_prev_sdf = sdf - iter_cos * dists.reshape(-1, 1) * 0.5
    prev_cdf = torch.sigmoid(estimated_prev_sdf * inv_s_value)
    next_cdf = torch.sigmoid(estimated_next_sdf * inv_s_value)
    p = prev_cdf - next_cdf
    c = prev_cdf
    alpha = ((p + 1e-5) / (c + 1e-5)).clip(0.0, 1.0)
    return alpha, 1 - alpha + 1e-7


def rgb_loss(gt_rgb, pred_rgb, does_ray_intersect_primitive):
    return ((gt_rgb - pred_rgb).abs() * does_ray_intersect_primitive * 1.0).mean()


def eikonal_loss(sdf_gradients):
    return ((torch.linalg.norm(sdf_gradients.reshape(-1, 3), ord=2, dim=-1) - 1.0) ** 2).mean()


class _IntegrateCompat(torch.autograd.Function):
    """sum_i w_i rgb_i per ray with the backward of the reference's kernel: integrate_with_weights_backward_gpu builds the
    sample colour as (r, g, G) -- `rgb_samples[..][1]` where channel 2 is meant, VolumeRenderingGPU.cuh:1247
    -- so d/dw = g_r r + g_g g + g_b G.  This is the gradient the reference trains with."""

    @staticmethod
    def forward(ctx, w, rgb):                       # w [R,n], rgb [R,n,3]
        ctx.save_for_backward(w, rgb)
        return (w[:, :, None] * rgb).sum(1)

    @staticmethod
    def backward(ctx, g):                           # g [R,3]
        w, rgb = ctx.saved_tensors
        g_rgb = g[:, None, :] * w[:, :, None]
        quirk = torch.stack([rgb[:, :, 0], rgb[:, :, 1], rgb[:, :, 1]], -1)
        return (g[:, None, :] * quirk).sum(-1), g_rgb


def composite_equal(alpha, one_minus, rgb, nr_rays, per_ray, reference_compat=True):
    """transmittance T_i = prod_{j<i} one_minus_j, weights alpha*T, per-ray sum of w*rgb; [R*n, .] packed ray-major.
    `reference_compat` selects the backward of the weighted sum: the reference kernel's (default) or the exact one."""
    om = one_minus.view(nr_rays, per_ray)
    T = torch.cumprod(torch.cat([torch.ones(nr_rays, 1, dtype=om.dtype, device=om.device), om[:, :-1]], 1), dim=1)
    w = alpha.view(nr_rays, per_ray) * T
    rgb3 = rgb.view(nr_rays, per_ray, -1)
    pred = _IntegrateCompat.apply(w, rgb3) if (reference_compat and rgb3.shape[-1] == 3) else (w[:, :, None] * rgb3).sum(1)
    return pred, w.reshape(-1, 1), T.reshape(-1, 1)
