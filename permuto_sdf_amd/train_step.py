"""BASELINE config 4: one optimisation step of PermutoSDF training (SDF + colour + background networks), data parallel.

What the reference does in permuto_sdf_py/train_permuto_sdf.py:311-429 (post sphere-init phase) and run_net (:111-169),
restated on this repository's operators -- same order of operations, same hyper-parameters (:76-105), same losses
(permuto_sdf_py/utils/permuto_sdf_utils.py:43-51) -- for ONE process per GPU with ray sharding:

  rays from the image reel -> sphere intersection -> occupancy-grid sampling (<= 64/ray) -> two rounds of SDF-driven
  importance sampling (no grad, sdf_utils.py:383-423) -> SDF + analytic gradient (autograd, create_graph) -> colour
  network (2nd lattice, SH(5), normals, geometry features, LipshitzMLP) -> NeuS weights -> integrate; background NeRF on
  32 inverse-depth samples of a 4-D lattice -> composite; losses: L1 colour, eikonal, curvature (2nd gradient evaluation
  at a tangentially shifted point), off-surface, Lipschitz bound; backward (double backward through the encoding);
  gradient all-reduce (RCCL) overlapped with nothing it depends on; fused AdamW; every 8th step the occupancy grid is
  refreshed from 262 144 random voxel centres (same seed on every rank: replicas stay identical without communication).

The reference's schedule is part of the trainer (round 3; `Trainer(..., reference_schedule=True)`): the sphere-initialisation
phase (train_permuto_sdf.py:322-330, loss_sphere_init permuto_sdf_utils.py:53-77: 30 000 points, 3e3 * sdf error + 5e1 * eikonal)
for `nr_iter_sphere_fit` iterations, then the linear learning-rate warm-up over 3 000 iterations followed by MultiStepLR (gamma
0.3 at the milestones, counted from the end of the warm-up: :304,419-422, schedulers/warmup.py, multisteplr.py -- `lr_schedule`
below is their closed form), the switch at `iter_start_reduce_curv` (weight decay 1.0 on the colour lattice, eikonal weight 0.01:
:400-405), the colour calibration module (`Colorcal`, models.py:678-730; weight decay 0.1, :299) applied to foreground and
background colours before the sigmoid, and the occupancy refresh BEFORE the backward / optimiser step of its iteration
(:383-391).  With `reference_schedule=False` (tools/train_bench.py: the steady-state step of BASELINE config 4) the sphere phase,
the warm-up and the calibration module are left out; everything else is the same code.  Network initialisation follows the
reference (leaky_relu_init, common_utils.py:248-293; sdf_shift added to the whole last bias vector, models.py:163-165).

Differences from the reference, all structural: no-grad SDF evaluations run in the fused single-launch evaluator with
a 1-row head (csrc/fused.hip); the samplers are exact-size (no second compaction pass); the per-step ray count adapts
from the sample count the step already knows (no extra sync).  The image data are synthetic (DTU is not available
offline): `SyntheticReel` has the TensorReel fields `random_rays_from_reel` reads (src/PermutoSDF.cu:70-102).
"""
import math

import numpy as np
import os

import torch
import torch.nn.functional as F

from . import parallel
from .bridge import OccupancyGrid, PermutoSDF, RaySampler, Sphere, VolumeRendering
from .encoding import Coarse2Fine, PermutoEncoding
from .fused import encode_mlp_forward_raw
from .mlp import FusedMLP, LipshitzMLP, pack_params
from .neus import (curvature_loss, curvature_shift, eikonal_loss, l1_loss, nerf_alpha, nerf_composite, neus_alpha, neus_composite,
                   normalize3, offsurface_loss)
from .optim import FusedAdamW


class HyperParams:
    """train_permuto_sdf.py:76-105"""
    lr = 1e-3
    forced_variance_finish_iter = 35000
    forced_variance_finish = 0.8
    eikonal_weight = 0.04
    curvature_weight = 0.65
    lipshitz_weight = 3e-6
    offsurface_weight = 1e-4
    iter_start_reduce_curv = 50000
    iter_finish_reduce_curv = 50000 + 1001
    nr_samples_bg = 32
    min_dist_between_samples = 1e-4
    max_nr_samples_per_ray = 64
    nr_samples_imp_sampling = 16
    nr_rays = 512
    sdf_geom_feat_size = 32
    sdf_nr_iters_for_c2f = 10000
    target_nr_of_samples = 512 * (64 + 16 + 16)
    # the schedule (train_permuto_sdf.py:80,89-90,98; warm-up length :420)
    nr_iter_sphere_fit = 4000
    lr_warmup_iters = 3000
    lr_milestones = (100000, 150000, 180000, 190000)
    lr_gamma = 0.3
    iter_finish_training = 200000
    use_color_calibration = True
    eikonal_weight_late = 0.01           # from iter_start_reduce_curv on (:405)
    rgb_lattice_weight_decay_late = 1.0  # (:402-403)
    mask_weight = 0.1                    # (:85) weight of the mask loss of `--with_mask` runs (:381-383)
    colorcal_weight_decay = 1e-1         # (:299)


def lr_schedule(global_iter, hp):
    """Learning rate the optimiser step of iteration `global_iter` (0-based, sphere phase included) runs with: the closed form
    of what the reference's scheduler objects do (train_permuto_sdf.py:304,419-422).  Through iteration nr_iter_sphere_fit the
    base rate; GradualWarmupScheduler(multiplier=1, total_epoch=W) is created right after that step (its constructor already
    steps once, to 0) and stepped after every later one: iteration n0 + k runs at base * k / W for k <= W, at base for
    k = W + 1 (the hand-over), and MultiStepLR -- constructed at start-up, stepped only from then on -- has counted
    m = k - (W + 1) epochs: base * gamma^(number of milestones <= m)."""
    n0, W = int(hp.nr_iter_sphere_fit), int(hp.lr_warmup_iters)
    if global_iter <= n0:
        return hp.lr
    k = global_iter - n0
    if k <= W:
        return hp.lr * (float(k) / W)
    m = k - (W + 1)
    return hp.lr * hp.lr_gamma ** sum(1 for ms in hp.lr_milestones if ms <= m)


def map_range_val(v, in_lo, in_hi, out_lo, out_hi):
    t = min(max((v - in_lo) / (in_hi - in_lo), 0.0), 1.0)
    return out_lo + t * (out_hi - out_lo)


# ------------------------------------------------------------------------------------- differentiable compositing
class _Cumprod(torch.autograd.Function):  # volume_rendering_funcs.py:55-118
    @staticmethod
    def forward(ctx, rs, one_minus_alpha):
        T, bg = VolumeRendering.cumprod_alpha2transmittance(rs, one_minus_alpha)
        ctx.save_for_backward(one_minus_alpha, T, bg)
        ctx.rs = rs
        return T, bg

    @staticmethod
    def backward(ctx, gT, gbg):
        a, T, bg = ctx.saved_tensors
        cs = VolumeRendering.cumsum_over_each_ray(ctx.rs, gT * T, True)
        return None, VolumeRendering.cumprod_alpha2transmittance_backward(gT, gbg, ctx.rs, a, T, bg, cs)


class _Integrate(torch.autograd.Function):  # volume_rendering_funcs.py:161-190
    @staticmethod
    def forward(ctx, rs, vals, w):
        out = VolumeRendering.integrate_with_weights(rs, vals, w)
        ctx.save_for_backward(vals, w, out)
        ctx.rs = rs
        return out

    @staticmethod
    def backward(ctx, g):
        vals, w, out = ctx.saved_tensors
        gv, gw = VolumeRendering.integrate_with_weights_backward(g.contiguous(), ctx.rs, vals, w, out)
        return None, gv, gw


class _SumRay(torch.autograd.Function):  # volume_rendering_funcs.py:194-224
    @staticmethod
    def forward(ctx, rs, v):
        per_ray, per_sample = VolumeRendering.sum_over_each_ray(rs, v)
        ctx.save_for_backward(v)
        ctx.rs = rs
        return per_ray, per_sample

    @staticmethod
    def backward(ctx, g_ray, g_sample):
        (v,) = ctx.saved_tensors
        return None, VolumeRendering.sum_over_each_ray_backward(g_ray.contiguous(), g_sample.contiguous(), ctx.rs, v)


# ---- feature-major glue: the fused kernels read and write [C, N]; the reference's Python speaks [N, C]
def cat_fm(parts):
    """torch.cat(parts, 1) for [N, c_i] tensors, laid out FEATURE-MAJOR underneath (the result is the transposed view of a
    contiguous [sum c_i, N] buffer): the fused MLPs consume it without the [N, C] -> [C, N] copy, and in the backward every
    part receives a contiguous row block of the MLP's input gradient instead of a strided column slice that the next kernel
    would have to copy."""
    return torch.cat([p.t() for p in parts], 0).t()


class _SplitHead(torch.autograd.Function):
    """y [N, 1 + g] (a transposed view of the MLP's feature-major output) -> (y[:, :1], y[:, 1:]); the backward assembles the
    gradient feature-major in ONE cat (autograd's slice backward: two zero fills, two strided copies, one add, and then the
    MLP backward's own transposing copy)."""

    @staticmethod
    def forward(ctx, y):
        ctx.shape = y.shape
        return y[:, 0:1], y[:, 1:]

    @staticmethod
    def backward(ctx, g0, g1):
        N, C = ctx.shape
        ref = g0 if g0 is not None else g1
        if g0 is None:
            g0 = ref.new_zeros(N, 1)
        if g1 is None:
            g1 = ref.new_zeros(N, C - 1)
        return torch.cat([g0.t(), g1.t()], 0).t()


# --------------------------------------------------------------------------------------------------- networks
def _lattice(pos_dim, points_scaling):
    return PermutoEncoding(pos_dim, 2 ** 18, 24, 2, np.geomspace(1.0, 1e-4, 24), appply_random_shift_per_level=True,
                           concat_points=True, concat_points_scaling=points_scaling)


class SdfNet(torch.nn.Module):
    """models.py:131-251: 24-level lattice + Linear(52,32)-GELU-(32,32)-GELU-(32,32)-GELU-(32,1+32) in the fused
    evaluator (twice differentiable: the eikonal/curvature losses differentiate THROUGH its input gradient,
    create_graph=True); the no-grad path (`sdf_only`) is one fused encode->MLP launch."""

    def __init__(self, hp):
        super().__init__()
        self.encoding = _lattice(3, 1e-3)
        g = hp.sdf_geom_feat_size
        self.mlp_sdf = FusedMLP([self.encoding.output_dims(), 32, 32, 32, 1 + g], reference_init=True)   # models.py:161-162
        with torch.no_grad():  # models.py:163-165: `mlp_sdf[-1].bias += sdf_shift`, the whole bias vector
            self.mlp_sdf.layers[-1].bias += 1e-2
        self.c2f = Coarse2Fine(24)
        self.nr_iters_for_c2f = hp.sdf_nr_iters_for_c2f

    def window(self, it):
        # read-only use: the cached window itself (no copy per call; Coarse2Fine.__call__ hands out copies)
        return self.c2f.window_readonly(map_range_val(it, 0.0, self.nr_iters_for_c2f, 0.3, 1.0)).to(self.encoding.lattice_values.device)

    def forward(self, points, it):
        return _SplitHead.apply(self.mlp_sdf(self.encoding(points, self.window(it))))

    @torch.no_grad()
    def sdf_only(self, points, it, key=None):
        """[N,1]; the SDF is row 0 of the last layer (models.py:190).  `key`: anything that changes whenever the parameters do
        (the trainer passes its iteration counter): calls with the same key share one packed weight image instead of packing
        it again (three evaluations per training step)."""
        lin = list(self.mlp_sdf.layers)
        dims = [lin[0].in_features, 32, 32, 32, 1]
        if key is not None:      # torch-level writes (load_state_dict, copy_) bump _version; the fused optimiser's do not: hence both
            key = (key, tuple(p._version for l in lin for p in (l.weight, l.bias)))
        cached = getattr(self, "_sdf_only_packed", None)
        if key is not None and cached is not None and cached[0] == key:
            packed = cached[1]
        else:
            ws, bs = [l.weight for l in lin], [l.bias for l in lin]
            ws[-1], bs[-1] = ws[-1][0:1].contiguous(), bs[-1][0:1].contiguous()
            packed = pack_params(dims, ws, bs)
            self._sdf_only_packed = (key, packed) if key is not None else None
        e = self.encoding
        y, _ = encode_mlp_forward_raw(e.cfg, points.contiguous(), e.lattice_values.detach(), e.scale_factor,
                                      e.random_shift_per_level.detach(), self.window(it).contiguous(), dims, packed)
        return y.view(-1, 1)

    def sdf_and_gradient(self, points, it):  # models.py:236-251
        with torch.enable_grad():
            if not points.requires_grad:   # the shifted points of the curvature term arrive WITH their graph (models.py:277)
                points = points.detach().requires_grad_(True)
            sdf, feat = self.forward(points, it)
            # autograd would compute the lattice and the MLP parameter gradients here and drop them
            with self.encoding.positions_gradient_only(), self.mlp_sdf.input_gradient_only():
                (grad,) = torch.autograd.grad(sdf, points, torch.ones_like(sdf), create_graph=True, retain_graph=True)
        return sdf, grad, feat

    def curvature(self, points, sdf_gradients, it):  # models.py:257-291, and the .mean() of train_permuto_sdf.py:363
        """mean angle (over pi) between the normal and the normal at a point shifted 1e-4 along a random tangent; the
        shifted point keeps its dependence on the normal, as in the reference (no detach at models.py:272-277)"""
        shifted = curvature_shift(points, sdf_gradients, torch.randn_like(points), 1e-4)
        _, g2, _ = self.sdf_and_gradient(shifted, it)
        return curvature_loss(sdf_gradients, g2)


class RgbNet(torch.nn.Module):
    """models.py:310-391: 2nd lattice (points scaling 1) + SH(5) of the view direction + normal + geometry features ->
    LipshitzMLP [128,128,64,3] -> sigmoid."""

    def __init__(self, hp):
        super().__init__()
        self.encoding = _lattice(3, 1.0)
        self.mlp = LipshitzMLP(self.encoding.output_dims() + 25 + 3 + hp.sdf_geom_feat_size, [128, 128, 64, 3], True)
        self.variance = torch.nn.Parameter(torch.tensor(0.3))  # SingleVarianceNetwork, volume_rendering_modules.py:96-115
        self.last_inv_s = None
        self.register_buffer("_win", torch.ones(24), persistent=False)  # rgb_nr_iters_for_c2f = 1: window is 1 from the first step

    def forward(self, points, dirs, sdf_gradients, geom_feat, colorcal=None, img_indices=None, ray_start_end_idx=None):
        win = self._win
        with torch.no_grad():
            sh = PermutoSDF.spherical_harmonics(dirs, 5)
        x = self.mlp(cat_fm([self.encoding(points, win), sh, normalize3(sdf_gradients), geom_feat]))
        if colorcal is not None:      # models.py:384-385
            x = colorcal.calib_RGB_samples_packed(x, img_indices, ray_start_end_idx)
        return torch.sigmoid(x)

    def neus_weights(self, rs, sdf, gradients, cos_anneal_ratio, forced_variance):  # volume_rendering_modules.py:129-174
        v = self.variance if forced_variance is None else torch.tensor(float(forced_variance), device=sdf.device)
        inv_s = torch.exp(v * 10.0).clip(1e-6, 1e6)
        self.last_inv_s = inv_s.detach()
        # cosine annealing, section-point SDFs, two sigmoids, clipped ratio: ONE launch per direction (csrc/neus.hip)
        # instead of ~30 torch elementwise launches; gradients flow to sdf, the SDF gradient and inv_s (the variance)
        alpha, one_minus = neus_alpha(sdf, rs.samples_dirs, gradients, rs.samples_dt, inv_s, cos_anneal_ratio)
        T, bg = _Cumprod.apply(rs, one_minus)
        w = (alpha * T).view(-1, 1)
        w_sum, _ = _SumRay.apply(rs, w)
        return w, w_sum, bg

    def neus_render(self, rs, max_per_ray, sdf, gradients, rgb, cos_anneal_ratio, forced_variance):
        """compute_weights + integrate (volume_rendering_modules.py:129-190) in one launch per direction
        (csrc/composite_fused.hip) -> (radiance [R,3], bg transmittance [R,1]); gradients flow to sdf, the SDF gradient, the
        colours and the variance"""
        v = self.variance if forced_variance is None else torch.tensor(float(forced_variance), device=sdf.device)
        inv_s = torch.exp(v * 10.0).clip(1e-6, 1e6)
        self.last_inv_s = inv_s.detach()
        return neus_composite(rs, max_per_ray, sdf, gradients, rgb, inv_s, cos_anneal_ratio)


# (RgbNet.neus_render below is the fused form of neus_weights + _Integrate: one launch per direction)


class BgNet(torch.nn.Module):
    """NerfHash, models.py:431-526: 4-D lattice, density+feature net 52->64x3->65, colour head [64+16]->64->64->3."""

    def __init__(self):
        super().__init__()
        self.encoding = _lattice(4, 1.0)
        # models.py:451-472: leaky_relu_init everywhere; only mlp_rgb's last layer gets the linear (gain 1) init
        self.mlp_feat_and_density = FusedMLP([self.encoding.output_dims(), 64, 64, 64, 65], reference_init=True,
                                             last_layer_linear_init=False)
        self.mlp_rgb = FusedMLP([64 + 16, 64, 64, 3], reference_init=True)
        self.register_buffer("_win", torch.ones(24), persistent=False)

    def forward(self, pos4d, dirs, colorcal=None, img_indices=None, ray_start_end_idx=None):
        win = self._win
        with torch.no_grad():
            sh = PermutoSDF.spherical_harmonics(dirs, 4)
        fd = self.mlp_feat_and_density(self.encoding(pos4d, win))
        rgb = self.mlp_rgb(cat_fm([F.gelu(fd[:, 1:65]), sh]))
        if colorcal is not None:      # models.py:523-524
            rgb = colorcal.calib_RGB_samples_packed(rgb, img_indices, ray_start_end_idx)
        return torch.sigmoid(rgb), fd[:, 0:1]     # colour, RAW density: softplus (models.py:520) is fused into nerf_weights

    @staticmethod
    def nerf_weights(rs, raw_density):  # models.py:520 + volume_rendering_modules.py:72-86
        alpha, one_minus = nerf_alpha(raw_density, rs.samples_dt)
        T, bg = _Cumprod.apply(rs, one_minus)
        return (alpha * T).view(-1, 1)


class Colorcal(torch.nn.Module):
    """models.py:678-730: per-image affine colour calibration of the predicted sample colours, rgb * (1 + weight_delta[img]) +
    bias[img]; image `idx_with_fixed_calib` keeps the identity.  Parameter names as in the reference (colorcal_model.pt)."""

    def __init__(self, nr_cams, idx_with_fixed_calib=0):
        super().__init__()
        self.idx_with_fixed_calib = idx_with_fixed_calib
        self.weight_delta = torch.nn.Parameter(torch.zeros(nr_cams, 3))
        self.bias = torch.nn.Parameter(torch.zeros(nr_cams, 3))

    def calib_RGB_samples_packed(self, rgb_samples, per_pixel_img_indices, ray_start_end_idx):
        from .bridge import RaySamplesPacked
        idx = per_pixel_img_indices.long()
        fixed = (idx == self.idx_with_fixed_calib)[:, None]
        w = torch.where(fixed, torch.ones_like(self.weight_delta[:1]), 1.0 + self.weight_delta.index_select(0, idx))
        b = torch.where(fixed, torch.zeros_like(self.bias[:1]), self.bias.index_select(0, idx))
        ray = RaySamplesPacked.compute_per_sample_ray_idx(ray_start_end_idx, rgb_samples.shape[0]).long()
        return rgb_samples * w.index_select(0, ray) + b.index_select(0, ray)


class SyntheticReel:
    """The TensorReel fields read by random_rays_from_reel (src/PermutoSDF.cu:70-102): `nr_images` random images, pinhole
    cameras on a sphere of radius `dist` looking at the origin."""

    def __init__(self, device, nr_images=49, height=300, width=400, dist=1.5, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.rgb_reel = torch.rand(nr_images, 3, height, width, generator=g).to(device)
        self.mask_reel = torch.ones(nr_images, 1, height, width, device=device)
        self.has_mask = False
        K = torch.tensor([[1.2 * width, 0, width / 2], [0, 1.2 * width, height / 2], [0, 0, 1.0]])
        self.K_reel = K.expand(nr_images, 3, 3).contiguous().to(device)
        tf = torch.zeros(nr_images, 4, 4)
        for i in range(nr_images):
            c = F.normalize(torch.randn(3, generator=g), dim=0) * dist
            z = F.normalize(-c, dim=0)
            x = F.normalize(torch.cross(torch.tensor([0.0, 1.0, 0.0]), z, dim=0), dim=0)
            y = torch.cross(z, x, dim=0)
            tf[i, :3, 0], tf[i, :3, 1], tf[i, :3, 2], tf[i, :3, 3], tf[i, 3, 3] = x, y, z, c, 1.0
        self.tf_world_cam_reel = tf.to(device)


# ------------------------------------------------------------------------------------------------------ trainer
class Trainer:
    def __init__(self, device, hp=None, seed=0, touched_rows=True, reference_schedule=False, nr_images=49, with_mask=False):
        """reference_schedule: run the reference's whole schedule (sphere phase, LR warm-up / decay, late switches, colour
        calibration over `nr_images` cameras: see the module docstring); False = the steady-state step only.
        with_mask: the reference's `--with_mask` mode (train_permuto_sdf.py:153-154,381-383; nerf_utils.py:519-522): no
        background samples and no background network evaluation, the rendered colour is the foreground's alone, and the
        per-ray weight sum is held to the mask of the image reel by a binary cross entropy (weight `mask_weight`)."""
        self.hp = hp or HyperParams()
        self.dev = torch.device(device)
        self.reference_schedule = bool(reference_schedule)
        self.with_mask = bool(with_mask)
        torch.manual_seed(seed)  # identical replicas on every rank
        self.sdf, self.rgb, self.bg = SdfNet(self.hp).to(self.dev), RgbNet(self.hp).to(self.dev), BgNet().to(self.dev)
        self.colorcal = (Colorcal(nr_images, 0).to(self.dev)
                         if (self.reference_schedule and self.hp.use_color_calibration) else None)
        self.sphere = Sphere(0.5, [0, 0, 0])
        self.grid = OccupancyGrid(256, 1.0, [0, 0, 0], device=self.dev)
        # parameter groups as in train_permuto_sdf.py:293-299: the colour lattice has its own (its weight decay is switched on
        # late), the calibration module decays from the start
        rgb_lat = self.rgb.encoding.lattice_values
        base = [p for m in (self.sdf, self.rgb, self.bg) for p in m.parameters() if p.requires_grad and p is not rgb_lat]
        groups = [{"params": base, "weight_decay": 0.0, "name": "base"},
                  {"params": [rgb_lat], "weight_decay": 0.0, "name": "model_rgb_only_encoding"}]
        if self.colorcal is not None:
            groups.append({"params": list(self.colorcal.parameters()), "weight_decay": self.hp.colorcal_weight_decay,
                           "name": "model_colorcal"})
        self.params = [p for g in groups for p in g["params"]]
        self.opt = FusedAdamW(groups, lr=self.hp.lr, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.0)
        # touched-rows path for the three 50-MB lattices: persistent gradient buffers that all backward calls
        # of the step accumulate into, AdamW + zero-fill only over row blocks that are touched or carry non-zero moments
        self.touched = []
        if touched_rows:
            for m in (self.sdf, self.rgb, self.bg):
                tr = m.encoding.enable_touched_rows()
                self.opt.attach(m.encoding.lattice_values, tr)
                self.touched.append(tr)
        # the SDF net is differentiated four times per step (two evaluations, two analytic input gradients): its parameter
        # gradients accumulate in one persistent buffer instead of four tensors per parameter summed by autograd
        self.grad_buffers = [self.sdf.mlp_sdf.enable_grad_buffer()] if touched_rows else []
        self.nr_rays = self.hp.nr_rays
        self.iter = 0
        self.capture_grads = None     # set to {} to have step() record the gradients it hands to the optimiser
        self.shard_optimizer = parallel.sharded_optimizer_default()   # data parallel: lattices updated by their owners only
        # data-parallel schedule of a step (round 6): a lattice's gradient leaves (reduce-scatter) as soon as its last backward
        # kernel is enqueued (_dp_lattice_final: background -> colour -> SDF in the hand-written step), the parameters come back
        # (all-gather) in the order the NEXT step reads them and are only waited for where it does (_params_ready)
        self._dp_started = {}         # lattice index -> (ShardedUpdate | None, own range | None) of the running step
        self._dp_buckets = None
        self._pending_gather = {}     # lattice index -> ShardedUpdate whose all-gather of the parameters is in flight
        self.defer_param_gather = os.environ.get("PSDF_DP_DEFER_GATHER", "1") == "1"
        self._pinned_counts = None    # host landing zones of the march's per-ray counts (one asynchronous copy per step)
        self._pinned_flip = 0
        self._colour_window_t = 1.0   # the t the colour / background lattices' windows (`_win`, ones) currently hold
        self._late_seen = False       # set by the first iteration at / after iter_start_reduce_curv (acts from the next one on)
        self.last = {}
        # per-rank random streams for rays/jitter, one shared stream for the grid refresh
        self._seed = seed + 1

    def _prefetch_valid(self, git, reel=None):
        """True when the first half of iteration `git`'s sampling (from `reel`) has already been issued (train_manual.ManualTrainer)"""
        return False

    def _param_key(self):
        """changes whenever the parameters may have: the optimiser counts its steps (`generation`: the fused kernels write the
        parameters without bumping torch's version counters), torch-level writes bump `_version` (SdfNet.sdf_only adds those)"""
        return (self.iter, self.opt.generation)

    def accumulate_grads(self):
        """context manager around the step's `loss.backward()`: opens the persistent gradient buffers (the lattices'
        TouchedRows.grad, the SDF net's GradBuffer) for accumulation.  Backward passes outside it -- an auxiliary
        torch.autograd.grad, a diagnostic .backward() -- behave like plain autograd and cannot leak into the next optimiser step."""
        import contextlib
        stack = contextlib.ExitStack()
        for b in list(self.touched) + list(self.grad_buffers):
            stack.enter_context(b.accumulate())
        return stack

    # ---- checkpoints in the reference's file layout (permuto_sdf_utils.py:222-237)
    def save_checkpoint(self, folder):
        from . import checkpoint
        checkpoint.save(folder, sdf=self.sdf, rgb=self.rgb, bg=self.bg, grid=self.grid, colorcal=self.colorcal)

    def load_checkpoint(self, folder):
        import os
        from . import checkpoint
        cc = self.colorcal if os.path.exists(os.path.join(folder, "colorcal_model.pt")) else None
        checkpoint.load(folder, sdf=self.sdf, rgb=self.rgb, bg=self.bg, grid=self.grid, colorcal=cc, map_location=self.dev)

    # ---- sampling (no grad): nerf_utils.py:502-525 + sdf_utils.py:383-423
    @torch.no_grad()
    def _samples_begin(self, o, d, jitter=True):
        """first half of the sampling phase, everything that needs neither a network nor the host: sphere intersection, the
        occupancy march, the background sampler, and the march's per-ray counts on their way to pinned memory (an asynchronous
        copy; `event` fires when it has landed).  Runs on whatever stream is current: train_manual.ManualTrainer issues it for
        the NEXT step on a side stream while this step's backward runs."""
        hp = self.hp
        _, te, _, tx, _ = self.sphere.ray_intersection(o, d)
        pool = self.grid.compute_samples_in_occupied_regions(o, d, te, tx, hp.min_dist_between_samples,
                                                             hp.max_nr_samples_per_ray, jitter)
        R = o.shape[0]
        pinned = self._pinned(R)
        pinned[:R].copy_(pool._ray_counts, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        # (the centre as the host list: handed the CUDA tensor the bridge has to read it back -- a stream sync per step)
        bg = None if self.with_mask else RaySampler.compute_samples_bg(o, d, tx, hp.nr_samples_bg, self.sphere.m_radius,
                                                                       self.sphere.m_center, jitter, False)
        return dict(o=o, d=d, tx=tx, pool=pool, bg=bg, counts=pinned[:R], event=event, jitter=jitter)

    def _pinned(self, R):
        """host landing zones of the march's per-ray counts: two, used alternately (a prefetched step's counts are in flight
        while the current step still reads its own)"""
        if self._pinned_counts is None or self._pinned_counts[0].numel() < R:
            self._pinned_counts = [torch.empty(max(R, 8192), dtype=torch.int32, device="cpu").pin_memory() for _ in range(2)]   # (explicit
            # device: the reference's scripts set a CUDA default tensor type)
        self._pinned_flip ^= 1
        return self._pinned_counts[self._pinned_flip]

    def _samples(self, o, d, it, jitter=True, between=None, begun=None):
        """-> (foreground container with both importance rounds merged in, background container or None).
        between(bg): optional work to ENQUEUE while the host waits for the march's counts -- anything that needs the background
        samples only (train_manual.ManualTrainer: the background network's forward).  The host waits for the EVENT of the counts'
        copy, not for the stream: the kernels `between` enqueued keep the GPU busy across what would otherwise be the step's one
        pipeline bubble.  begun: the result of an earlier `_samples_begin` for these rays (the prefetched first half)."""
        hp = self.hp
        b = begun if begun is not None else self._samples_begin(o, d, jitter)
        tx, pool, bg = b["tx"], b["pool"], b["bg"]
        # ONE host sync for the whole sampling phase (the reference has three: a `.item()` per compaction, src/RaySamplesPacked.cu:
        # 44-54): the march's per-ray counts come to the host once; a ray holds 0 or >= 3 samples (OccupancyGridGPU.cuh:685-689),
        # and every importance round adds exactly nr_samples_imp_sampling to each non-empty ray and nothing to the others
        # (combine_count_kernel: n <= 1 ? 0 : n + nr_imp), so the later counts are host arithmetic
        if between is not None:
            between(bg)
        b["event"].synchronize()
        counts = b["counts"]
        n_known, nonempty = int(counts.sum()), int((counts > 0).sum())
        if n_known > pool.max_nr_samples:      # pool overflow (silent in the reference): the generic path sorts it out
            n_known = nonempty = None
        fg = pool.compact_to_valid_samples(known_nr_samples=n_known)
        if fg.samples_pos.shape[0] == 0:
            return fg, bg
        fg.set_sdf(self.sdf.sdf_only(fg.samples_pos, it, key=self._param_key()))
        for rnd, mult in ((0, 1.0), (1, 2.0)):
            # sdf2alpha -> clip -> 1 - alpha + 1e-7 -> cumprod -> alpha * T -> per-ray sum -> normalise -> cdf
            # (sdf_utils.py:403-417), one launch, bit-identical to the nine of the operator chain
            cdf = VolumeRendering.sdf_importance_cdf(fg, fg.samples_sdf, 512.0, True, mult)
            imp = VolumeRendering.importance_sample(o, d, fg, cdf, hp.nr_samples_imp_sampling, jitter)
            if rnd == 0:
                imp.set_sdf(self.sdf.sdf_only(imp.samples_pos, it, key=self._param_key()))
            else:
                fg.remove_sdf()
            if n_known is not None:
                n_known += nonempty * hp.nr_samples_imp_sampling
            fg = VolumeRendering.combine_uniform_samples_with_imp(o, d, tx, fg, imp).compact_to_valid_samples(known_nr_samples=n_known)
        return fg, bg

    # ---- run_net: train_permuto_sdf.py:111-169
    def _render(self, o, d, it, cos_anneal_ratio, forced_variance, jitter=True, img_indices=None):
        """`jitter` = the reference's `model.training` flag (nerf_utils.py:507, sdf_utils.py:401): False in eval mode"""
        fg, bg = self._samples(o, d, it, jitter)
        cc = self.colorcal if img_indices is not None else None
        R = o.shape[0]
        w_sum = None
        if fg.samples_pos.shape[0] == 0:
            pred = torch.zeros(R, 3, device=self.dev)
            sdf_grad, bgT = torch.zeros(0, 3, device=self.dev), torch.ones(R, 1, device=self.dev)
            w_sum = torch.zeros(R, 1, device=self.dev)
        else:
            sdf, sdf_grad, feat = self.sdf.sdf_and_gradient(fg.samples_pos, it)
            rgb = self.rgb(fg.samples_pos, fg.samples_dirs, sdf_grad, feat, cc, img_indices, fg.ray_start_end_idx)
            # a ray holds at most max_nr_samples_per_ray uniform + 2 rounds of importance samples
            per_ray = self.hp.max_nr_samples_per_ray + 2 * self.hp.nr_samples_imp_sampling
            if per_ray <= 256 and not self.with_mask:
                pred, bgT = self.rgb.neus_render(fg, per_ray, sdf, sdf_grad, rgb, cos_anneal_ratio, forced_variance)
            else:       # the operator chain: any ray length, and the per-ray weight sum the mask loss needs
                w, w_sum, bgT = self.rgb.neus_weights(fg, sdf, sdf_grad, cos_anneal_ratio, forced_variance)
                pred = _Integrate.apply(fg, rgb, w)
        if bg is None:                      # --with_mask: no background model (train_permuto_sdf.py:153-154)
            return pred, sdf_grad, fg, w_sum
        rgb_bg, dens = self.bg(bg.samples_pos_4d, bg.samples_dirs, cc, img_indices, bg.ray_start_end_idx)
        if self.hp.nr_samples_bg <= 256:    # density activation, weights, integration and the composition: one launch per direction
            pred = nerf_composite(bg, self.hp.nr_samples_bg, dens, rgb_bg, pred, bgT.view(-1, 1))
        else:
            pred = pred + bgT.view(-1, 1) * _Integrate.apply(bg, rgb_bg, BgNet.nerf_weights(bg, dens.view(-1, 1)))
        return pred, sdf_grad, fg, w_sum

    def _sphere_init_loss(self, it):
        """loss_sphere_init (permuto_sdf_utils.py:53-77 -> sdf_utils.py:60-83, dataset dtu): fit the SDF of a radius-0.3 sphere
        at 30 000 random points of the bounding sphere, 3e3 * mean squared SDF error + 5e1 * eikonal"""
        pts = self.sphere.rand_points_inside(30000)
        sdf, grad, _ = self.sdf.sdf_and_gradient(pts, it)
        dists = pts.norm(dim=-1, keepdim=True) - 0.3
        return ((sdf - dists) ** 2).mean() * 3e3 + eikonal_loss(grad) * 5e1

    # ------------------------------------------------------------------ data parallel: early reduction, late gather
    def _dp_lattice_final(self, k):
        """The gradient buffer of lattice k (0 SDF, 1 colour, 2 background) is final for this step: start its reduction NOW, on
        the collective's stream, while the rest of the backward is still being enqueued.  Idempotent; step() calls it for
        whatever was not started earlier.  Also ORs the touched-block byte map over the ranks (a block touched on ANY rank
        carries gradient after the sum)."""
        if k in self._dp_started or not parallel.collectives_active() or not self.touched:
            return
        tr = self.touched[k]
        su = parallel.ShardedUpdate()
        own = su.reduce_scatter(tr.grad.view(-1), unit=tr.block_elems) if self.shard_optimizer else None
        if own is None:
            if self._dp_buckets is None:
                self._dp_buckets = parallel.GradientBuckets()
            self._dp_buckets.reduce([tr.grad])       # one bucket per lattice: the ring is per-link bound, fewer larger messages
            su = None
        parallel.all_reduce_max_(tr.touched)
        self._dp_started[k] = (su, own)

    def _params_ready(self, k=None):
        """make the current stream wait for the all-gather of lattice k's parameters (all of them: k = None) that the previous
        step left in flight; no-op when nothing is pending.  Everything that READS lattice parameters calls this first."""
        if not self._pending_gather:
            return
        for key in ([k] if k is not None else list(self._pending_gather)):
            su = self._pending_gather.pop(key, None)
            if su is not None:
                su.wait()

    def sync_parameters(self):
        """public form of _params_ready(): call before reading parameters from outside a step (checkpoints, comparisons)"""
        self._params_ready()

    def _refresh_and_adapt(self, it, git, n_fg):
        """occupancy refresh every 8th step, BEFORE this iteration's backward / optimiser step as in the reference
        (train_permuto_sdf.py:383-391), from the same random voxels on every rank; adaptive ray count (:393-397)"""
        hp = self.hp
        with torch.no_grad():
            if git % 8 == 0:
                self._params_ready()            # (the refresh evaluates the SDF)
                parallel.seed_generators(977 + git, self.dev)
                centres, idx = self.grid.compute_random_sample_of_grid_points(256 * 256 * 4, True)
                inv_s = self.rgb.last_inv_s if self.rgb.last_inv_s is not None else torch.tensor(20.0, device=self.dev)
                self.grid.update_with_sdf_random_sample(idx, self.sdf.sdf_only(centres, it, key=self._param_key()), inv_s.view(1), 1e-4)
            if n_fg:  # the count is already on the host
                self.nr_rays = max(64, min(8192, int(self.nr_rays * hp.target_nr_of_samples / n_fg)))

    def _draw_rays(self, reel):
        with torch.no_grad():
            o, d, gt, mask, img_idx = PermutoSDF.random_rays_from_reel(reel, self.nr_rays)
            _, _, _, _, hit = self.sphere.ray_intersection(o, d)
        return o, d, gt, hit, img_idx, mask

    def _main_phase(self, reel, it, git, eikonal_weight):
        """forward + losses of one iteration of the main phase as an autograd graph -> (loss, n_fg, nr_rays, False): the caller
        runs loss.backward().  train_manual.ManualTrainer overrides this with a hand-written backward over the raw kernels."""
        hp = self.hp
        self._params_ready()
        cos_anneal_ratio = map_range_val(it, 0.0, hp.forced_variance_finish_iter, 0.0, 1.0)
        forced_variance = map_range_val(it, 0.0, hp.forced_variance_finish_iter, 0.3, hp.forced_variance_finish)
        o, d, gt, hit, img_idx, gt_mask = self._draw_rays(reel)
        pred, sdf_grad, fg, w_sum = self._render(o, d, it, cos_anneal_ratio, forced_variance,
                                                 img_indices=img_idx if self.colorcal is not None else None)
        loss = l1_loss(pred, gt, hit)                                                      # rgb_loss, one launch
        n_fg = fg.samples_pos.shape[0]
        if n_fg:
            loss = loss + eikonal_loss(sdf_grad) * eikonal_weight
            gw = map_range_val(it, hp.iter_start_reduce_curv, hp.iter_finish_reduce_curv, 1.0, 0.0)
            if gw > 0.0:
                loss = loss + self.sdf.curvature(fg.samples_pos, sdf_grad, it) * (hp.curvature_weight * gw)
        off = self.sphere.rand_points_inside(1024)
        sdf_off, _ = self.sdf(off, it)
        loss = loss + offsurface_loss(sdf_off, 1e2) * hp.offsurface_weight
        if it >= hp.iter_start_reduce_curv:
            loss = loss + self.rgb.mlp.lipshitz_bound_full().mean() * hp.lipshitz_weight
        if self.with_mask:                                                                 # train_permuto_sdf.py:381-383
            loss = loss + F.binary_cross_entropy(w_sum.clip(1e-3, 1.0 - 1e-3), gt_mask) * hp.mask_weight
        self._refresh_and_adapt(it, git, n_fg)
        return loss, n_fg, o.shape[0], False

    def step(self, reel):
        """one optimisation step; returns the loss (device tensor, no sync)"""
        hp, git = self.hp, self.iter                                  # git: global iteration (sphere phase included)
        n0 = int(hp.nr_iter_sphere_fit) if self.reference_schedule else 0
        in_sphere_init = git < n0
        it = git if in_sphere_init else git - n0                      # iter_nr_for_anneal (permuto_sdf_utils.py:80-88)
        if not self._prefetch_valid(git, reel):     # (a prefetched step has been seeded, and its rays drawn, by the step before it)
            if hasattr(self, "_drop_prefetch"):
                self._drop_prefetch()             # a prefetch for another iteration / reel: its jitter draws are rolled back
            parallel.seed_generators(parallel.step_seed(self._seed, parallel.rank(), git), self.dev)  # this rank's rays / jitter
        late = (not in_sphere_init) and it >= hp.iter_start_reduce_curv
        for group in self.opt.param_groups:
            if self.reference_schedule:
                group["lr"] = lr_schedule(git, hp)
            if group.get("name") == "model_rgb_only_encoding":        # train_permuto_sdf.py:400-403: set before the optimiser
                group["weight_decay"] = hp.rgb_lattice_weight_decay_late if late else 0.0   # step of the same iteration
        # (:405) the reference lowers hyperparams.eikonal_weight AFTER this iteration's loss was built: from the next one on
        eikonal_weight = hp.eikonal_weight_late if self._late_seen else hp.eikonal_weight
        n_fg, nr_rays_used = 0, 0
        for p in self.params:
            p.grad = None
        grads_done = False
        if in_sphere_init:
            self._params_ready()
            loss = self._sphere_init_loss(it)
        else:
            # rgb_nr_iters_for_c2f = background_nr_iters_for_c2f = 1 (train_permuto_sdf.py:102-103; models.py:368,496): the colour and
            # background lattices see the window of t = 0.3 at annealing iteration 0 and are fully open from iteration 1 on
            t_col = map_range_val(it, 0.0, 1.0, 0.3, 1.0)
            if t_col != self._colour_window_t:
                w = Coarse2Fine(24)(t_col).to(self.dev)
                self.rgb._win.copy_(w)
                self.bg._win.copy_(w)
                self._colour_window_t = t_col
            loss, n_fg, nr_rays_used, grads_done = self._main_phase(reel, it, git, eikonal_weight)
            if late:
                self._late_seen = True
        # ---- backward (unless the main phase produced the gradients itself: train_manual.ManualTrainer), all-reduce, optimiser
        if not grads_done:
            with self.accumulate_grads():      # the persistent gradient buffers are open for THIS backward only
                loss.backward()
        if self.grad_buffers:
            self.sdf.mlp_sdf.assign_grads()
        buffered = {id(m.encoding.lattice_values) for m in (self.sdf, self.rgb, self.bg)} if self.touched else set()
        dense = [p for p in self.params if id(p) not in buffered]
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in dense]
        for p, g in zip(dense, grads):
            p.grad = g
        if self.capture_grads is not None:      # tests: the gradients of this step as the optimiser is about to see them
            self.capture_grads = {"dense": [g.detach().clone() for g in grads], "lattices": [tr.grad.clone() for tr in self.touched],
                                  "loss": loss.detach().clone()}
        # a lattice gradient that a backward OUTSIDE accumulate_grads() handed to autograd (p.grad) joins the buffer BEFORE the
        # reduction: folded in later (FusedAdamW.step does that for single-process use) it would never be summed over the ranks
        for m, tr in zip((self.sdf, self.rgb, self.bg), self.touched):
            p = m.encoding.lattice_values
            if p.grad is not None:
                tr.grad.add_(p.grad)
                p.grad = None
        owned = None
        dp = parallel.collectives_active()
        if dp:
            buckets = self._dp_buckets if self._dp_buckets is not None else parallel.GradientBuckets()
            self._dp_buckets = None
            small = [g for g in grads if g.numel() < (1 << 20)]
            big = [g for g in grads if g.numel() >= (1 << 20)]
            buckets.reduce(small)
            for g in big:
                buckets.reduce([g])
            # the lattices (50 MB each): reduce-scatter IN PLACE, the owner updates its 1/world of the table, the ranks all-gather
            # the PARAMETERS (parallel.ShardedUpdate); PSDF_DP_OPTIMIZER=replicated (or a table that cannot be cut evenly):
            # reduce-scatter + all-gather of the gradient, every rank updates everything.  Lattices whose gradient was final
            # earlier in the backward are already on their way (_dp_lattice_final).
            for k in range(len(self.touched)):
                self._dp_lattice_final(k)
            owned, mine, sus, sbytes = {}, {}, {}, []
            for k, (m, tr) in enumerate(zip((self.sdf, self.rgb, self.bg), self.touched)):
                su, own = self._dp_started[k]
                if su is None:
                    continue
                p = m.encoding.lattice_values
                mine[k] = (p, own)
                sus[k] = su
                owned[p] = [parallel.shard_bounds(p.numel(), tr.block_elems, rank_=r) for r in su.virtual_ranks()]
            buckets.finish()
            for su in sus.values():
                su.wait()
                sbytes += list(su.bytes)
            self._dp_started = {}
            if self.capture_grads is not None:      # tests: the lattice gradients AFTER the sum over the ranks
                self.capture_grads["lattices_reduced"] = [tr.grad.clone() for tr in self.touched]
        self.opt.step(grad_scale=1.0 / parallel.world_size(), owned=owned or None)
        if owned:
            # the owners' bytes to everybody, in the order the next step reads the tables (background first: its forward is
            # enqueued during the march; then the SDF lattice; the colour lattice last); waited for where they are read
            for k in sorted(mine, key=lambda kk: {2: 0, 0: 1, 1: 2}.get(kk, 3)):
                p, own = mine[k]
                g = parallel.ShardedUpdate()
                g.all_gather(p.data.view(-1), own)
                self._pending_gather[k] = g
            if not self.defer_param_gather:
                self._params_ready()
            gather_bytes = [p.numel() * 4 for p, _ in mine.values()]
            self.last_dp = {"optimizer": "sharded", "sharded_bytes": sbytes, "bucket_bytes": list(buckets.bytes),
                            "gather_bytes": gather_bytes, "deferred_gather": bool(self.defer_param_gather)}
        elif dp:
            self.last_dp = {"optimizer": "replicated", "bucket_bytes": list(buckets.bytes)}
        for gb in self.grad_buffers:
            gb.zero()
        self.iter += 1
        self.last = {"nr_rays": nr_rays_used, "nr_fg_samples": n_fg, "lr": self.opt.param_groups[0]["lr"],
                     "phase": "sphere_init" if in_sphere_init else "train"}
        return loss.detach()
