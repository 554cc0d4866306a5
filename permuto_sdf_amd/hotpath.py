"""The data-parallel hot path as ONE object: encode -> fused SDF MLP -> NeuS compositing, forward and backward, over a
packed batch of ray samples, entirely on the feature-major ([C, N]) fast path of the kernels (no autograd graph,
no transposes).  It strings together the operators the reference's Python strings together in
permuto_sdf_py/train_permuto_sdf.py:111-169 (run_net: SDF eval -> NeuS weights -> integrate) and
permuto_sdf_py/volume_rendering/volume_rendering_modules.py:129-172 (section-point opacity -> transmittance -> weights),
with their backward kernels, and it is a TRUE gradient: dL/d(radiance) flows through integrate -> weights -> transmittance
-> opacity (csrc/neus.hip) -> dL/d(sdf) -> fused MLP backward -> encoding backward.  bench.py times this object; the
drop-in API (permuto_sdf / permutohedral_encoding) exposes the same kernels one by one for the reference's unmodified
Python.
"""
import torch

from . import _lib as L
from . import parallel
from .bridge import VolumeRendering as VR
from .encoding import PermutoEncoding, encode_backward_raw, encode_forward_raw
from .mlp import FusedMLP, f16_forward_supported, mlp_backward_raw, mlp_forward_raw, pack_params
from .neus import (FUSED_MAX_PER_RAY, neus_alpha_backward_raw, neus_alpha_forward_raw, neus_composite_backward_raw,
                   neus_composite_forward_raw)


class SdfHotPath:
    def __init__(self, nr_levels=16, hidden=64, out_channels=1, capacity=2 ** 18, device="cuda", seed=0, lr=1e-3):
        import numpy as np
        torch.manual_seed(seed)
        self.dev = torch.device(device)
        self.enc = PermutoEncoding(3, capacity, nr_levels, 2, np.geomspace(1.0, 1e-4, nr_levels),
                                   appply_random_shift_per_level=True, concat_points=True, concat_points_scaling=1e-3,
                                   init_scale=1e-2).to(self.dev)
        C = self.enc.output_dims()
        self.mlp = FusedMLP([C, hidden, hidden, hidden, out_channels]).to(self.dev)
        with torch.no_grad():  # sphere-ish start so that alphas are neither all 0 nor all 1
            self.mlp.layers[-1].bias.fill_(0.05)
        self.window = torch.ones(nr_levels, device=self.dev)
        # exp(10 * variance) of SingleVarianceNetwork (volume_rendering_modules.py:96-115) at variance 0.5; fixed here
        self.inv_s = torch.full((1,), float(torch.exp(torch.tensor(5.0))), device=self.dev)
        self.cos_anneal_ratio = 1.0
        self.params = [self.enc.lattice_values] + [p for l in self.mlp.layers for p in (l.weight, l.bias)]
        from .optim import FusedAdamW
        self.opt = FusedAdamW(self.params, lr=lr)
        self.events = None
        # fused compositing (one launch per direction) where the container says how long its rays are at most; `want_rgb_grad`:
        # the per-sample radiance is an INPUT of this path (in training it comes from the colour network: its gradient is then
        # wanted); the benchmark's synthetic radiance has no consumer for it
        self.fuse_compositing = True
        self.want_rgb_grad = True
        import os
        self.fwd_f16 = os.environ.get("PSDF_MLP_FWD_SPLIT", "f16") != "bf16"
        # data parallel: the lattice is updated by its owners and the PARAMETERS are all-gathered (parallel.ShardedUpdate)
        self.shard_optimizer = parallel.sharded_optimizer_default()
        self.last_dp = None

    @staticmethod
    def _max_per_ray(rs):
        """an upper bound of the samples per ray known WITHOUT a host sync, or None"""
        if rs.rays_have_equal_nr_of_samples and 0 < int(rs.fixed_nr_of_samples_per_ray) <= FUSED_MAX_PER_RAY:
            return int(rs.fixed_nr_of_samples_per_ray)
        return None

    # ------------------------------------------------------------------ forward
    def forward(self, rs, rgb_samples, normals):
        """rs: RaySamplesPacked (positions, directions, dt, ray ranges); rgb_samples [M,3]; normals [M,3]: the SDF gradient
        direction at the samples, an INPUT here (the reference obtains it with a second, differentiated evaluation of the
        same net, models.py:236-251; config 2 of BASELINE.json times the first-order path).  Returns per-ray radiance [R,3]
        and the tensors the backward needs."""
        cfg = self.enc.cfg
        pos = rs.samples_pos
        # two-piece fp16 forward where it exists (the BASELINE net): its inputs are encoding features and 1e-3-scaled points,
        # far below fp16's range; PSDF_MLP_FWD_SPLIT=bf16 keeps the three-piece bf16 evaluation
        f16 = self.fwd_f16 and f16_forward_supported(self.mlp.dims)
        packed = pack_params(self.mlp.dims, [l.weight for l in self.mlp.layers], [l.bias for l in self.mlp.layers], f16=f16)
        # Two launches on purpose: the level-major encode kernel keeps one 2-MiB table at a time in every XCD's L2 and
        # runs at full occupancy, which measured faster than the single fused launch of csrc/fused.hip at this size
        # (tools/fused_bench.py: 0.38 + 0.67 ms against 1.13 ms); the fused launch is used where its per-sample skip
        # mask pays (sphere tracing).
        feat = encode_forward_raw(cfg, pos, self.enc.lattice_values.detach(), self.enc.scale_factor,
                                  self.enc.random_shift_per_level.detach(), self.window)
        timed = self.events is not None and "mlp_fwd" in self.events
        if timed:
            self.events["mlp_fwd"][0].record()
        sdf = mlp_forward_raw(self.mlp.dims, feat, packed, f16=f16)           # [1, N] feature-major == [N,1] memory
        if timed:
            self.events["mlp_fwd"][1].record()
        sdf_col = sdf.view(-1, 1)
        per_ray = self._max_per_ray(rs)
        if self.fuse_compositing and per_ray is not None:
            # opacity -> transmittance -> weights -> radiance in ONE launch (csrc/composite_fused.hip): the same arithmetic as the
            # four operators below, for callers that own the whole chain (tests/test_gpu_neus.py compares the two)
            pred, bg, _ = neus_composite_forward_raw(rs, sdf_col, normals, rgb_samples, self.inv_s, self.cos_anneal_ratio)
            return pred, dict(feat=feat, packed=packed, sdf=sdf, bg=bg, normals=normals, fused_per_ray=per_ray)
        alpha, one_minus = neus_alpha_forward_raw(sdf_col, rs.samples_dirs, normals, rs.samples_dt, self.inv_s,
                                                  self.cos_anneal_ratio)
        T, bg = VR.cumprod_alpha2transmittance(rs, one_minus)
        w = alpha * T
        pred = VR.integrate_with_weights(rs, rgb_samples, w)
        return pred, dict(feat=feat, packed=packed, sdf=sdf, alpha=alpha, one_minus=one_minus, T=T, bg=bg, w=w,
                          normals=normals)

    # ------------------------------------------------------------------ backward (+ optional all-reduce and optimiser)
    def backward(self, rs, rgb_samples, saved, grad_pred, reduce=True, optimizer_step=True, split_levels=None):
        """dL/d(pred) [R,3] -> gradients of the lattice and the MLP parameters (and of the per-sample radiance).
        integrate_backward -> (w = alpha T) -> transmittance backward (per-ray inverse cumsum + the reference's
        cumprod backward, volume_rendering_funcs.py:55-118) -> opacity backward (csrc/neus.hip) -> dL/d(sdf) -> fused MLP
        backward -> encoding backward."""
        cfg = self.enc.cfg
        N = rs.samples_pos.shape[0]
        if "fused_per_ray" in saved:
            g_sdf, _, g_rgb, _ = neus_composite_backward_raw(rs, saved["fused_per_ray"], grad_pred, None, saved["sdf"].view(-1, 1),
                                                             saved["normals"], rgb_samples, self.inv_s, self.cos_anneal_ratio,
                                                             need_grad=False, need_rgb=self.want_rgb_grad, need_inv_s=False)
            g_om = g_alpha = None
        else:
            g_rgb, g_w = VR.integrate_with_weights_backward(grad_pred, rs, rgb_samples, saved["w"], None)
            g_T = g_w * saved["alpha"]
            cs = VR.cumsum_over_each_ray(rs, g_T * saved["T"], True)
            g_om = VR.cumprod_alpha2transmittance_backward(g_T, torch.zeros_like(saved["bg"]), rs, saved["one_minus"], saved["T"],
                                                           saved["bg"], cs)
            g_alpha = torch.addcmul(-g_om, g_w, saved["T"])      # alpha enters as w = alpha T and as 1 - alpha + 1e-7
            g_sdf, _, _ = neus_alpha_backward_raw(g_alpha, saved["sdf"].view(-1, 1), rs.samples_dirs, saved["normals"],
                                                   rs.samples_dt, self.inv_s, self.cos_anneal_ratio, need_grad=False,
                                                   need_inv_s=False)
        grad_sdf = g_sdf.view(1, N)
        if self.events is not None:
            self.events["mlp_bwd"][0].record()
        d_feat, dWs, dbs = mlp_backward_raw(self.mlp.dims, saved["feat"], [l.weight for l in self.mlp.layers],
                                            [l.bias for l in self.mlp.layers], grad_sdf, need_dx=True)
        if self.events is not None:
            self.events["mlp_bwd"][1].record()
        buckets = parallel.GradientBuckets()
        su = parallel.ShardedUpdate()
        sharded = reduce and optimizer_step and self.shard_optimizer and parallel.collectives_active()
        lat_owned, lat_mine = [], []           # element ranges of the flat lattice: updated here / all-gathered from here
        per_level = self.enc.lattice_values[0].numel()

        def reduce_lattice(l0, l1):
            """the gradient of levels [l0, l1) is final: send it on its way"""
            if not reduce:
                return
            own = su.reduce_scatter(g_lat[l0:l1].view(-1), unit=4) if sharded else None
            base = l0 * per_level
            if own is None:                    # replicated for this range: all ranks get the sum and update all of it
                buckets.reduce([g_lat[l0:l1]])
                if sharded:
                    lat_owned.append((base, l1 * per_level))
            else:
                n = (l1 - l0) * per_level
                lat_owned.extend((base + lo, base + hi) for lo, hi in
                                 (parallel.shard_bounds(n, 4, rank_=r) for r in su.virtual_ranks()))
                lat_mine.append((base, l1 * per_level, own))
        if reduce:
            buckets.reduce(dWs + dbs)          # small bucket first: overlaps the encoding backward
        g_lat = torch.zeros_like(self.enc.lattice_values)
        if self.events is not None:
            self.events["enc_bwd"][0].record()
        if split_levels is None:
            split_levels = reduce and parallel.collectives_active()
        if split_levels:
            # Data parallel: the lattice gradient (4*L*T*F bytes, the same 2 MiB for every level) is the only large message
            # of the step.  The levels are independent, so the backward runs as two launches over level ranges and the
            # all-reduce of the first range travels over xGMI while the second range is still being computed.  Where to
            # cut (tools/enc_bwd_ranges.py, 2 M points, 16 levels; one launch 1.14 ms): the binning kernel needs many levels
            # in flight to fill the chip, so two equal halves cost +0.26 ms, a 9/7 cut +0.25 ms, but a 6/10 cut only
            # +0.06 ms -- the first six levels are the cheap, coarse ones (0.15 ms), and the 10 expensive ones that hide
            # their 12.6 MB of traffic still run as one launch.  With the all-reduce at B GB/s the step pays
            # 0.06 ms + 21 MB / B instead of 33.5 MB / B unsplit: ahead for every B below ~200 GB/s.
            L_ = cfg.nr_levels
            cut = max(1, min(L_ - 1, (3 * L_ + 4) // 8))
            for l0, l1 in ((0, cut), (cut, L_)):
                self._encode_backward_levels(rs.samples_pos, d_feat, g_lat, l0, l1)
                reduce_lattice(l0, l1)
        else:
            encode_backward_raw(cfg, rs.samples_pos, self.enc.lattice_values, self.enc.scale_factor,
                                self.enc.random_shift_per_level, self.window, d_feat, g_lat, None)
            reduce_lattice(0, cfg.nr_levels)
        if self.events is not None:
            self.events["enc_bwd"][1].record()
        if reduce:
            # what the step WAITS for communication: from the end of the last backward kernel to the last bucket's arrival
            # (0 when the all-reduce is hidden behind the encode backward; bench.py reports it per rank)
            if self.events is not None and "comm_wait" in self.events:
                self.events["comm_wait"][0].record()
            buckets.finish()
            su.wait()
            if self.events is not None and "comm_wait" in self.events:
                self.events["comm_wait"][1].record()
            self.last_bucket_bytes = list(buckets.bytes) + list(su.bytes)
        grads = [g_lat] + [t for pair in zip(dWs, dbs) for t in pair]
        if optimizer_step:
            for p, g in zip(self.params, grads):
                p.grad = g
            lat = self.enc.lattice_values
            self.opt.step(grad_scale=1.0 / parallel.world_size(), owned={lat: lat_owned} if (sharded and lat_owned) else None)
            if sharded and lat_mine:           # the owners' bytes to everybody (g_lat outside a rank's own ranges is NOT the sum)
                flat = lat.data.view(-1)
                for b0, b1, own in lat_mine:
                    su.all_gather(flat[b0:b1], own)
                su.wait()
            self.last_dp = {"optimizer": "sharded" if (sharded and lat_mine) else "replicated"}
        return dict(g_rgb=g_rgb, g_one_minus=g_om, g_alpha=g_alpha, g_sdf=g_sdf, grads=grads)

    def _encode_backward_levels(self, pos, d_feat, g_lat, l0, l1):
        """lattice gradient of levels [l0, l1) only: every operand of the kernel is contiguous per level (tables
        [L,T,F], constants [L,P], window [L], feature-major upstream gradient [2L+.., N]), so a level range is just a
        set of offset pointers and the C ABI needs no extra entry point."""
        from .encoding import _Cfg
        c = self.enc.cfg
        sub = _Cfg(c.pos_dim, c.capacity, l1 - l0, c.nr_feat, False, 1.0)
        F = c.nr_feat
        encode_backward_raw(sub, pos, self.enc.lattice_values[l0:l1], self.enc.scale_factor[l0:l1],
                            self.enc.random_shift_per_level[l0:l1], self.window[l0:l1], d_feat[F * l0:F * l1],
                            g_lat[l0:l1], None)

    def step(self, rs, rgb_samples, normals, gt, **kw):
        """forward -> L1 radiance loss against `gt` [R,3] (rgb_loss, permuto_sdf_utils.py:43-47) -> backward"""
        from .neus import l1_loss_raw
        pred, saved = self.forward(rs, rgb_samples, normals)
        loss, g_pred = l1_loss_raw(pred, gt)
        out = self.backward(rs, rgb_samples, saved, g_pred, **kw)
        out["loss"] = loss
        return pred, saved, out
