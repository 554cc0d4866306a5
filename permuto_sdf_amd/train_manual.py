"""BASELINE config 4 with a HAND-WRITTEN backward: the same optimisation step as `train_step.Trainer` (same networks, samplers,
losses, schedule, optimiser, data-parallel reduction -- it IS a Trainer), but the main phase's forward and backward are
written out over the raw feature-major kernels instead of being recorded and replayed by torch autograd.

Why: the step is host bound (train_step.py): ~250 launches, the autograd engine's bookkeeping
(`run_backward` alone is a third of the host time), [N, C] <-> [C, N] glue around every Function, gradient-accumulation adds
and zero fills that autograd inserts.  Written by hand, every tensor stays feature-major, every gradient is produced once where
it is needed, the three evaluations of the SDF network share one parameter image, and the four contributions to the SDF
network's parameter gradient land in one buffer.

What is differentiated (train_permuto_sdf.py:311-429, run_net :111-169; names as in train_step.py):
  fg samples -> SDF net: y = mlp(enc(p)) -> sdf = y[0], geom = y[1:33];  n = d sdf / d p  (MLP data backward of e0, then the
  encoding's position backward);  colour = sigmoid(Lipshitz-MLP([enc2(p), SH5(dir), normalize(n), geom]));
  (radiance, bgT) = NeuS(sdf, n, colour);  bg samples -> NerfHash -> radiance_bg;  pred = radiance + bgT * radiance_bg;
  loss = L1(pred, gt) + w_e eikonal(n) + w_c curvature(n, n(p + eps cross(norm n, norm r))) + w_o offsurface(sdf(p_off)) [+ Lipschitz].
The backward of  n = d sdf / d p  is the double backward: `psdf_encode_double_backward` (gradient w.r.t. the lattice and w.r.t.
the feature gradient) -> `psdf_mlp_double_backward` (gradient w.r.t. features and parameters) -> the encoding's lattice backward.
`tests/test_gpu_train_step.py::test_manual_backward_equals_autograd` holds every gradient of this file against the autograd
trainer's on the same batch.
"""
import contextlib
import os

import torch

from . import _lib as L
from .bridge import PermutoSDF, RaySamplesPacked, VolumeRendering as VR
from .encoding import encode_backward_raw, encode_double_backward_raw, encode_forward_raw
from .mlp import (_grad_views, double_backward_plus_supported, mlp_forward_wide_f16_raw, lipshitz_normalize_all_backward_raw, lipshitz_normalize_all_raw, mlp_backward_raw,
                  mlp_double_backward, mlp_forward_raw, pack_params)
from .neus import (eikonal_loss_raw, l1_loss_raw, nerf_composite_backward_raw, nerf_composite_forward_raw, neus_composite_backward_raw,
                   neus_composite_forward_raw, sigmoid_rows_backward_raw, sigmoid_rows_raw)
from .train_step import Trainer, map_range_val


def _raw(enc):
    """(cfg, lattice, scale factors, shifts, touched-rows record) of an encoding as the raw kernels take them.  Kept on the module:
    a step asks 17 times, and every ask is five nn.Module attribute lookups and two detach() calls on the host -- the step is host
    bound at iteration 0 (tools/prof_step.py).  Rebuilt when the parameter's storage has moved (.to(), load of a new tensor) or the
    touched-rows record was replaced."""
    lat = enc._parameters["lattice_values"]
    r = enc.__dict__.get("_psdf_raw")
    if r is None or r[5] != lat.data_ptr() or r[4] is not enc.__dict__.get("touched_rows"):
        r = (enc.cfg, lat.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), enc.touched_rows, lat.data_ptr())
        enc.__dict__["_psdf_raw"] = r
    return r


def _enc_fwd(enc, pts, win, train=True, out=None):
    cfg, lat, sf, sh, tr, _ = _raw(enc)
    return encode_forward_raw(cfg, pts, lat, sf, sh, win, out=out, touched=tr.touched if train else None,
                              block_rows_log2=tr.block_rows_log2)


def _enc_bwd(enc, pts, win, g_fm, want_pos=False, want_lattice=True, zeroed=None):
    """lattice gradient into the encoding's persistent buffer; optionally the position gradient [N, P] (the kernels accumulate:
    `zeroed` = a zero-filled [N, P] tensor to use for it, else one is filled here)"""
    g_pos = (zeroed if zeroed is not None else torch.zeros_like(pts)) if want_pos else None
    cfg, lat, sf, sh, tr, _ = _raw(enc)
    encode_backward_raw(cfg, pts, lat, sf, sh, win, g_fm, tr.grad if want_lattice else None, g_pos)
    return g_pos


def _enc_dbl_gather(enc, pts, win, dd_pos, g_fm):
    """backward of the position gradient, first half: the gradient w.r.t. the feature gradient [C, N] (a gather)"""
    gg = torch.empty_like(g_fm)
    cfg, lat, sf, sh, _, _ = _raw(enc)
    encode_double_backward_raw(cfg, pts, lat, sf, sh, win, dd_pos, g_fm, None, gg)
    return gg


def _enc_dbl_scatter(enc, pts, win, dd_pos, g_fm, direct_fm):
    """second half, together with the plain backward of `direct_fm` (the gradient that reached the features): ONE scatter of both
    into the lattice buffer -- they land on the same rows of the same simplices"""
    cfg, lat, sf, sh, tr, _ = _raw(enc)
    encode_double_backward_raw(cfg, pts, lat, sf, sh, win, dd_pos, g_fm, tr.grad, None, direct_fm)


def _normalize3(x, gy=None):
    out = torch.empty_like(x)
    L.call("psdf_normalize3", L.c_l(x.shape[0]), L.ptr(x), L.ptr(gy), L.ptr(out), L.stream())
    return out


def _set_grads(layers, dWs, dbs):
    for l, dW, db in zip(layers, dWs, dbs):
        l.weight.grad, l.bias.grad = dW, db


class ManualTrainer(Trainer):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        assert self.touched, "ManualTrainer accumulates the lattice gradients in the touched-rows buffers"
        # second stream for the background branch: OFF by default -- measured slower (see _main_phase); PSDF_TRAIN_STREAMS=1 turns it on
        self.overlap_streams = os.environ.get("PSDF_TRAIN_STREAMS", "0") == "1"
        self._side = None
        self._g_yo = None
        # the NEXT step's rays, sphere intersection, occupancy march and background samples are issued on a side stream as soon
        # as this step's grid refresh is enqueued, and run beside this step's backward (PSDF_TRAIN_PREFETCH=0: off)
        self.prefetch_sampling = os.environ.get("PSDF_TRAIN_PREFETCH", "1") != "0"
        self._prefetched = None
        self._prefetch_side = None
        self._pos_pool = None
        # the SDF net's plain backward (g_y) inside the double backward's launch (g_n); PSDF_TRAIN_FUSE_SDF_BWD=0: two launches
        self.fuse_sdf_backward = os.environ.get("PSDF_TRAIN_FUSE_SDF_BWD", "1") != "0"
        self._events = [torch.cuda.Event() for _ in range(4)] if self.dev.type == "cuda" else []

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
        return self._side

    # ------------------------------------------------------------------ next step's sampling, first half, ahead of time
    def _prefetch_valid(self, git, reel=None):
        """reel: the image reel the coming step samples from (None: not checked).  A prefetch drawn from ANOTHER reel must be
        dropped by the step's own validity test too -- step() skips its seeding on the strength of this answer."""
        pf = self._prefetched
        return (pf is not None and pf["git"] == git and pf["nr_rays"] == self.nr_rays and self._hand_written_step_applies()
                and (reel is None or pf["reel"] == id(reel))
                and "_draw_rays" not in self.__dict__ and "_samples" not in self.__dict__)

    def _drop_prefetch(self):
        """A prefetch that does not fit the coming step is undone as far as the random streams go: the jitter generators
        (process-global PCG32 streams of the occupancy grid / ray sampler / volume renderer, advanced once by the prefetched
        launches) return to their state before it.  The caller re-seeds torch's generators for its iteration."""
        pf, self._prefetched = self._prefetched, None
        if pf is not None and pf.get("pcg") is not None:
            from .bridge import OccupancyGrid, RaySampler, VolumeRendering
            owners = (OccupancyGrid, RaySampler, VolumeRendering)
            # only where the stream still stands where the prefetch left it: a caller that has set the generators itself since
            # (tools/reference_step_parity.py copies the reference's states in) keeps what it set
            for cls, (st, inc), after in zip(owners, pf["pcg"], pf["pcg_after"]):
                if (cls._rng.state, cls._rng.inc) == after:
                    cls._rng.state, cls._rng.inc = st, inc

    def _launch_prefetch(self, reel, next_git):
        """The march is the largest kernel of the step (0.21 ms) and runs on 12 waves: nothing it needs -- the grid, the image
        reel, the ray count -- changes after `_refresh_and_adapt`, and nothing else of the step draws from torch's generators
        after that point.  So the generators are seeded for the next iteration HERE, its rays are drawn, intersected and marched
        and its background samples placed on a side stream while this step's backward and optimiser run; the next step skips
        its own seeding and starts from the generator state this leaves -- the very state it would have had after drawing its
        rays itself.  The side stream first waits for the main stream (the refreshed grid), the next step's main stream waits
        for the side stream's event.  Tensors allocated here live in the side stream's pool and are next reused by the NEXT
        prefetch, whose work again starts behind a fresh event of the main stream: no kernel of either stream can still be
        reading them.  A prefetch that does not fit the step that comes (iteration counter or ray count changed from outside,
        samplers patched by a test) is dropped."""
        if (not self.prefetch_sampling or self.dev.type != "cuda" or "_draw_rays" in self.__dict__
                or "_samples" in self.__dict__):
            self._prefetched = None
            return
        from . import parallel
        main = torch.cuda.current_stream(self.dev)
        if self._prefetch_side is None:
            self._prefetch_side = torch.cuda.Stream(device=self.dev)
        side = self._prefetch_side
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        from .bridge import OccupancyGrid, RaySampler, VolumeRendering
        pcg = [(c._rng.state, c._rng.inc) for c in (OccupancyGrid, RaySampler, VolumeRendering)]     # (see _drop_prefetch)
        parallel.seed_generators(parallel.step_seed(self._seed, parallel.rank(), next_git), self.dev)
        with torch.cuda.stream(side), torch.no_grad():
            rays = self._draw_rays(reel)
            begun = self._samples_begin(rays[0], rays[1], True)
            done = torch.cuda.Event()
            done.record(side)
        pcg_after = [(c._rng.state, c._rng.inc) for c in (OccupancyGrid, RaySampler, VolumeRendering)]
        self._prefetched = dict(git=next_git, nr_rays=self.nr_rays, reel=id(reel), rays=rays, begun=begun, done=done, pcg=pcg,
                                pcg_after=pcg_after)

    def _hand_written_step_applies(self):
        """The hand-written step is built on the fused compositing kernels (at most 256 samples per ray: foreground
        max_nr_samples_per_ray + 2 * nr_samples_imp_sampling, background nr_samples_bg) and on the reference's default mode
        (background network, no mask loss).  Anything else runs `Trainer._main_phase`: the same step as an autograd graph over
        the per-operator chain -- slower, any ray length, `--with_mask` included."""
        hp = self.hp
        return (not self.with_mask and hp.nr_samples_bg <= 256
                and hp.max_nr_samples_per_ray + 2 * hp.nr_samples_imp_sampling <= 256)

    # ------------------------------------------------------------------ the SDF network, one evaluation
    def _sdf_gradient(self, feat, pts, win, ws, bs):
        """n = d sdf / d p and the feature gradient it came through"""
        dims = self.sdf.mlp_sdf.dims
        e0 = None       # "the unit gradient of output 0": the kernels take NULL for it (no [33, N] tensor that is 1 in one row)
        dfeat, _, _ = mlp_backward_raw(dims, feat, ws, bs, e0, need_dx=True, need_dw=False)
        n = _enc_bwd(self.sdf.encoding, pts, win, dfeat, want_pos=True, want_lattice=False, zeroed=self._zeroed_pos(pts))
        return n, dfeat, e0

    def _zeroed_pos(self, pts):
        """a zero-filled [N, 3] tensor for a position gradient: the step needs three of the same size (normals of the samples, of
        the shifted samples, gradient of the shifted points) -- ONE fill launch for the three instead of one each"""
        pool = self._pos_pool
        if pool is None or pool[0].shape[1:] != pts.shape or pool[1] >= pool[0].shape[0]:
            pool = self._pos_pool = [torch.zeros((3,) + tuple(pts.shape), dtype=torch.float32, device=pts.device), 0]
        pool[1] += 1
        return pool[0][pool[1] - 1]

    def _grad_arena(self):
        """zero-filled parameter-gradient views for the three nets without a persistent gradient buffer (background density net,
        background colour head, colour network's normalised weights) and the four Lipschitz bounds: ONE fill launch per step
        instead of four (the kernels accumulate into what they are given) -> (bg1, bg2, rgb, dc_flat), each (dWs, dbs)"""
        lay = self.__dict__.get("_arena_layout")
        if lay is None:
            def need(dims):
                return sum(((dims[l + 1] * dims[l] + 3) & ~3) + ((dims[l + 1] + 3) & ~3) for l in range(len(dims) - 1))
            d1, d2, dc = self.bg.mlp_feat_and_density.dims, self.bg.mlp_rgb.dims, self.rgb.mlp.dims
            offs = [0]
            for sz in (need(d1), need(d2), need(dc), 4 * (len(dc) - 1)):
                offs.append(offs[-1] + sz)
            lay = self._arena_layout = (offs, d1, d2, dc)
        offs, d1, d2, dc = lay
        flat = torch.zeros(offs[-1], dtype=torch.float32, device=self.dev)
        views = [_grad_views(d, flat=flat[offs[i]:offs[i + 1]])[1:] for i, d in enumerate((d1, d2, dc))]
        return views[0], views[1], views[2], flat[offs[3]:offs[4]]

    def _sdf_gradient_backward(self, g_n, feat, dfeat, e0, pts, win, ws, bs, gb, want_pos=False, extra_dfeat=None, extra_gy=None):
        """backward of  n = d sdf / d p  for an upstream g_n [N,3]: lattice and parameter gradients are accumulated; returns the
        position gradient when asked (the shifted points of the curvature term depend on n).  extra_gy [33, N]: an upstream
        gradient of the net's outputs on the same samples -- its plain backward rides in the double backward's launch (round 6:
        one forward recomputation and one sweep for both); extra_dfeat: a data gradient to add instead (the two-launch form)"""
        enc, dims = self.sdf.encoding, self.sdf.mlp_sdf.dims
        gg = _enc_dbl_gather(enc, pts, win, g_n, dfeat)
        dX2, _, _ = mlp_double_backward(dims, feat, ws, bs, e0, gg, into=(gb.dWs, gb.dbs), module=self.sdf.mlp_sdf, gy2_fm=extra_gy)
        if extra_dfeat is not None:
            dX2 = dX2 + extra_dfeat
        _enc_dbl_scatter(enc, pts, win, g_n, dfeat, dX2)
        return _enc_bwd(enc, pts, win, dX2, want_pos=True, want_lattice=False, zeroed=self._zeroed_pos(pts)) if want_pos else None

    # ------------------------------------------------------------------ the background branch (NerfHash, models.py:431-526)
    def _bg_forward(self, bg, calib):
        """4-D lattice -> density + feature net -> [gelu(features), SH4(dir)] -> colour head -> (calibrated) sigmoid.  Everything
        the compositing and the backward need, as a dict."""
        bgn = self.bg
        self._params_ready(2)              # the background lattice's parameters (all-gather of the previous step, data parallel)
        M = bg.samples_pos_4d.shape[0]
        p4, dirs_b = bg.samples_pos_4d, bg.samples_dirs
        feat4 = _enc_fwd(bgn.encoding, p4, bgn._win)
        l1b = list(bgn.mlp_feat_and_density.layers)
        w1b, b1b = [l.weight for l in l1b], [l.bias for l in l1b]
        d1 = bgn.mlp_feat_and_density.dims
        fd = mlp_forward_wide_f16_raw(d1, feat4, w1b, b1b)                                # [65, M] on the fp16 matrix pipe (round 6)
        if fd is None:      # (the library declined: fp32 MFMAs)
            fd = mlp_forward_raw(d1, feat4, pack_params(d1, w1b, b1b))
        gel = torch.nn.functional.gelu(fd[1:65])
        sh4 = PermutoSDF.spherical_harmonics(dirs_b, 4)
        x2 = torch.cat([gel, sh4.t()], 0)                                                 # [80, M]
        l2b = list(bgn.mlp_rgb.layers)
        w2b, b2b = [l.weight for l in l2b], [l.bias for l in l2b]
        d2 = bgn.mlp_rgb.dims
        rgbb_fm = mlp_forward_raw(d2, x2, pack_params(d2, w2b, b2b))                      # [3, M]
        B = dict(p4=p4, feat4=feat4, fd=fd, x2=x2, l1b=l1b, w1b=w1b, b1b=b1b, d1=d1, l2b=l2b, w2b=w2b, b2b=b2b, d2=d2)
        if calib is not None:
            cw, cb = calib
            B["rgbb_raw"] = rgbb_fm.t().contiguous()                                      # [M, 3]
            B["ridx"] = RaySamplesPacked.compute_per_sample_ray_idx(bg.ray_start_end_idx, M).long()
            B["rgbb"] = torch.sigmoid(B["rgbb_raw"] * cw.index_select(0, B["ridx"]) + cb.index_select(0, B["ridx"]))
        else:
            B["rgbb"] = sigmoid_rows_raw(rgbb_fm)
        B["raw_den"] = fd[0]                                                              # [M], a row of the feature-major output
        return B

    def _bg_backward(self, B, g_raw, g_rgbb, calib, R):
        """-> (g_cw, g_cb) of the colour calibration (or None, None); the networks' gradients are set / accumulated"""
        bgn = self.bg
        g_cw = g_cb = None
        rgbb = B["rgbb"]
        if calib is not None:
            cw, _ = calib
            g_pre_b = g_rgbb * rgbb * (1.0 - rgbb)                                        # sigmoid
            g_cb = torch.zeros(R, 3, device=self.dev).index_add_(0, B["ridx"], g_pre_b)
            g_cw = torch.zeros(R, 3, device=self.dev).index_add_(0, B["ridx"], g_pre_b * B["rgbb_raw"])
            g_pre_b_fm = (g_pre_b * cw.index_select(0, B["ridx"])).t().contiguous()
        else:
            g_pre_b_fm = sigmoid_rows_backward_raw(g_rgbb, rgbb)                          # [3, M], one launch
        dX2b, dW2b, db2b = mlp_backward_raw(B["d2"], B["x2"], B["w2b"], B["b2b"], g_pre_b_fm, need_dx=True, into=B.get("into2"))
        _set_grads(B["l2b"], dW2b, db2b)
        g_fd = torch.cat([g_raw.view(1, -1), torch.ops.aten.gelu_backward(dX2b[:64], B["fd"][1:65])], 0)      # [65, M]
        dX4, dW1b, db1b = mlp_backward_raw(B["d1"], B["feat4"], B["w1b"], B["b1b"], g_fd, need_dx=True, into=B.get("into1"))
        _set_grads(B["l1b"], dW1b, db1b)
        _enc_bwd(bgn.encoding, B["p4"], bgn._win, dX4)
        return g_cw, g_cb

    # ------------------------------------------------------------------ one iteration of the main phase
    def _main_phase(self, reel, it, git, eikonal_weight):
        """One stream by default.  PSDF_TRAIN_STREAMS=1 (round 4, measured and NOT kept as the default): the background branch is
        independent of the SDF / colour branch between the samplers and the composition, and again between the composition's
        backward and the optimiser, so its forward can run on a side stream beside the foreground forward, join for
        `nerf_composite` (+ losses), and its backward beside the foreground backward.  On MI355X that LOST 9 %: 427 -> 389 it/s
        at iteration 0, 400 -> 370 with every level open (profiles/r04_train_streams_ab.jsonl) -- the step is balanced between
        host and device (~2.2 ms of launches against ~2.25 ms of kernels), and the stream switches and event records cost the
        host more than the overlapped ~0.2 ms of small kernels give back.  (Cross-stream tensors live until this function
        returns, after both streams have joined: the caching allocator's per-stream pools never hand a block to new work that
        an unfinished kernel of the other stream still reads.)"""
        if not self._hand_written_step_applies():
            return Trainer._main_phase(self, reel, it, git, eikonal_weight)
        hp, dev = self.hp, self.dev
        cos_r = map_range_val(it, 0.0, hp.forced_variance_finish_iter, 0.0, 1.0)
        forced_variance = map_range_val(it, 0.0, hp.forced_variance_finish_iter, 0.3, hp.forced_variance_finish)
        main = torch.cuda.current_stream(dev)
        side = self._side_stream() if self.overlap_streams else None

        def fork(ev):
            if side is not None:
                ev.record(main)
                side.wait_event(ev)

        def join(ev):
            if side is not None:
                ev.record(side)
                main.wait_event(ev)
        side_ctx = (lambda: torch.cuda.stream(side)) if side is not None else contextlib.nullcontext
        with torch.no_grad():
            begun = None
            if self._prefetch_valid(git, reel):
                pf, self._prefetched = self._prefetched, None
                main.wait_event(pf["done"])
                (o, d, gt, hit, img_idx, _), begun = pf["rays"], pf["begun"]
            else:
                # (step() has seeded this iteration itself in that case: it asks the same question with the same reel)
                self._drop_prefetch()
                o, d, gt, hit, img_idx, _ = self._draw_rays(reel)
            R = o.shape[0]
            cc = self.colorcal
            calib = None
            if cc is not None:              # per-ray calibration of both branches (models.py:384-385,523-524)
                cam = img_idx.long()
                fixed = (cam == cc.idx_with_fixed_calib)[:, None]
                cw = torch.where(fixed, torch.ones_like(cc.weight_delta[:1]), 1.0 + cc.weight_delta.index_select(0, cam))
                cb = torch.where(fixed, torch.zeros_like(cc.bias[:1]), cc.bias.index_select(0, cam))
                calib = (cw, cb)
            # one stream: the background network's forward is enqueued while the host waits for the march's sample counts (it
            # needs the background samples only), so the step's one host sync leaves no bubble on the GPU
            early_bg = {}
            fg, bg = self._samples(o, d, it, True, begun=begun, between=None if side is not None else
                                   (lambda bg_: early_bg.__setitem__("B", self._bg_forward(bg_, calib))))
            n_fg = fg.samples_pos.shape[0]
            self._params_ready(0)          # the SDF lattice's parameters (all-gather of the previous step, data parallel)
            sdfn, rgbn = self.sdf, self.rgb
            gb = self.grad_buffers[0]
            lin = list(sdfn.mlp_sdf.layers)
            ws, bs = [l.weight for l in lin], [l.bias for l in lin]
            dims_s = sdfn.mlp_sdf.dims
            win = sdfn.window(it).contiguous()
            packed_s = pack_params(dims_s, ws, bs)
            # exp(10 v) clipped, as RgbNet.neus_render computes it: evaluated in float32 on the host and uploaded (one 4-byte copy
            # instead of a copy + exp + clamp on the device; the schedule is a host scalar anyway)
            inv_s = torch.exp(torch.tensor(float(forced_variance) * 10.0, dtype=torch.float32, device="cpu")).clip(1e-6, 1e6).view(1).to(dev)
            rgbn.last_inv_s = inv_s.view(())
            loss = L.zeroed_scalar(dev)     # ONE accumulator: every loss kernel of the step adds its (already weighted) term to it
            self._pos_pool = None           # (zero-filled position gradients: a fresh pool per step)
            arena = self._grad_arena()      # zero-filled parameter gradients of the nets without a persistent buffer: one fill
            # ================================================================= forward
            if "B" in early_bg:
                B = early_bg["B"]
            else:
                fork(self._events[0])
                with side_ctx():
                    B = self._bg_forward(bg, calib)
            if n_fg:
                pts, dirs = fg.samples_pos, fg.samples_dirs
                feat = _enc_fwd(sdfn.encoding, pts, win)
                y = mlp_forward_raw(dims_s, feat, packed_s)                                   # [33, N]: sdf, geometry features
                n, dfeat, e0 = self._sdf_gradient(feat, pts, win, ws, bs)
                # colour network
                self._params_ready(1)
                feat2 = _enc_fwd(rgbn.encoding, pts, rgbn._win)
                sh = PermutoSDF.spherical_harmonics(dirs, 5)
                nn = _normalize3(n)
                x_rgb = torch.cat([feat2, sh.t(), nn.t(), y[1:]], 0)                         # [111, N]
                c_enc, c_sh = feat2.shape[0], feat2.shape[0] + sh.shape[1]
                m = rgbn.mlp
                wn = lipshitz_normalize_all_raw(m.weights_per_layer, m.lipshitz_bound_per_layer)      # all four layers, one launch
                bsr = [b.detach() for b in m.biases_per_layer]
                rgb_fm = mlp_forward_wide_f16_raw(m.dims, x_rgb, wn, bsr)                                        # [3, N]
                if rgb_fm is None:      # (the library declined: fp32 MFMAs)
                    rgb_fm = mlp_forward_raw(m.dims, x_rgb, pack_params(m.dims, wn, bsr))
                if cc is not None:
                    rgb_raw = rgb_fm.t().contiguous()                                                            # [N, 3]
                    ridx_fg = RaySamplesPacked.compute_per_sample_ray_idx(fg.ray_start_end_idx, n_fg).long()
                    rgb = torch.sigmoid(rgb_raw * cw.index_select(0, ridx_fg) + cb.index_select(0, ridx_fg))
                else:
                    rgb = sigmoid_rows_raw(rgb_fm)
                per_ray = hp.max_nr_samples_per_ray + 2 * hp.nr_samples_imp_sampling
                sdf_col = y[0].view(-1, 1)
                pred_fg, bgT, _ = neus_composite_forward_raw(fg, sdf_col, n, rgb, inv_s, cos_r)
            else:
                pred_fg = torch.zeros(R, 3, device=dev)
                bgT = torch.ones(R, 1, device=dev)
            join(self._events[1])
            raw_den, rgbb = B["raw_den"], B["rgbb"]
            # softplus -> opacity -> transmittance -> weights -> background radiance -> pred = pred_fg + bgT * pred_bg: one launch
            _, pred = nerf_composite_forward_raw(bg, raw_den, rgbb, pred_fg, bgT)
            # ---- losses (forward values; their gradients are produced by the same launches)
            _, g_pred = l1_loss_raw(pred, gt, hit, loss=loss)
            # pred = pred_fg + bgT * pred_bg and the whole background compositing backward, one launch; then the background
            # networks' backward leaves for the side stream while this one goes on with the SDF losses
            g_raw, g_rgbb, g_bgT = nerf_composite_backward_raw(bg, hp.nr_samples_bg, g_pred, raw_den, rgbb, bgT)
            fork(self._events[2])
            B["into1"], B["into2"] = arena[0], arena[1]
            with side_ctx():
                g_cw_bg, g_cb_bg = self._bg_backward(B, g_raw, g_rgbb, calib, R)
            if side is None:
                self._dp_lattice_final(2)      # the background lattice's gradient is final: its reduction overlaps what follows
            g_n = None
            curv = None
            if n_fg:
                _, g_n = eikonal_loss_raw(n, scale=eikonal_weight / n_fg, loss=loss)
                gw = map_range_val(it, hp.iter_start_reduce_curv, hp.iter_finish_reduce_curv, 1.0, 0.0)
                if gw > 0.0:
                    rnd = torch.randn_like(pts)
                    shifted = torch.empty_like(pts)
                    L.call("psdf_curvature_shift", L.c_l(n_fg), L.ptr(pts), L.ptr(n), L.ptr(rnd), L.c_f(1e-4), None, L.ptr(shifted),
                           L.stream())
                    feat_s = _enc_fwd(sdfn.encoding, shifted, win)
                    n2, dfeat_s, _ = self._sdf_gradient(feat_s, shifted, win, ws, bs)
                    ga, gb2 = torch.empty_like(n), torch.empty_like(n2)
                    L.call("psdf_curvature_loss", L.c_l(n_fg), L.ptr(n), L.ptr(n2), L.c_f(hp.curvature_weight * gw / n_fg), L.ptr(loss),
                           L.ptr(ga), L.ptr(gb2), L.stream())
                    curv = (rnd, shifted, feat_s, dfeat_s, ga, gb2)
            off = self.sphere.rand_points_inside(1024)
            feat_o = _enc_fwd(sdfn.encoding, off, win)
            y_o = mlp_forward_raw(dims_s, feat_o, packed_s)
            # the off-surface term's gradient w.r.t. the net's 33 outputs is zero except in row 0 (the SDF): a persistent
            # [33, 1024] buffer whose rows 1.. stay zero, row 0 rewritten by the loss kernel every step (no fill, no copy)
            if self._g_yo is None:
                self._g_yo = torch.zeros((dims_s[-1], 1024), dtype=torch.float32, device=dev)
            g_so = self._g_yo[0]
            L.call("psdf_offsurface_loss", L.c_l(1024), L.ptr(y_o[0]), L.c_f(1e2), L.c_f(hp.offsurface_weight / 1024.0),
                   L.ptr(loss), L.ptr(g_so), L.stream())
            self._refresh_and_adapt(it, git, n_fg)
            self._launch_prefetch(reel, git + 1)

            # ================================================================= backward (foreground; the background's is under way)
            g_cw = g_cb = None
            if n_fg:
                g_sdf, g_nc, g_rgb, _ = neus_composite_backward_raw(fg, per_ray, g_pred.contiguous(), g_bgT.contiguous(), sdf_col, n,
                                                                    rgb, inv_s, cos_r, need_grad=True, need_rgb=True, need_inv_s=False)
                if cc is not None:
                    g_pre = g_rgb * rgb * (1.0 - rgb)
                    g_cb = torch.zeros(R, 3, device=dev).index_add_(0, ridx_fg, g_pre)
                    g_cw = torch.zeros(R, 3, device=dev).index_add_(0, ridx_fg, g_pre * rgb_raw)
                    g_pre_fm = (g_pre * cw.index_select(0, ridx_fg)).t().contiguous()
                else:
                    g_pre_fm = sigmoid_rows_backward_raw(g_rgb, rgb)
                dXr, dWn, dbr = mlp_backward_raw(m.dims, x_rgb, wn, bsr, g_pre_fm, need_dx=True, into=arena[2])
                dws, dcs = lipshitz_normalize_all_backward_raw(m.weights_per_layer, m.lipshitz_bound_per_layer, dWn, dc_flat=arena[3])
                for i, (w, c) in enumerate(zip(m.weights_per_layer, m.lipshitz_bound_per_layer)):
                    w.grad, c.grad, m.biases_per_layer[i].grad = dws[i], dcs[i].view_as(c), dbr[i]
                _enc_bwd(rgbn.encoding, pts, rgbn._win, dXr[:c_enc])
                self._dp_lattice_final(1)      # the colour lattice's gradient is final
                g_n = g_n + g_nc + _normalize3(n, dXr[c_sh:c_sh + 3].t().contiguous())
                g_y = torch.cat([g_sdf.view(1, -1), dXr[c_sh + 3:]], 0)                           # [33, N]
                if curv is not None:
                    rnd, shifted, feat_s, dfeat_s, ga, gb2 = curv
                    g_n = g_n + ga
                    g_shift = self._sdf_gradient_backward(gb2, feat_s, dfeat_s, e0, shifted, win, ws, bs, gb, want_pos=True)
                    g_from_shift = torch.empty_like(n)
                    L.call("psdf_curvature_shift", L.c_l(n_fg), None, L.ptr(n), L.ptr(rnd), L.c_f(1e-4), L.ptr(g_shift),
                           L.ptr(g_from_shift), L.stream())
                    g_n = g_n + g_from_shift
                # first evaluation: from (sdf, geom) directly and from n through the double backward; ONE lattice scatter
                if self.fuse_sdf_backward and double_backward_plus_supported(dims_s):
                    self._sdf_gradient_backward(g_n, feat, dfeat, e0, pts, win, ws, bs, gb, extra_gy=g_y)
                else:
                    dX1, _, _ = mlp_backward_raw(dims_s, feat, ws, bs, g_y, need_dx=True, into=(gb.dWs, gb.dbs))
                    self._sdf_gradient_backward(g_n, feat, dfeat, e0, pts, win, ws, bs, gb, extra_dfeat=dX1)
            else:
                self._dp_lattice_final(1)      # (no foreground samples on this rank: the same collective at the same point)
            # ---- off-surface points
            dXo, _, _ = mlp_backward_raw(dims_s, feat_o, ws, bs, self._g_yo, need_dx=True, into=(gb.dWs, gb.dbs))
            _enc_bwd(sdfn.encoding, off, win, dXo)
            join(self._events[3])
            if cc is not None:
                g_cw = g_cw_bg if g_cw is None else g_cw_bg + g_cw         # background first, as the single-stream order added them
                g_cb = g_cb_bg if g_cb is None else g_cb_bg + g_cb
                fixed3 = fixed.expand(-1, 3)
                gwd = torch.zeros_like(cc.weight_delta).index_add_(0, cam, torch.where(fixed3, torch.zeros_like(g_cw), g_cw))
                gbi = torch.zeros_like(cc.bias).index_add_(0, cam, torch.where(fixed3, torch.zeros_like(g_cb), g_cb))
                cc.weight_delta.grad, cc.bias.grad = gwd, gbi
        if it >= hp.iter_start_reduce_curv:      # the Lipschitz bound: four scalars, autograd is fine
            lb = rgbn.mlp.lipshitz_bound_full().mean() * hp.lipshitz_weight
            gs = torch.autograd.grad(lb, list(rgbn.mlp.lipshitz_bound_per_layer))
            for c, g in zip(rgbn.mlp.lipshitz_bound_per_layer, gs):
                c.grad = g if c.grad is None else c.grad + g
            loss = loss + lb.detach()
        return loss.view(()), n_fg, R, True
