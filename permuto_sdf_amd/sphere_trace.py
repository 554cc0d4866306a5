"""Fixed-shape sphere tracing of an encoded SDF (BASELINE config 5: 1920x1080 rays, 15 iterations).

Mirrors ``sphere_trace`` of the reference (permuto_sdf_py/utils/sdf_utils.py:120-218) on top of the same kernels, but
keeps one slot per ray instead of compacting the unconverged rays with boolean masks every iteration: the whole trace
is a fixed sequence of launches -- first hit -> [masked encode -> masked MLP -> step (+ compacted long marches)] x n ->
masked final SDF + analytic normal -- has no host synchronisation and can therefore be captured ONCE into a hipGraph and
replayed per frame (``SphereTracer.capture``).  Converged rays are masked out of the SDF evaluations (no gathers; fully
converged 32-ray tiles no MLP), rays that met no occupied voxel also out of the final evaluation.  End points are bit
identical to the reference-style compacting loop (tests/test_gpu_sphere_trace.py).
"""
import ctypes

import torch

from . import _lib as L
from .encoding import _head, _tail, encode_forward_raw
from .mlp import _dims_array, mlp_forward_raw, pack_params


class SphereTracer:
    def __init__(self, encoding, mlp, occupancy_grid, sphere, window=None):
        self.enc, self.mlp, self.grid, self.sphere = encoding, mlp, occupancy_grid, sphere
        dev = encoding.lattice_values.device
        self.window = window if window is not None else torch.ones(encoding.nr_levels, device=dev)
        self._graph = None
        # step + march as two launches, the (few) long marches densely packed in the second one (csrc/sampling.hip,
        # sphere_trace_step_a_kernel); False: the single kernel, where a marching lane holds its whole wave
        self.compact_marches = True
        self.coarse_mask_for_marches = True

    # ---- building blocks -----------------------------------------------------------------------------------
    def _sdf(self, pts, dims, packed, skip=None, out=None, feat_buf=None):
        """SDF channel of the net at `pts` -> (feat [C,N], sdf [1,N]).  Two launches: the level-major encode kernel keeps
        one 2-MiB table at a time in every XCD's L2 and runs at full occupancy, which beats the single fused launch
        (csrc/fused.hip, whose 24 tables thrash the L2) even more clearly at 24 levels; both honour the per-ray mask
        (converged rays: no gathers, fully converged 32-ray tiles: no MLP)."""
        e = self.enc
        feat = encode_forward_raw(e.cfg, pts, e.lattice_values.detach(), e.scale_factor,
                                  e.random_shift_per_level.detach(), self.window, skip=skip, out=feat_buf)
        return feat, mlp_forward_raw(dims, feat, packed, skip=skip, out=out)

    def _grid_args(self):
        g = self.grid
        return L.c_i(g.m_nr_voxels_per_dim), L.c_f(g.m_grid_extent), (L.c_f * 3)(*g.m_grid_translation)

    @torch.no_grad()
    def trace(self, ray_origins, ray_dirs, nr_sphere_traces=15, sdf_multiplier=0.9, sdf_converged_tresh=2e-4,
              return_gradients=True):
        """-> pts [R,3], sdf [R,1], sdf_gradients [R,3] or None, converged [R,1] bool.  Rays that never met an occupied voxel
        are reported converged with their point left at the ray origin and sdf = 0, gradient = 0: the reference's trace
        does not hold them at all (it compacts to the rays that shoot through occupancy, sdf_utils.py:127-129), and the
        final SDF + normal evaluation skips them here too (`self.no_hit`, [R] bool, tells them apart afterwards)."""
        o, d = ray_origins.contiguous(), ray_dirs.contiguous()
        R, dev = o.shape[0], o.device
        _, te, _, tx, _ = self.sphere.ray_intersection(o, d)
        pts = torch.empty((R, 3), dtype=torch.float32, device=dev)
        conv = torch.empty((R, 1), dtype=torch.bool, device=dev)
        occ = self.grid._occ()
        # no coarse occupancy mask here (bridge.OccupancyGrid._coarse): traced rays sit in or next to occupied voxels, the
        # marches are a few steps long, and the frame measured 1.4 % slower with it (117.8 against 119.5 FPS)
        coarse = None
        L.call("psdf_first_hit_dense", L.c_i(R), *self._grid_args(), L.ptr(occ), L.ptr(o), L.ptr(d), L.ptr(te), L.ptr(tx),
               L.ptr(pts), L.ptr(conv), L.ptr(coarse), L.stream())
        no_hit = conv.view(-1).clone()      # before the iterations `converged` means exactly "met no occupied voxel"
        self.no_hit = no_hit
        # channel 0 of the last layer is the SDF (models.py:190-192); the geometry features are not needed to trace,
        # so the net is evaluated with a 1-row head (same arithmetic for that row, 1/33 of the output traffic)
        ws = [l.weight.detach() for l in self.mlp.layers]
        bs = [l.bias.detach() for l in self.mlp.layers]
        ws[-1], bs[-1] = ws[-1][0:1].contiguous(), bs[-1][0:1].contiguous()
        dims = list(self.mlp.dims[:-1]) + [1]
        packed = pack_params(dims, ws, bs)
        sdf = torch.zeros((1, R), dtype=torch.float32, device=dev)
        feat_buf = torch.zeros((self.enc.cfg.channels, R), dtype=torch.float32, device=dev)
        if self.compact_marches:   # work buffers of the two-launch step: flags, list, one counter per iteration (one fill)
            flags = torch.empty(R, dtype=torch.uint8, device=dev)
            todo = torch.empty(R, dtype=torch.int32, device=dev)
            counts = torch.zeros(max(1, nr_sphere_traces), dtype=torch.int32, device=dev)
            # the compacted marches are the long ones through empty space: there the coarse mask pays (one mask per trace,
            # the grid does not change during it)
            coarse_b = self.grid._coarse(1 << 30) if self.coarse_mask_for_marches else None
        for it in range(nr_sphere_traces):
            self._sdf(pts, dims, packed, skip=conv.view(-1), out=sdf, feat_buf=feat_buf)   # converged rays are skipped
            if self.compact_marches:
                L.call("psdf_sphere_trace_step_compacted", L.c_i(R), *self._grid_args(), L.ptr(occ), L.ptr(d), L.ptr(sdf),
                       L.c_f(sdf_multiplier), L.c_f(sdf_converged_tresh), L.ptr(pts), L.ptr(conv), L.ptr(flags), L.ptr(todo),
                       L.ptr(counts[it:it + 1]), L.ptr(coarse_b), L.stream())
            else:
                L.call("psdf_sphere_trace_step", L.c_i(R), *self._grid_args(), L.ptr(occ), L.ptr(d), L.ptr(sdf),
                       L.c_f(sdf_multiplier), L.c_f(sdf_converged_tresh), L.ptr(pts), L.ptr(conv), L.ptr(coarse), L.stream())
        # final SDF (+ analytic normal) at the end points of the rays that took part
        sdf.zero_()
        feat, sdf = self._sdf(pts, dims, packed, skip=no_hit, out=sdf, feat_buf=feat_buf)
        grads = None
        if return_gradients:
            # analytic normal: d sdf / d x = encode_backward_positions( mlp_backward_dX( 1 ) )
            n_layers = len(dims) - 1
            d_feat = torch.empty_like(feat)
            gy = torch.ones_like(sdf)
            Wp = (ctypes.c_void_p * n_layers)(*[w.data_ptr() for w in ws])
            Bp = (ctypes.c_void_p * n_layers)(*[b.data_ptr() for b in bs])
            L.call("psdf_mlp_backward_data_masked", L.c_i(n_layers), _dims_array(dims), L.c_l(R), L.ptr(feat), Wp, Bp,
                   L.ptr(gy), L.ptr(no_hit), L.ptr(d_feat), L.stream())
            grads = torch.zeros((R, 3), dtype=torch.float32, device=dev)
            cfg = self.enc.cfg
            L.call("psdf_encode_backward_positions_masked", *_head(cfg, R), L.ptr(pts), L.ptr(self.enc.lattice_values.detach()),
                   L.ptr(self.enc.scale_factor), L.ptr(self.enc.random_shift_per_level.detach()), L.ptr(self.window),
                   *_tail(cfg), L.ptr(d_feat), L.ptr(no_hit), L.ptr(grads), L.stream())
        return pts, sdf.reshape(-1, 1), grads, conv

    # ---- hipGraph ------------------------------------------------------------------------------------------------
    def capture(self, ray_origins, ray_dirs, **kw):
        """Capture one full trace for rays of this shape; `replay()` re-runs it on the CURRENT contents of the two
        input tensors (update them in place) and returns the same output tensors."""
        self._o, self._d = ray_origins.contiguous(), ray_dirs.contiguous()
        self.trace(self._o, self._d, **kw)                      # warm-up outside capture (allocator, lazy init)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._out = self.trace(self._o, self._d, **kw)
        return self._out

    def replay(self):
        self._graph.replay()
        return self._out
