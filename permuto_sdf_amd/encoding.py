"""Host-side mirror of the `permutohedral_encoding` operator API (boundary #2 of the hot path).

Mirrors what the reference imports at permuto_sdf_py/models/models.py:20 and uses at
models.py:149,154,172,183,186,333,370,408-420,442 and train_sdf_from_mesh.py:155:
``PermutoEncoding(pos_dim, capacity, nr_levels, nr_feat_per_level, scale_list,
appply_random_shift_per_level=..., concat_points=..., concat_points_scaling=...)``, ``output_dims()``,
``__call__(points, window)``, a parameter named ``lattice_values``, and ``Coarse2Fine(nr_levels)``.
Compute goes through the C-ABI HIP library only (csrc/encode.hip); autograd structure (a forward
Function whose backward is itself a Function, so that ``create_graph=True`` works — models.py:245-251)
follows SURVEY.md App. A.4.
"""
import math

import numpy as np
import torch

from . import _lib as L
from . import conventions as CV


def scale_factor_tensor(scale_list, pos_dim):
    scale_list = np.asarray(scale_list, dtype=np.float64)
    sf = np.empty((len(scale_list), pos_dim), dtype=np.float64)
    for i in range(pos_dim):
        sf[:, i] = 1.0 / (CV.scale_term(i, pos_dim) * scale_list)        # csrc/encode_conventions.h
    return torch.from_numpy(sf.astype(np.float32))


class _Cfg:
    """Fixed (non-tensor) parameters of one encoding instance."""

    def __init__(self, pos_dim, capacity, nr_levels, nr_feat, concat_points, points_scaling, concat_layout=None):
        self.pos_dim, self.capacity, self.nr_levels, self.nr_feat = pos_dim, capacity, nr_levels, nr_feat
        self.concat_points, self.points_scaling = bool(concat_points), float(points_scaling)
        # PSDF_ENC_CONCAT_*: padded pseudo-levels (52 channels for L=24, P=3, F=2) or exactly P appended channels (51)
        self.concat_mode = CV.concat_mode(concat_points, concat_layout)
        self.extra = int(math.ceil(pos_dim / nr_feat)) if concat_points else 0
        self.channels = CV.channels(pos_dim, nr_levels, nr_feat, self.concat_mode)


def _head(cfg, N):
    return (L.c_i(cfg.pos_dim), L.c_i(cfg.nr_feat), L.c_l(N), L.c_i(cfg.nr_levels), L.c_i(cfg.capacity))


def _tail(cfg):
    return (L.c_i(int(cfg.concat_mode)), L.c_f(cfg.points_scaling))


def encode_forward_raw(cfg, positions, lattice, scale_factor, shifts, window, skip=None, out=None, touched=None,
                       block_rows_log2=7):
    """-> sliced [channels, N] (feature-major).  `skip` [N] bool/uint8: masked points are not evaluated and their columns
    of `out` (pass a persistent buffer) stay as they are.  `touched` [L, ceil(T / 2^block_rows_log2)] uint8: the blocks of
    table rows this batch reads are set to 1 (training forward, see TouchedRows)."""
    L.require_cuda(positions, lattice)
    N = positions.shape[0]
    sliced = out if out is not None else torch.empty((cfg.channels, N), dtype=torch.float32, device=positions.device)
    if touched is not None and skip is None:
        L.call("psdf_encode_forward_mark", *_head(cfg, N), L.ptr(positions), L.ptr(lattice), L.ptr(scale_factor),
               L.ptr(shifts), L.ptr(window), *_tail(cfg), L.ptr(sliced), L.ptr(touched), L.c_i(block_rows_log2), L.stream())
    elif skip is None:
        L.call("psdf_encode_forward", *_head(cfg, N), L.ptr(positions), L.ptr(lattice), L.ptr(scale_factor),
               L.ptr(shifts), L.ptr(window), *_tail(cfg), L.ptr(sliced), L.stream())
    else:
        L.call("psdf_encode_forward_masked", *_head(cfg, N), L.ptr(positions), L.ptr(lattice), L.ptr(scale_factor),
               L.ptr(shifts), L.ptr(window), *_tail(cfg), L.ptr(skip), L.ptr(sliced), L.stream())
    return sliced


def encode_backward_raw(cfg, positions, lattice, scale_factor, shifts, window, g_fm, g_lat, g_pos):
    """Accumulates into g_lat [L,T,F] / g_pos [N,P] (either may be None).  Large batches go through the binned
    queue + LDS-reduction path, whose scratch buffer comes from torch's caching allocator."""
    import ctypes
    N = positions.shape[0]
    ws, nbytes = None, 0
    if g_lat is not None:
        fn = L.lib().psdf_encode_backward_workspace_bytes
        fn.restype = ctypes.c_int64
        nbytes = int(fn(L.c_i(cfg.pos_dim), L.c_i(cfg.nr_feat), L.c_l(N), L.c_i(cfg.nr_levels), L.c_i(cfg.capacity)))
        if nbytes > 0:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=positions.device)
    L.call("psdf_encode_backward_ws", *_head(cfg, N), L.ptr(positions), L.ptr(lattice), L.ptr(scale_factor),
           L.ptr(shifts), L.ptr(window), *_tail(cfg), L.ptr(g_fm), L.ptr(g_lat), L.ptr(g_pos), L.ptr(ws), L.c_l(nbytes),
           L.stream())


def _backward_workspace(cfg, N, device):
    import ctypes
    fn = L.lib().psdf_encode_backward_workspace_bytes
    fn.restype = ctypes.c_int64
    nbytes = int(fn(L.c_i(cfg.pos_dim), L.c_i(cfg.nr_feat), L.c_l(N), L.c_i(cfg.nr_levels), L.c_i(cfg.capacity)))
    return (torch.empty(nbytes, dtype=torch.uint8, device=device) if nbytes > 0 else None), nbytes


def encode_double_backward_raw(cfg, positions, lattice, scale_factor, shifts, window, dd_positions, g_fm, g_lat, gg_fm,
                               direct_fm=None):
    """backward of the position gradient (models.py:245-251, create_graph=True): accumulates into g_lat [L,T,F] (or None) the
    gradient w.r.t. the lattice and overwrites gg_fm [C,N] (or None: not wanted) with the gradient w.r.t. the feature gradient
    g_fm.  direct_fm [C,N] (optional): the lattice scatter of the PLAIN backward for this upstream gradient of the features is
    added in the same pass.  Batches large enough for the binned plan get its scratch from torch's caching allocator (as
    encode_backward_raw does)."""
    N = positions.shape[0]
    ws, nbytes = _backward_workspace(cfg, N, positions.device) if g_lat is not None else (None, 0)
    L.call("psdf_encode_double_backward_ws", *_head(cfg, N), L.ptr(positions), L.ptr(lattice), L.ptr(scale_factor), L.ptr(shifts),
           L.ptr(window), *_tail(cfg), L.ptr(dd_positions), L.ptr(g_fm), L.ptr(g_lat), L.ptr(gg_fm), L.ptr(direct_fm), L.ptr(ws),
           L.c_l(nbytes), L.stream())


def morton_order(positions, lo=None, hi=None):
    """Permutation that sorts unordered points [N, P <= 3] along a 10-bit-per-axis Morton curve of their bounding box: for point
    clouds that are NOT ray ordered (cfg 2's "points in a ball") neighbouring lanes then share simplices -- and table rows -- at
    the coarse and medium levels, as consecutive samples of a ray do.  The operator itself never reorders its input (outputs are
    per point, in the caller's order); a caller that can evaluate in any order does `perm = morton_order(p); f = enc(p[perm])`."""
    p = positions.detach().float()
    lo = p.min(0).values if lo is None else lo
    hi = p.max(0).values if hi is None else hi
    q = ((p - lo) / (hi - lo).clamp_min(1e-20) * 1023.0).clamp(0, 1023).to(torch.int64)
    code = torch.zeros(p.shape[0], dtype=torch.int64, device=p.device)
    for b in range(10):
        for a in range(p.shape[1]):
            code |= ((q[:, a] >> b) & 1) << (p.shape[1] * b + a)
    return torch.argsort(code)


def _feature_major(g):
    """[N, C] gradient (any strides) -> contiguous [C, N]."""
    gt = g.t()
    return gt if gt.is_contiguous() else gt.contiguous()


class TouchedRows:
    """Opt-in state of one encoding for the touched-rows optimiser: a PERSISTENT dense gradient buffer
    that every backward / double backward of the step accumulates into (no `zeros_like` + add per call), and the byte map
    of row blocks the step's forwards read.  `FusedAdamW.step` then updates only blocks that are touched or carry non-zero
    moments and clears their gradient in the same pass (csrc/optim.hip: adamw_blocks_kernel)."""

    def __init__(self, lattice, block_rows_log2=7):
        L_, T, F = lattice.shape
        self.block_rows_log2 = block_rows_log2
        self.blocks_per_level = (T + (1 << block_rows_log2) - 1) >> block_rows_log2
        if T % (1 << block_rows_log2) != 0 or ((1 << block_rows_log2) * F) % 4 != 0:
            raise ValueError("capacity must be a multiple of the row block (2^%d rows)" % block_rows_log2)
        self.block_elems = (1 << block_rows_log2) * F
        self.grad = torch.zeros_like(lattice)
        self.touched = torch.zeros((L_, self.blocks_per_level), dtype=torch.uint8, device=lattice.device)
        self.active = torch.zeros_like(self.touched)
        # Accumulation into `grad` is a SIDE EFFECT of a backward pass, so it happens only while the owner says so:
        # `with tr.accumulate(): loss.backward()`.  Any other backward through the encoding (torch.autograd.grad of an
        # auxiliary quantity, a create_graph pass, a second .backward() outside the step) returns its dense lattice gradient to
        # autograd as usual and leaves the buffer alone -- intent is never inferred from the grad mode.
        self.accumulating = False

    def accumulate(self):
        """context manager: plain backward passes inside it add their lattice gradient to `self.grad` (autograd gets None)"""
        return _Flag(self, "accumulating")


class _Flag:
    """`with _Flag(obj, name):` sets obj.name = True inside the block (re-entrant: restores the previous value)"""

    def __init__(self, obj, name):
        self.obj, self.name = obj, name

    def __enter__(self):
        self.prev = getattr(self.obj, self.name)
        setattr(self.obj, self.name, True)
        return self.obj

    def __exit__(self, *exc):
        setattr(self.obj, self.name, self.prev)


def _buffer_open(cfg):
    """the persistent gradient buffer of this encoding, if its owner has opened it for THIS backward pass (and the pass is a
    plain one: a differentiable backward being recorded must stay free of side effects)"""
    tr = getattr(cfg, "touched_rows", None)
    return tr if (tr is not None and tr.accumulating and not torch.is_grad_enabled()) else None


class PermutoEncodingFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, scale_factor, shifts, lattice, positions, window, mark=False):
        positions = positions.contiguous()
        window = window.contiguous()
        tr = getattr(cfg, "touched_rows", None)
        mark = mark and tr is not None       # decided by the caller: grad mode is always off inside Function.forward
        sliced = encode_forward_raw(cfg, positions, lattice, scale_factor, shifts, window,
                                    touched=tr.touched if mark else None,
                                    block_rows_log2=tr.block_rows_log2 if mark else 7)
        ctx.cfg = cfg
        ctx.save_for_backward(scale_factor, shifts, lattice, positions, window)
        # [N, C] view of the feature-major buffer: zero copy; BLAS consumes the transposed operand natively
        return sliced.t()

    @staticmethod
    def backward(ctx, grad_out):
        scale_factor, shifts, lattice, positions, window = ctx.saved_tensors
        need_lat, need_pos = ctx.needs_input_grad[3], ctx.needs_input_grad[4]
        # `needs_input_grad` says what requires grad, not what THIS backward pass was asked for: torch.autograd.grad(sdf,
        # points, create_graph=True) (models.py:245-251) comes through here with need_lat = True and throws the lattice
        # gradient away.  (a) A caller that knows it (the trainer) says so with `positions_gradient_only()` and the lattice
        # scatter is not even launched; (b) the persistent-buffer accumulation of TouchedRows is a side effect, so it is only
        # done inside the owner's `with tr.accumulate():` block and only in a plain backward (grad mode off here), never
        # while a differentiable backward is being recorded.
        if getattr(ctx.cfg, "skip_lattice_grad", False):
            need_lat = False
        buffer_ok = _buffer_open(ctx.cfg) is not None
        g_lat, g_pos = PermutoEncodingBackFunc.apply(ctx.cfg, scale_factor, shifts, lattice, positions, window,
                                                     grad_out, need_lat, need_pos, buffer_ok)
        # buffered mode (TouchedRows): the lattice gradient went into the persistent buffer, autograd gets None
        return None, None, None, (g_lat if (need_lat and g_lat.dim() > 0) else None), (g_pos if need_pos else None), None, None


class PermutoEncodingBackFunc(torch.autograd.Function):
    """Backward as a Function so it can itself be differentiated (double backward from positions)."""

    @staticmethod
    def forward(ctx, cfg, scale_factor, shifts, lattice, positions, window, grad_out, need_lat, need_pos, buffer_ok=False):
        g = _feature_major(grad_out)
        N = positions.shape[0]
        tr = getattr(cfg, "touched_rows", None) if buffer_ok else None
        if need_lat and tr is not None:
            # accumulate straight into the persistent buffer; autograd sees no lattice gradient (None) for this call
            g_pos = torch.zeros_like(positions) if need_pos else None
            encode_backward_raw(cfg, positions, lattice, scale_factor, shifts, window, g, tr.grad, g_pos)
            g_lat = None
        else:
            g_lat = torch.zeros_like(lattice) if need_lat else None
            g_pos = torch.zeros_like(positions) if need_pos else None
            encode_backward_raw(cfg, positions, lattice, scale_factor, shifts, window, g, g_lat, g_pos)
        ctx.cfg = cfg
        ctx.save_for_backward(scale_factor, shifts, lattice, positions, window, g)
        if g_lat is None:
            g_lat = L.zero_scalar(positions.device)
            ctx.mark_non_differentiable(g_lat)
        if g_pos is None:
            g_pos = L.zero_scalar(positions.device)
            ctx.mark_non_differentiable(g_pos)
        return g_lat, g_pos

    @staticmethod
    def backward(ctx, dd_lattice, dd_positions):
        # d(grad_lattice)/d(.) is not propagated (the reference never differentiates through the
        # lattice gradient); only the path from grad_positions is (eikonal / curvature losses).
        scale_factor, shifts, lattice, positions, window, g = ctx.saved_tensors
        cfg = ctx.cfg
        need_lat, need_g = ctx.needs_input_grad[3], ctx.needs_input_grad[6]
        if dd_positions is None or not (need_lat or need_g):
            return (None,) * 10
        N = positions.shape[0]
        dd = dd_positions.contiguous()
        tr = _buffer_open(cfg)                                      # side effects only where the owner opened the buffer
        buffered = need_lat and tr is not None
        g_lat = tr.grad if buffered else (torch.zeros_like(lattice) if need_lat else None)
        gg = torch.empty_like(g)
        encode_double_backward_raw(cfg, positions, lattice, scale_factor, shifts, window, dd, g, g_lat, gg)
        return None, None, None, (None if buffered else g_lat), None, None, (gg.t() if need_g else None), None, None, None


class PermutoEncoding(torch.nn.Module):
    def __init__(self, pos_dim, capacity, nr_levels, nr_feat_per_level, scale_per_level,
                 appply_random_shift_per_level=True, concat_points=False, concat_points_scaling=1.0,
                 init_scale=None, apply_random_shift_per_level=None, concat_layout=None):
        super().__init__()
        if apply_random_shift_per_level is not None:  # accept the correctly spelled keyword as well
            appply_random_shift_per_level = apply_random_shift_per_level
        scale_per_level = list(np.asarray(scale_per_level, dtype=np.float64).reshape(-1))
        if len(scale_per_level) != nr_levels:
            raise ValueError("scale_per_level must have nr_levels=%d entries, got %d" % (nr_levels, len(scale_per_level)))
        if (pos_dim, nr_feat_per_level) not in ((2, 2), (3, 2), (4, 2), (3, 4)):
            raise ValueError("unsupported (pos_dim, nr_feat_per_level)=(%d,%d); built variants: (2,2),(3,2),(4,2),(3,4)"
                             % (pos_dim, nr_feat_per_level))
        self.pos_dim, self.capacity, self.nr_levels, self.nr_feat_per_level = pos_dim, int(capacity), nr_levels, nr_feat_per_level
        self.scale_per_level = scale_per_level
        self.concat_points, self.concat_points_scaling = concat_points, concat_points_scaling
        # `concat_layout`: None = the default of csrc/encode_conventions.h; "pseudo_levels" | "append" to pick explicitly
        self.cfg = _Cfg(pos_dim, int(capacity), nr_levels, nr_feat_per_level, concat_points, concat_points_scaling,
                        concat_layout)
        if init_scale is None:
            init_scale = CV.C["PSDF_ENC_LATTICE_INIT_SCALE"]
        lattice_values = torch.randn(int(capacity), nr_levels, nr_feat_per_level) * init_scale
        self.lattice_values = torch.nn.Parameter(lattice_values.permute(1, 0, 2).contiguous())
        if appply_random_shift_per_level:
            shift = torch.randn(nr_levels, pos_dim) * CV.C["PSDF_ENC_RANDOM_SHIFT_SCALE"]
        else:
            shift = torch.zeros(nr_levels, pos_dim)
        # fixed (saved with the checkpoint, never trained)
        self.random_shift_per_level = torch.nn.Parameter(shift, requires_grad=False)
        self.register_buffer("scale_factor", scale_factor_tensor(scale_per_level, pos_dim), persistent=False)
        self.register_buffer("anneal_window_ones", torch.ones(nr_levels), persistent=False)
        # PSDF_FUSE_REFERENCE_MLPS=1: when this encoding is being built inside the constructor of a reference model, that
        # model's Linear/GELU stacks get the fused evaluators at our first forward call (reference_fusion.py)
        self._fuse_owner = None
        from . import reference_fusion as RF
        if RF.enabled():
            self._fuse_owner = RF.owner_under_construction()

    def output_dims(self):
        return self.cfg.channels

    def positions_gradient_only(self):
        """context manager: backward passes through this encoding inside it compute the gradient w.r.t. the POSITIONS only
        (the lattice scatter-add is not launched).  For `torch.autograd.grad(sdf, points, create_graph=True)`: autograd would
        compute the lattice gradient there and drop it."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            old = getattr(self.cfg, "skip_lattice_grad", False)
            self.cfg.skip_lattice_grad = True
            try:
                yield
            finally:
                self.cfg.skip_lattice_grad = old
        return cm()

    def enable_touched_rows(self, block_rows_log2=7):
        """Opt in to the touched-rows optimiser path (trainer only; the reference's Python keeps the plain autograd
        semantics): inside `with self.touched_rows.accumulate():` lattice gradients accumulate in `self.touched_rows.grad`
        and `lattice_values.grad` stays None; `FusedAdamW.step` must be given this object (`optim.FusedAdamW.attach`).
        Backward passes outside that block behave like plain autograd."""
        self.touched_rows = TouchedRows(self.lattice_values, block_rows_log2)
        self.cfg.touched_rows = self.touched_rows
        return self.touched_rows

    def forward(self, positions, anneal_window=None):
        if positions.dim() != 2 or positions.shape[1] != self.pos_dim:
            raise ValueError("positions must be [N, %d], got %s" % (self.pos_dim, tuple(positions.shape)))
        L.require_cuda(positions, self.lattice_values)
        if self._fuse_owner is not None:
            owner, self._fuse_owner = self._fuse_owner(), None
            if owner is not None:
                from . import reference_fusion as RF
                RF.fuse_model(owner, verbose=True)
        if anneal_window is None:
            anneal_window = self.anneal_window_ones
        else:
            anneal_window = anneal_window.to(device=positions.device, dtype=torch.float32).reshape(-1)
            if anneal_window.numel() != self.nr_levels:
                raise ValueError("anneal_window must have nr_levels entries")
        positions = positions.to(torch.float32)
        mark = torch.is_grad_enabled() and self.lattice_values.requires_grad      # a forward whose backward will run
        return PermutoEncodingFunc.apply(self.cfg, self.scale_factor, self.random_shift_per_level.detach(),
                                         self.lattice_values, positions, anneal_window, mark)

    def forward_feature_major(self, positions, anneal_window=None):
        """No-grad fast path used by the fused evaluators: returns the [channels, N] buffer itself."""
        if anneal_window is None:
            anneal_window = self.anneal_window_ones
        with torch.no_grad():
            return encode_forward_raw(self.cfg, positions.contiguous(), self.lattice_values.detach(), self.scale_factor,
                                      self.random_shift_per_level.detach(), anneal_window.contiguous())


class Coarse2Fine(torch.nn.Module):
    """Cosine-eased per-level window (same formula as reference common_utils.py:51-62)."""

    def __init__(self, nr_levels):
        super().__init__()
        self.nr_levels = nr_levels
        self.last_t = 0.0
        self.register_buffer("level_idx", torch.arange(nr_levels, dtype=torch.float32), persistent=False)

    def _update(self, t):
        """the window of `t` into the cache (no copy handed out)"""
        self.last_t = float(t)
        # a training step asks for the window of the same t half a dozen times: computed once per t, and on the HOST (five
        # elementwise launches on 24 floats cost more on the GPU than the whole evaluation does here; one 96-byte upload)
        key = (self.last_t, self.level_idx.device)
        if getattr(self, "_cached_key", None) != key:
            alpha = torch.tensor(float(t) * self.nr_levels, dtype=torch.float32, device="cpu")   # (explicit: the reference sets
            x = torch.clamp(alpha - torch.arange(self.nr_levels, dtype=torch.float32, device="cpu"), 0.0, 1.0)   # a CUDA default type)
            self._cached = (0.5 * (1.0 + torch.cos(math.pi * x + math.pi))).to(self.level_idx.device)
            self._cached_key = key
        return self._cached

    def forward(self, t):
        # a copy (one launch): callers own what they get -- unmodified reference Python may write into it
        return self._update(t).clone()

    def window_readonly(self, t):
        """the cached window itself, no copy: for callers that promise not to modify it (train_step.SdfNet)"""
        return self._update(t)

    def get_last_t(self):
        return self.last_t
