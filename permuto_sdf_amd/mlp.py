"""Fused small-MLP evaluator (csrc/mlp.hip) behind a torch.nn.Module.

Replaces the reference's ``torch.nn.Sequential(Linear, GELU, ..., Linear)`` evaluators
(permuto_sdf_py/models/models.py:153-161, :451-470) with one MFMA kernel per pass.  Parameters keep
torch.nn.Linear layout/names (``layers.{i}.weight|bias``) so a state_dict maps 1:1 onto the reference's
``mlp_sdf.{2i}.weight|bias``.  Activations are exchanged feature-major ([C, N]) with the encoding kernels.
"""
import ctypes
import math

import torch

from . import _lib as L


def _dims_array(dims):
    return (ctypes.c_int * len(dims))(*dims)


def packed_size(dims):
    n = L.lib().psdf_mlp_packed_size
    n.restype = ctypes.c_int64
    r = n(L.c_i(len(dims) - 1), _dims_array(dims))
    if r < 0:
        raise L.PsdfError("psdf_mlp_packed_size: bad layer widths %s" % (dims,))
    return int(r)


def f16_forward_supported(dims):
    """psdf_mlp_forward_f16 exists for the BASELINE net: <= 64 inputs, 64x3, <= 4 outputs"""
    return len(dims) == 5 and dims[0] <= 64 and dims[1] == dims[2] == dims[3] == 64 and dims[4] <= 4


def pack_params(dims, weights, biases, f16=False):
    """torch-layout weights/biases -> MFMA-operand ordered buffer (one small kernel).  f16: the split image holds two fp16
    pieces per weight (for mlp_forward_raw(..., f16=True) only)."""
    n_layers = len(dims) - 1
    L.require_cuda(*weights)
    packed = torch.empty(packed_size(dims), dtype=torch.float32, device=weights[0].device)
    ws = [w.detach().contiguous() for w in weights]
    bs = [b.detach().contiguous() for b in biases]
    W = (ctypes.c_void_p * n_layers)(*[w.data_ptr() for w in ws])
    B = (ctypes.c_void_p * n_layers)(*[b.data_ptr() for b in bs])
    L.call("psdf_mlp_pack_f16" if f16 else "psdf_mlp_pack", L.c_i(n_layers), _dims_array(dims), W, B, L.ptr(packed), L.stream())
    return packed


def mlp_forward_raw(dims, x_fm, packed, skip=None, out=None, f16=False):
    """x_fm [dims[0], N] feature-major -> y [dims[-1], N] feature-major.  `skip` [N]: 32-sample tiles that are masked
    entirely are not evaluated (their entries of `out` stay as they are).  f16: the two-piece fp16 arithmetic (`packed` from
    pack_params(..., f16=True); BASELINE net only; values below 65504)."""
    N = x_fm.shape[1]
    y = out if out is not None else torch.empty((dims[-1], N), dtype=torch.float32, device=x_fm.device)
    if f16:
        assert skip is None
        L.call("psdf_mlp_forward_f16", L.c_i(len(dims) - 1), _dims_array(dims), L.c_l(N), L.ptr(x_fm), L.ptr(packed), L.ptr(y),
               L.stream())
    elif skip is None:
        L.call("psdf_mlp_forward", L.c_i(len(dims) - 1), _dims_array(dims), L.c_l(N), L.ptr(x_fm), L.ptr(packed), L.ptr(y),
               L.stream())
    else:
        L.call("psdf_mlp_forward_masked", L.c_i(len(dims) - 1), _dims_array(dims), L.c_l(N), L.ptr(x_fm), L.ptr(packed),
               L.ptr(skip), L.ptr(y), L.stream())
    return y


_wide_f16_fn = None


def wide_f16_forward_candidate(dims):
    """the shapes psdf_mlp_forward_wide_f16 is instantiated for (csrc/mlp_wide.hip): the colour network 111-128-128-64-3 and the
    background density net 52-64-64-64-65; asked before the call so that other nets do not pay an ABI round trip for a -2"""
    if len(dims) != 5:
        return False
    d = dims
    colour = d[0] <= 112 and d[1] <= 128 and d[2] <= 128 and d[3] <= 64 and d[4] <= 16 and not (d[1] <= 64 and d[2] <= 64)
    density = d[0] <= 64 and 32 < d[1] <= 64 and 32 < d[2] <= 64 and 32 < d[3] <= 64 and 16 < d[4] <= 80
    return colour or density


def mlp_forward_wide_f16_raw(dims, x_fm, weights, biases, out=None):
    """The colour network's forward (LipshitzMLP 111 -> 128 -> 128 -> 64 -> 3, models.py:349-350) on the fp16 matrix pipe with
    two pieces per fp32 operand (csrc/mlp_wide.hip, round 6): x_fm [dims[0], N] feature-major, `weights` / `biases` the torch
    layout (for a LipshitzMLP the NORMALISED weights) -> y [dims[-1], N], or None when the library declines (-2: another
    shape, stream capture, PSDF_MLP_WIDE_SPLIT=f32, a value beyond the fp16 range met earlier) -- the caller then takes
    mlp_forward_raw (fp32 MFMAs)."""
    global _wide_f16_fn
    if len(dims) != 5:
        return None
    if _wide_f16_fn is None:
        _wide_f16_fn = L.lib().psdf_mlp_forward_wide_f16
        _wide_f16_fn.restype = ctypes.c_int
    N = x_fm.shape[1]
    y = out if out is not None else torch.empty((dims[-1], N), dtype=torch.float32, device=x_fm.device)
    arr = lambda ts: (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ts])
    ws = [w.detach() if w.is_contiguous() else w.detach().contiguous() for w in weights]
    bs = [b.detach() for b in biases]
    rc = _wide_f16_fn(L.c_i(4), _dims_array(dims), L.c_l(N), L.ptr(x_fm), arr(ws), arr(bs), L.ptr(y), L.stream())
    if rc == -2:
        return None
    if rc != 0:
        L.check(rc, "psdf_mlp_forward_wide_f16")
    return y


def _grad_views(dims, flat=None, dev=None):
    """dW_l [d_{l+1}, d_l] and db_l [d_{l+1}] as views of ONE buffer (every slice starts on a 16-byte boundary); allocates
    it zero-filled when `flat` is None"""
    n_layers = len(dims) - 1
    sizes = [dims[l + 1] * dims[l] for l in range(n_layers)] + [dims[l + 1] for l in range(n_layers)]
    offs, tot = [], 0
    for sz in sizes:
        offs.append(tot)
        tot += (sz + 3) & ~3
    if flat is None:
        flat = torch.zeros(tot, dtype=torch.float32, device=dev)
    dWs = [flat[offs[l]:offs[l] + sizes[l]].view(dims[l + 1], dims[l]) for l in range(n_layers)]
    dbs = [flat[offs[n_layers + l]:offs[n_layers + l] + sizes[n_layers + l]] for l in range(n_layers)]
    return flat, dWs, dbs


def _zero_grads(dims, dev):
    """one fill launch instead of 2 per layer"""
    _, dWs, dbs = _grad_views(dims, dev=dev)
    return dWs, dbs


class GradBuffer:
    """Persistent parameter-gradient buffer of a FusedMLP (FusedMLP.enable_grad_buffer): every PLAIN backward of a step --
    the backward of each evaluation and of each analytic input gradient (double backward) -- accumulates into the same
    dW / db views (the kernels accumulate anyway), instead of returning fresh tensors that autograd then sums with one add
    per parameter and contribution.  The owner zeroes it after the optimiser step (`zero()`, one fill).  A backward under
    create_graph never touches it (its parameter gradients are dropped by the caller, see input_gradient_only)."""

    def __init__(self, dims, device):
        self.flat, self.dWs, self.dbs = _grad_views(dims, dev=device)
        self.accumulating = False       # set by the owner around ITS backward: `with gb.accumulate(): loss.backward()`

    def accumulate(self):
        """context manager: plain backward passes of the net inside it add their parameter gradients to this buffer; outside
        it every backward returns its gradients to autograd (no side effects for auxiliary autograd.grad / .backward calls)"""
        from .encoding import _Flag
        return _Flag(self, "accumulating")

    def zero(self):
        self.flat.zero_()


def mlp_backward_raw(dims, x_fm, weights, biases, gy_fm, need_dx=True, need_dw=True, into=None):
    """weights/biases: the torch-layout parameters (the backward kernel builds its own LDS image from them)
    -> (dx_fm [dims[0], N] or None, [dW_l], [db_l]); need_dw=False: data gradient only (lighter kernel, empty lists);
    into = (dWs, dbs): accumulate into these instead of fresh zero-filled tensors; gy_fm = None (with need_dw=False): the unit
    gradient of output 0, d y_0 / d x"""
    assert gy_fm is not None or not need_dw
    N = x_fm.shape[1]
    n_layers = len(dims) - 1
    dev = x_fm.device
    ws = [w.detach().contiguous() for w in weights]
    bs = [b.detach().contiguous() for b in biases]
    dx = torch.empty((dims[0], N), dtype=torch.float32, device=dev) if need_dx else None
    Wp = (ctypes.c_void_p * n_layers)(*[w.data_ptr() for w in ws])
    Bp = (ctypes.c_void_p * n_layers)(*[b.data_ptr() for b in bs])
    if need_dw:
        dWs, dbs = into if into is not None else _zero_grads(dims, dev)
        W = (ctypes.c_void_p * n_layers)(*[w.data_ptr() for w in dWs])
        B = (ctypes.c_void_p * n_layers)(*[b.data_ptr() for b in dbs])
    else:
        dWs, dbs, W, B = [], [], None, None
    L.call("psdf_mlp_backward", L.c_i(n_layers), _dims_array(dims), L.c_l(N), L.ptr(x_fm), Wp, Bp, L.ptr(gy_fm),
           L.ptr(dx), W, B, L.stream())
    return dx, dWs, dbs


def backward_supported(dims):
    """True when csrc/mlp_bwd.hip has an instantiation for these widths (tile signature, 16-wide tiles)."""
    n_layers = len(dims) - 1
    if n_layers not in (3, 4):
        return False
    t = [(d + 15) // 16 for d in dims]
    if n_layers == 4 and t[0] <= 7 and t[1] <= 8 and t[2] <= 8 and t[3] <= 4 and dims[4] <= 16 and (t[1] > 4 or t[2] > 4):
        return True     # csrc/mlp_wide.hip: the 128-wide colour network (workgroup-cooperative kernel)
    sig = (t[0], t[1], t[2], t[3] if n_layers == 4 else 0, t[n_layers], dims[-1] <= 4)
    return sig in {(3, 4, 4, 4, 1, True), (4, 4, 4, 4, 1, True), (2, 4, 4, 4, 1, True), (4, 2, 2, 2, 1, True),
                   (3, 2, 2, 2, 1, True), (2, 2, 2, 2, 1, True), (4, 2, 2, 2, 3, False), (4, 4, 4, 4, 5, False),
                   (4, 4, 4, 4, 3, False), (5, 4, 4, 0, 1, True), (3, 2, 2, 2, 3, False), (3, 4, 4, 4, 3, False)}


_ANNOUNCED = set()


def torch_fallback_allowed(module=None):
    """Widths without a fused backward / double-backward instantiation are an ERROR by default: this package is the hand-written
    kernel path, and a silent dispatch to a vendor BLAS would be measured and trusted as if it were that path.  Opt in per net
    (`FusedMLP(..., allow_torch_fallback=True)` / `net.allow_torch_fallback = True`) or per process (PSDF_MLP_TORCH_FALLBACK=1)
    to have such nets differentiated by torch autograd over rocBLAS on the GPU instead (announced once per net shape)."""
    import os
    return bool(getattr(module, "allow_torch_fallback", False)) or os.environ.get("PSDF_MLP_TORCH_FALLBACK") == "1"


def _announce(kind, dims, module=None):
    """say ONCE per (operator, widths) that a net is served by torch/rocBLAS on the GPU instead of a fused kernel"""
    if not torch_fallback_allowed(module):
        raise L.PsdfError("permuto_sdf_amd: no fused %s kernel is instantiated for the MLP widths %s (csrc/mlp_bwd.hip lists the "
                          "instantiated widths).  Refusing to fall back to torch/rocBLAS silently: pass allow_torch_fallback=True "
                          "to FusedMLP or set PSDF_MLP_TORCH_FALLBACK=1 to opt in." % (kind, list(dims)))
    key = (kind, tuple(dims))
    if key not in _ANNOUNCED:
        _ANNOUNCED.add(key)
        import warnings
        warnings.warn("permuto_sdf_amd: no fused %s kernel is instantiated for the MLP widths %s; using torch autograd over "
                      "rocBLAS on the GPU for it (csrc/mlp_bwd.hip lists the instantiated widths)" % (kind, list(dims)),
                      RuntimeWarning, stacklevel=3)


def dx_only_supported(dims):
    """the data-gradient-only variant exists for the narrow kernel family (csrc/mlp_bwd.hip), not for mlp_wide.hip"""
    t = [(d + 15) // 16 for d in dims]
    return backward_supported(dims) and not (len(dims) == 5 and (t[1] > 4 or t[2] > 4))


def _torch_gpu_backward(dims, x_fm, weights, biases, gy_fm, need_dx, module=None):
    """Backward for widths the fused kernel family does not cover yet (the 128-wide colour net): the same
    Linear/GELU stack re-evaluated with torch ops ON THE GPU (rocBLAS) under autograd.  Not a CPU path."""
    _announce("backward", dims, module)
    n_layers = len(dims) - 1
    with torch.enable_grad():
        x = x_fm.t().detach().requires_grad_(need_dx)
        ws = [w.detach().requires_grad_(True) for w in weights]
        bs = [b.detach().requires_grad_(True) for b in biases]
        h = x
        for i in range(n_layers):
            h = torch.nn.functional.linear(h, ws[i], bs[i])
            if i < n_layers - 1:
                h = torch.nn.functional.gelu(h)
        outs = torch.autograd.grad(h, ([x] if need_dx else []) + ws + bs, gy_fm.t())
    k = 1 if need_dx else 0
    dx = outs[0].t().contiguous() if need_dx else None
    return dx, list(outs[k:k + n_layers]), list(outs[k + n_layers:])


def _torch_gpu_double_backward(dims, x_fm, weights, biases, gy_fm, v_fm, module=None):
    """VJP of the map (x, params) -> dx = J_x^T gy with the upstream gradient v (what differentiating through the
    analytic input gradient needs: eikonal / curvature losses, models.py:245-251), by torch autograd ON THE GPU.
    Generic fallback for widths without a fused double-backward kernel.  -> (dX [C,N], [dW_l], [db_l])"""
    _announce("double backward", dims, module)
    n_layers = len(dims) - 1
    with torch.enable_grad():
        x = x_fm.t().detach().requires_grad_(True)
        ws = [w.detach().requires_grad_(True) for w in weights]
        bs = [b.detach().requires_grad_(True) for b in biases]
        h = x
        for i in range(n_layers):
            h = torch.nn.functional.linear(h, ws[i], bs[i])
            if i < n_layers - 1:
                h = torch.nn.functional.gelu(h)
        (gx,) = torch.autograd.grad(h, x, gy_fm.t(), create_graph=True)
        outs = torch.autograd.grad(gx, [x] + ws + bs, v_fm.t(), allow_unused=True)
    outs = [o if o is not None else torch.zeros_like(t) for o, t in zip(outs, [x] + ws + bs)]
    return outs[0].t().contiguous(), list(outs[1:1 + n_layers]), list(outs[1 + n_layers:])


class _InputGradOnly:
    """`with input_gradient_only(net, ...):` -- backward passes of THESE FusedMLP modules inside the block compute d/dx only
    (the data-gradient-only kernel: no accumulators, more waves per CU).  For torch.autograd.grad(sdf, points,
    create_graph=True) (models.py:245-251), where autograd would compute the parameter gradients too and drop them.  The flag
    lives on the module (like `cfg.skip_lattice_grad` of the encoding): another net differentiated inside the block, or on
    another thread, keeps its parameter gradients."""

    def __init__(self, modules):
        self.modules = modules

    def __enter__(self):
        self.prev = [m._input_grad_only for m in self.modules]
        for m in self.modules:
            m._input_grad_only = True

    def __exit__(self, *exc):
        for m, v in zip(self.modules, self.prev):
            m._input_grad_only = v


def input_gradient_only(*modules):
    if not modules:
        raise TypeError("input_gradient_only(net, ...): name the FusedMLP modules whose parameter gradients are to be skipped")
    return _InputGradOnly(modules)


def _grad_buffer_open(module):
    gb = getattr(module, "grad_buffer", None)
    return gb if (gb is not None and gb.accumulating and not torch.is_grad_enabled()) else None


class _FusedMLPFunc(torch.autograd.Function):
    """y = MLP(x).  Its backward is itself a Function (_FusedMLPBackFunc) so that `create_graph=True` works: the
    reference differentiates through d sdf / d x (models.py:236-251)."""

    @staticmethod
    def forward(ctx, module, x, *params):
        n_layers = module.n_layers
        weights, biases = params[:n_layers], params[n_layers:]
        x_fm = x.t()
        if not x_fm.is_contiguous():
            x_fm = x_fm.contiguous()
        y = None
        if wide_f16_forward_candidate(module.dims):      # colour network / background density net: the fp16 matrix pipe (round 6)
            y = mlp_forward_wide_f16_raw(module.dims, x_fm, weights, biases)
        if y is None:
            packed = pack_params(module.dims, weights, biases)
            y = mlp_forward_raw(module.dims, x_fm, packed)
        ctx.module = module
        ctx.n_layers = n_layers
        ctx.save_for_backward(x, *weights, *biases)   # the INPUTS themselves: they keep their place in the graph
        return y.t()

    @staticmethod
    def backward(ctx, gy):
        x = ctx.saved_tensors[0]
        params = ctx.saved_tensors[1:]
        if getattr(ctx.module, "_input_grad_only", False):      # the caller only wants d/dx from this pass
            outs = _FusedMLPBackFunc.apply(ctx.module, (ctx.needs_input_grad[1], False), x, gy, *params)
            return (None, outs[0] if ctx.needs_input_grad[1] else None, *([None] * len(params)))
        gb = _grad_buffer_open(ctx.module)
        if gb is not None:   # plain backward inside the owner's accumulate(): parameter gradients go to the module's buffer
            outs = _FusedMLPBackFunc.apply(ctx.module, (ctx.needs_input_grad[1], True, True), x, gy, *params)
            return (None, outs[0] if ctx.needs_input_grad[1] else None, *([None] * len(params)))
        outs = _FusedMLPBackFunc.apply(ctx.module, ctx.needs_input_grad[1], x, gy, *params)
        dx = outs[0] if ctx.needs_input_grad[1] else None
        return (None, dx, *outs[1:])


class _FusedMLPBackFunc(torch.autograd.Function):
    """(x, gy, params) -> (dx, dW.., db..).  backward: only the path from dx is propagated (into x and the parameters);
    second derivatives through dW/db are never needed by the reference and are not provided."""

    @staticmethod
    def forward(ctx, module, need_dx, x, gy, *params):
        need_dw, buffered = True, False
        if isinstance(need_dx, tuple):
            need_dx, need_dw, buffered = (tuple(need_dx) + (False,))[:3]
        n_layers = module.n_layers
        weights, biases = params[:n_layers], params[n_layers:]
        x_fm = x.t()
        if not x_fm.is_contiguous():
            x_fm = x_fm.contiguous()
        gy_fm = gy.t()
        if not gy_fm.is_contiguous():
            gy_fm = gy_fm.contiguous()
        if not need_dw and dx_only_supported(module.dims):
            dx, dWs, dbs = mlp_backward_raw(module.dims, x_fm, weights, biases, gy_fm, need_dx=need_dx, need_dw=False)
        elif backward_supported(module.dims):
            gb = module.grad_buffer if buffered else None
            dx, dWs, dbs = mlp_backward_raw(module.dims, x_fm, weights, biases, gy_fm, need_dx=need_dx,
                                            into=(gb.dWs, gb.dbs) if gb is not None else None)
            if not need_dw or buffered:
                dWs, dbs = [], []
        else:
            dx, dWs, dbs = _torch_gpu_backward(module.dims, x_fm, weights, biases, gy_fm, need_dx, module)
            if buffered:
                gb = module.grad_buffer
                for dst, src in zip(gb.dWs + gb.dbs, list(dWs) + list(dbs)):
                    dst.add_(src)
                dWs, dbs = [], []
        ctx.module, ctx.n_layers = module, n_layers
        ctx.set_materialize_grads(False)      # unused outputs (dW, db are never differentiated) arrive as None in backward
        ctx.save_for_backward(x_fm, gy_fm, *weights, *biases)
        if dx is None:
            dx_out = L.zero_scalar(x.device)
            ctx.mark_non_differentiable(dx_out)
        else:
            dx_out = dx.t()
        return (dx_out, *dWs, *dbs)

    @staticmethod
    def backward(ctx, g_dx, *g_params):
        n_layers = ctx.n_layers
        x_fm, gy_fm = ctx.saved_tensors[0], ctx.saved_tensors[1]
        weights = ctx.saved_tensors[2:2 + n_layers]
        biases = ctx.saved_tensors[2 + n_layers:]
        # Only the path the reference needs is implemented: d<dx, v>/d(x, params) with gy a CONSTANT (models.py:240-251 passes
        # ones).  The two others would be silently wrong gradients, so they raise.
        if any(g is not None for g in g_params):
            raise NotImplementedError("FusedMLP: differentiating through the parameter gradients (dW, db) is not implemented")
        if ctx.needs_input_grad[3] and g_dx is not None:
            raise NotImplementedError("FusedMLP: the upstream gradient of the MLP output requires grad (a differentiable "
                                      "function of the output is being differentiated twice); the J_x v term is not implemented")
        if g_dx is None:
            return (None,) * (4 + 2 * n_layers)
        v_fm = g_dx.t()
        if not v_fm.is_contiguous():
            v_fm = v_fm.contiguous()
        gb = _grad_buffer_open(ctx.module)
        if gb is not None:
            dX, _, _ = mlp_double_backward(ctx.module.dims, x_fm, weights, biases, gy_fm, v_fm, into=(gb.dWs, gb.dbs),
                                           module=ctx.module)
            return (None, None, dX.t(), None, *([None] * (2 * n_layers)))
        dX, dWs, dbs = mlp_double_backward(ctx.module.dims, x_fm, weights, biases, gy_fm, v_fm, module=ctx.module)
        return (None, None, dX.t(), None, *dWs, *dbs)


def double_backward_supported(dims):
    """True when csrc/mlp_bwd.hip has a fused double-backward instantiation for these widths"""
    if len(dims) != 5:
        return False
    t = [(d + 15) // 16 for d in dims]
    sig = (t[0], t[1], t[2], t[3], t[4], dims[-1] <= 4)
    return sig in {(4, 2, 2, 2, 3, False), (3, 2, 2, 2, 3, False), (4, 2, 2, 2, 1, True), (3, 2, 2, 2, 1, True),
                   (2, 2, 2, 2, 1, True), (3, 4, 4, 4, 1, True), (4, 4, 4, 4, 1, True)}


def double_backward_plus_supported(dims):
    """psdf_mlp_double_backward_plus: the reference's SDF net shapes (<= 64 inputs, 32 x 3 hidden, matrix output layer <= 48 rows)"""
    return (len(dims) == 5 and 32 < dims[0] <= 64 and 16 < dims[1] <= 32 and 16 < dims[2] <= 32 and 16 < dims[3] <= 32
            and 32 < dims[4] <= 48)


def mlp_double_backward(dims, x_fm, weights, biases, gy_fm, v_fm, into=None, module=None, gy2_fm=None):
    """-> (dX [C,N], [dW_l], [db_l]) of <dx(x, params; gy), v>; fused kernel where one is built, torch (GPU) otherwise;
    into = (dWs, dbs): accumulate the parameter gradients there; gy_fm = None: the unit gradient of output 0.
    gy2_fm [dims[-1], N] (only where double_backward_plus_supported(dims)): the plain backward of this upstream gradient of the
    outputs rides in the same launch -- dX is then the sum of both data gradients, the parameter gradients hold both"""
    if gy2_fm is not None:
        assert double_backward_plus_supported(dims), dims
    if gy_fm is None and not double_backward_supported(dims):      # the torch route needs the tensor: unit gradient of output 0
        gy_fm = torch.zeros((dims[-1], x_fm.shape[1]), dtype=torch.float32, device=x_fm.device)
        gy_fm[0].fill_(1.0)
    if not double_backward_supported(dims):
        dX, dWs, dbs = _torch_gpu_double_backward(dims, x_fm, weights, biases, gy_fm, v_fm, module)
        if into is not None:
            for dst, src in zip(list(into[0]) + list(into[1]), list(dWs) + list(dbs)):
                if src is not None:
                    dst.add_(src)
            dWs, dbs = into
        return dX, dWs, dbs
    N = x_fm.shape[1]
    n_layers = len(dims) - 1
    dev = x_fm.device
    ws = [w.detach().contiguous() for w in weights]
    bs = [b.detach().contiguous() for b in biases]
    dx2 = torch.empty((dims[0], N), dtype=torch.float32, device=dev)
    dWs, dbs = into if into is not None else _zero_grads(dims, dev)
    Wp = (ctypes.c_void_p * n_layers)(*[w.data_ptr() for w in ws])
    Bp = (ctypes.c_void_p * n_layers)(*[b.data_ptr() for b in bs])
    W = (ctypes.c_void_p * n_layers)(*[w.data_ptr() for w in dWs])
    B = (ctypes.c_void_p * n_layers)(*[b.data_ptr() for b in dbs])
    if gy2_fm is not None:
        L.call("psdf_mlp_double_backward_plus", L.c_i(n_layers), _dims_array(dims), L.c_l(N), L.ptr(x_fm), Wp, Bp, L.ptr(gy_fm),
               L.ptr(v_fm), L.ptr(gy2_fm), L.ptr(dx2), W, B, L.stream())
    else:
        L.call("psdf_mlp_double_backward", L.c_i(n_layers), _dims_array(dims), L.c_l(N), L.ptr(x_fm), Wp, Bp, L.ptr(gy_fm),
               L.ptr(v_fm), L.ptr(dx2), W, B, L.stream())
    return dx2, dWs, dbs


class FusedMLP(torch.nn.Module):
    """Linear(d0,d1)-GELU-...-Linear(d_{n-1},d_n), GELU(erf) after every layer but the last."""

    def __init__(self, dims, reference_init=False, last_layer_linear_init=True, allow_torch_fallback=False):
        """reference_init: initialise like the reference's nets (leaky_relu_init on every layer, gain 1 on the last one when
        `last_layer_linear_init`; models.py:161-162) instead of torch.nn.Linear's default.  allow_torch_fallback: see
        `torch_fallback_allowed` (default: widths without a fused backward kernel raise)"""
        super().__init__()
        self.allow_torch_fallback = bool(allow_torch_fallback)
        self.dims = [int(d) for d in dims]
        self.n_layers = len(self.dims) - 1
        self.layers = torch.nn.ModuleList(
            [torch.nn.Linear(self.dims[i], self.dims[i + 1]) for i in range(self.n_layers)])
        if reference_init:
            for i, l in enumerate(self.layers):
                leaky_relu_init_(l, 1.0 if (i == self.n_layers - 1 and last_layer_linear_init) else 0.0)
        self.grad_buffer = None
        self._input_grad_only = False

    def input_gradient_only(self):
        """context manager, see mlp.input_gradient_only"""
        return _InputGradOnly((self,))

    def enable_grad_buffer(self):
        """-> GradBuffer: plain backward passes inside `with gb.accumulate():` ADD this net's parameter gradients into it (and
        return None to autograd); the caller hands its views to the optimiser (`assign_grads`) and zeroes it after the step"""
        self.grad_buffer = GradBuffer(self.dims, self.layers[0].weight.device)
        return self.grad_buffer

    def assign_grads(self):
        """point every parameter's .grad at its view of the buffer (no copy)"""
        gb = self.grad_buffer
        for l, dW, db in zip(self.layers, gb.dWs, gb.dbs):
            l.weight.grad, l.bias.grad = dW, db

    @classmethod
    def from_sequential(cls, seq):
        lin = [m for m in seq if isinstance(m, torch.nn.Linear)]
        m = cls([lin[0].in_features] + [l.out_features for l in lin])
        for dst, src in zip(m.layers, lin):
            dst.weight.data.copy_(src.weight.data)
            dst.bias.data.copy_(src.bias.data)
        return m

    def forward(self, x):
        """x [N, d0] (any strides; the transposed view of a feature-major buffer is consumed zero-copy)."""
        ws = [l.weight for l in self.layers]
        bs = [l.bias for l in self.layers]
        return _FusedMLPFunc.apply(self, x, *ws, *bs)

    def forward_feature_major(self, x_fm):
        """No-grad fast path: [d0, N] -> [d_n, N]."""
        with torch.no_grad():
            packed = pack_params(self.dims, [l.weight for l in self.layers], [l.bias for l in self.layers])
            return mlp_forward_raw(self.dims, x_fm, packed)


def leaky_relu_init_(linear, negative_slope=0.0):
    """The reference's initialiser for every Linear (permuto_sdf_py/utils/common_utils.py:248-293, `leaky_relu_init`): uniform
    in +-sqrt(3) * gain * sqrt(2 / (n_in + n_out)) with gain = sqrt(2 / (1 + slope^2)), zero bias.  The nets use slope 0 on
    hidden layers and slope 1 (gain 1) on a linear last layer (models.py:161-162, 69-71, 465-472)."""
    gain = math.sqrt(2.0 / (1.0 + negative_slope ** 2))
    std = gain * math.sqrt(2.0 / (linear.in_features + linear.out_features))
    with torch.no_grad():
        linear.weight.uniform_(-std * math.sqrt(3.0), std * math.sqrt(3.0))
        linear.bias.zero_()
    return linear


class _LipshitzNormFunc(torch.autograd.Function):
    """LipshitzMLP.normalization (models.py:98-104) as one launch per direction (csrc/mlp_wide.hip)"""

    @staticmethod
    def forward(ctx, w, c):
        wc, cc = w.detach().contiguous(), c.detach().contiguous()
        wn = torch.empty_like(wc)
        L.call("psdf_lipshitz_normalize_forward", L.c_i(wc.shape[0]), L.c_i(wc.shape[1]), L.ptr(wc), L.ptr(cc), L.ptr(wn),
               L.stream())
        ctx.save_for_backward(wc, cc)
        return wn

    @staticmethod
    def backward(ctx, g):
        wc, cc = ctx.saved_tensors
        dw = torch.empty_like(wc)
        dc = torch.zeros_like(cc)
        L.call("psdf_lipshitz_normalize_backward", L.c_i(wc.shape[0]), L.c_i(wc.shape[1]), L.ptr(wc), L.ptr(cc),
               L.ptr(g.contiguous()), L.ptr(dw), L.ptr(dc), L.stream())
        return dw, dc


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def lipshitz_normalize_all_raw(weights, bounds):
    """LipshitzMLP.normalization (models.py:98-104) of EVERY layer in one launch -> [Wn_l]"""
    ws, cs = [w.detach().contiguous() for w in weights], [c.detach().contiguous() for c in bounds]
    outs = [torch.empty_like(w) for w in ws]
    n = len(ws)
    L.call("psdf_lipshitz_normalize_forward_multi", L.c_i(n), (ctypes.c_int * n)(*[w.shape[0] for w in ws]),
           (ctypes.c_int * n)(*[w.shape[1] for w in ws]), _ptr_array(ws), _ptr_array(cs), _ptr_array(outs), L.stream())
    return outs


def lipshitz_normalize_all_backward_raw(weights, bounds, grads, dc_flat=None):
    """-> ([dW_l], [dc_l]) for dL/dWn_l = grads[l], one launch.  dc_flat: a ZERO-filled [4 * n] tensor to hold the dc_l (one fill
    shared with other buffers of the caller's step), else one is filled here"""
    ws, cs = [w.detach().contiguous() for w in weights], [c.detach().contiguous() for c in bounds]
    gs = [g.contiguous() for g in grads]
    dws = [torch.empty_like(w) for w in ws]
    if dc_flat is None:
        dc_flat = torch.zeros(4 * len(cs), dtype=torch.float32, device=ws[0].device)   # one fill; every [1] slice 16-byte aligned
    dcs = [dc_flat[4 * i:4 * i + 1] for i in range(len(cs))]                        # (the fused optimiser batches aligned tensors)
    n = len(ws)
    L.call("psdf_lipshitz_normalize_backward_multi", L.c_i(n), (ctypes.c_int * n)(*[w.shape[0] for w in ws]),
           (ctypes.c_int * n)(*[w.shape[1] for w in ws]), _ptr_array(ws), _ptr_array(cs), _ptr_array(gs), _ptr_array(dws),
           _ptr_array(dcs), L.stream())
    return dws, dcs


class _LipshitzNormAllFunc(torch.autograd.Function):
    """(W_0, .., W_{n-1}, c_0, .., c_{n-1}) -> (Wn_0, .., Wn_{n-1}): one launch per direction for the whole net"""

    @staticmethod
    def forward(ctx, n, *params):
        ws, cs = params[:n], params[n:]
        ctx.n = n
        ctx.save_for_backward(*[w.detach() for w in ws], *[c.detach() for c in cs])
        return tuple(lipshitz_normalize_all_raw(ws, cs))

    @staticmethod
    def backward(ctx, *grads):
        n = ctx.n
        ws, cs = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        grads = [g if g is not None else torch.zeros_like(w) for g, w in zip(grads, ws)]
        dws, dcs = lipshitz_normalize_all_backward_raw(ws, cs, grads)
        return (None, *dws, *[dc.view_as(c) for dc, c in zip(dcs, cs)])


class LipshitzMLP(torch.nn.Module):
    """Lipschitz-regularised MLP of the colour network (reference: permuto_sdf_py/models/models.py:54-129, used at
    :349-350 as 111 -> 128 -> 128 -> 64 -> 3): every layer's weight is rescaled per row by
    min(1, softplus(c_i) / sum(abs(W_row))) before the Linear; GELU between layers.  Same constructor, methods and
    parameter names as the reference class (`layers.i.weight|bias`, `lipshitz_bound_per_layer.i`).  The normalisation
    is a handful of torch ops on <= 16 K-element tensors; the Linear/GELU stack runs in the fused MFMA evaluator."""

    def __init__(self, in_channels, nr_out_channels_per_layer, last_layer_linear):
        super().__init__()
        self.last_layer_linear = last_layer_linear
        self.dims = [int(in_channels)] + [int(c) for c in nr_out_channels_per_layer]
        self.n_layers = len(self.dims) - 1
        self.layers = torch.nn.ModuleList(
            [torch.nn.Linear(self.dims[i], self.dims[i + 1]) for i in range(self.n_layers)])
        for i, l in enumerate(self.layers):   # reference: leaky_relu_init, slope 0 (hidden) / 1 (linear last layer), models.py:68-71
            last = i == self.n_layers - 1
            leaky_relu_init_(l, 1.0 if (last and last_layer_linear) else 0.0)
        self.weights_per_layer = torch.nn.ParameterList([l.weight for l in self.layers])
        self.biases_per_layer = torch.nn.ParameterList([l.bias for l in self.layers])
        self.lipshitz_bound_per_layer = torch.nn.ParameterList()
        for l in self.layers:
            max_w = torch.max(torch.sum(torch.abs(l.weight.detach()), dim=1))
            self.lipshitz_bound_per_layer.append(torch.nn.Parameter(torch.ones(1) * max_w * 2))
        self.weights_initialized = True

    @staticmethod
    def normalization(w, softplus_ci):
        scale = torch.clamp(softplus_ci / torch.sum(torch.abs(w), dim=1), max=1.0)
        return w * scale[:, None]

    def lipshitz_bound_full(self):
        full = 1
        for c in self.lipshitz_bound_per_layer:
            full = full * torch.nn.functional.softplus(c)
        return full

    def forward(self, x):
        if x.is_cuda:
            ws = list(_LipshitzNormAllFunc.apply(self.n_layers, *self.weights_per_layer, *self.lipshitz_bound_per_layer))
        else:   # parameter bookkeeping on the CPU (checkpoint tests); compute is GPU only and raises below
            ws = [self.normalization(w, torch.nn.functional.softplus(c))
                  for w, c in zip(self.weights_per_layer, self.lipshitz_bound_per_layer)]
        y = _FusedMLPFunc.apply(self, x, *ws, *list(self.biases_per_layer))
        return y if self.last_layer_linear else torch.nn.functional.gelu(y)
