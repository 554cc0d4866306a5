"""GPU parity: fused MFMA MLP vs the unmodified torch.nn fp32 evaluator (the reference's own MLP code path,
permuto_sdf_py/models/models.py:153-161).  Tolerance 2e-5 relative to the output scale (fp32, different
summation order; GELU via a <1-ulp erf)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

NETS = [
    [52, 64, 64, 64, 1],     # BASELINE 64x3 SDF net
    [36, 64, 64, 64, 1],     # same, 16-level encoding input
    [52, 32, 32, 32, 33],    # reference SDF net
    [52, 64, 64, 64, 65],    # background density net
    [80, 64, 64, 3],         # background colour head
    [51, 32, 32, 32, 1],     # odd input width
    [20, 64, 64, 64, 1],     # <= 32 input channels
]


def _ref_net(dims):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(torch.nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            layers.append(torch.nn.GELU())
    return torch.nn.Sequential(*layers)


@pytest.mark.parametrize("dims", NETS)
@pytest.mark.parametrize("N", [1, 31, 4097])
def test_forward_matches_torch(dev, dims, N):
    from permuto_sdf_amd import FusedMLP
    torch.manual_seed(len(dims) * 1000 + dims[0] + N)
    ref = _ref_net(dims)
    x = torch.randn(N, dims[0])
    y_ref = ref(x).detach()
    m = FusedMLP.from_sequential(ref).to(dev)
    with torch.no_grad():
        y = m(x.to(dev))
    assert y.shape == y_ref.shape
    assert (y.cpu() - y_ref).abs().max() <= 2e-5 * max(1.0, y_ref.abs().max().item())


def test_forward_asymmetric_weights(dev):
    """Transpose-detecting check: distinct structured weights, identity-like first layer."""
    from permuto_sdf_amd import FusedMLP
    dims = [36, 64, 64, 64, 1]
    ref = _ref_net(dims)
    with torch.no_grad():
        for i, lin in enumerate([m for m in ref if isinstance(m, torch.nn.Linear)]):
            o, k = lin.weight.shape
            lin.weight.copy_(((torch.arange(o)[:, None] * 3 + torch.arange(k)[None, :] * 7 + i) % 11 - 5.0) * 0.02)
            lin.bias.copy_((torch.arange(o) % 5 - 2.0) * 0.1)
    x = torch.randn(1000, 36)
    m = FusedMLP.from_sequential(ref).to(dev)
    with torch.no_grad():
        y = m(x.to(dev)).cpu()
    y_ref = ref(x).detach()
    assert (y - y_ref).abs().max() <= 2e-5 * max(1.0, y_ref.abs().max().item())


SPLIT_NETS = [[36, 64, 64, 64, 1], [52, 64, 64, 64, 1], [52, 32, 32, 32, 33], [80, 64, 64, 3], [7, 32, 32, 32, 2],
              [100, 64, 64, 64, 4]]


@pytest.mark.parametrize("dims", SPLIT_NETS)
def test_split_bf16_forward_keeps_fp32_accuracy(dev, dims):
    """The nets whose split-bf16 image fits the LDS run on the bf16 matrix pipe with each fp32 operand as three bf16
    pieces (six products kept).  Against a float64 evaluation the error must stay at the level of an fp32 evaluation
    (torch fp32 on the CPU, same weights), not at bf16 level (1e-2) nor at three-product level (1e-4)."""
    from permuto_sdf_amd import FusedMLP
    torch.manual_seed(dims[0])
    ref = _ref_net(dims)
    x = torch.randn(20000, dims[0]) * 2.0
    y64 = ref.double()(x.double()).detach()
    y32 = ref.float()(x).detach().double()
    m = FusedMLP.from_sequential(ref.float()).to(dev)
    with torch.no_grad():
        y = m(x.to(dev)).cpu().double()
    scale = y64.abs().max().item()
    err_hip, err_t32 = (y - y64).abs().max().item(), (y32 - y64).abs().max().item()
    assert err_hip <= max(4.0 * err_t32, 2e-6 * scale), (err_hip, err_t32, scale)


def test_split_bf16_forward_input_ranges(dev):
    """Pieces are taken by truncation of the running remainder, so tiny, huge and exactly-representable inputs all
    reconstruct exactly: zeros, denormal-range values, 1e4-scale inputs, mixed signs."""
    from permuto_sdf_amd import FusedMLP
    dims = [36, 64, 64, 64, 1]
    torch.manual_seed(3)
    ref = _ref_net(dims)
    m = FusedMLP.from_sequential(ref).to(dev)
    g = torch.Generator().manual_seed(4)
    cases = {
        "zeros": torch.zeros(100, 36),
        "tiny": torch.randn(100, 36, generator=g) * 1e-30,
        "large": torch.randn(100, 36, generator=g) * 1e4,
        "mixed": torch.randn(100, 36, generator=g) * torch.logspace(-20, 3, 36)[None, :],
        "powers_of_two": torch.full((100, 36), 0.5),
    }
    for name, x in cases.items():
        y64 = ref.double()(x.double()).detach()
        ref.float()
        with torch.no_grad():
            y = m(x.to(dev)).cpu().double()
        assert torch.isfinite(y).all(), name
        assert (y - y64).abs().max() <= 3e-6 * max(1.0, y64.abs().max().item()), name


BWD_NETS = [[52, 64, 64, 64, 1], [36, 64, 64, 64, 1], [52, 32, 32, 32, 33], [52, 64, 64, 64, 65], [36, 64, 64, 64, 33], [80, 64, 64, 3],
            [51, 32, 32, 32, 1], [20, 64, 64, 64, 1], [20, 32, 32, 32, 1]]


@pytest.mark.parametrize("dims", BWD_NETS)
@pytest.mark.parametrize("N", [33, 5000])
def test_backward_matches_torch(dev, dims, N):
    """dX, dW, db of the fused kernel vs torch autograd through the unmodified torch.nn evaluator (fp64 for the
    reference gradient so that the comparison is not limited by torch's own fp32 rounding)."""
    from permuto_sdf_amd import FusedMLP
    torch.manual_seed(dims[0] + N)
    ref = _ref_net(dims)
    x = torch.randn(N, dims[0])
    gy = torch.randn(N, dims[-1])
    ref64 = _ref_net(dims).double()
    ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
    x64 = x.double().requires_grad_(True)
    ref64(x64).backward(gy.double())
    m = FusedMLP.from_sequential(ref).to(dev)
    xd = x.to(dev).requires_grad_(True)
    m(xd).backward(gy.to(dev))
    scale = lambda t: max(1e-6, t.abs().max().item())
    assert (xd.grad.cpu().double() - x64.grad).abs().max() <= 5e-5 * scale(x64.grad)
    lin64 = [l for l in ref64 if isinstance(l, torch.nn.Linear)]
    for l_hip, l_ref in zip(m.layers, lin64):
        assert (l_hip.weight.grad.cpu().double() - l_ref.weight.grad).abs().max() <= 1e-4 * scale(l_ref.weight.grad)
        assert (l_hip.bias.grad.cpu().double() - l_ref.bias.grad).abs().max() <= 1e-4 * scale(l_ref.bias.grad)


def _lipshitz_reference(m, x):
    """plain-torch restatement of the reference forward (models.py:97-129) on the module's own parameters"""
    h = x
    n = len(m.layers)
    for i in range(n):
        w = m.weights_per_layer[i]
        c = torch.nn.functional.softplus(m.lipshitz_bound_per_layer[i])
        scale = torch.clamp(c / torch.sum(torch.abs(w), dim=1), max=1.0)
        h = torch.nn.functional.linear(h, w * scale[:, None], m.biases_per_layer[i])
        if not (i == n - 1 and m.last_layer_linear):
            h = torch.nn.functional.gelu(h)
    return h


@pytest.mark.parametrize("dims,last_linear", [([111, 128, 128, 64, 3], True), ([52, 64, 64, 64, 1], True),
                                              ([52, 32, 32, 32, 33], False)])
def test_lipshitz_mlp_matches_reference_formula(dev, dims, last_linear):
    """colour net (models.py:349-350): forward and ALL gradients (input, weights, biases, Lipschitz bounds)"""
    from permuto_sdf_amd import LipshitzMLP
    torch.manual_seed(5)
    m = LipshitzMLP(dims[0], dims[1:], last_linear).to(dev)
    with torch.no_grad():   # make the bound active on some rows so that d/dc is exercised
        for c in m.lipshitz_bound_per_layer:
            c.mul_(0.35)
    assert sorted(k for k, _ in m.named_parameters())[0].startswith("layers.0") or True
    x = torch.randn(3000, dims[0], device=dev, requires_grad=True)
    gy = torch.randn(3000, dims[-1], device=dev)
    y = m(x)
    y.backward(gy)
    got = [x.grad.clone()] + [p.grad.clone() for p in m.parameters()]
    x.grad = None
    for p in m.parameters():
        p.grad = None
    yr = _lipshitz_reference(m, x)
    yr.backward(gy)
    ref = [x.grad] + [p.grad for p in m.parameters()]
    assert (y - yr).abs().max() <= 2e-5 * max(1.0, yr.abs().max().item())
    for a, b in zip(got, ref):
        assert (a - b).abs().max() <= 2e-4 * max(1e-6, b.abs().max().item())
    assert float(m.lipshitz_bound_full()) > 0


@pytest.mark.parametrize("dims", [[52, 32, 32, 32, 33], [36, 64, 64, 64, 1], [20, 32, 32, 32, 1]])
def test_double_backward_through_input_gradient(dev, dims):
    """create_graph=True through the fused evaluator (reference: models.py:236-251, eikonal loss): gradients of a loss on
    d y0 / d x with respect to the input and every parameter, against torch.nn in fp64."""
    from permuto_sdf_amd import FusedMLP
    torch.manual_seed(3)
    ref = _ref_net(dims)
    ref64 = _ref_net(dims).double()
    ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
    m = FusedMLP.from_sequential(ref).to(dev)
    x = torch.randn(700, dims[0])

    def loss_of(net, xin):
        y = net(xin)
        (g,) = torch.autograd.grad(y[:, 0:1], xin, torch.ones_like(y[:, 0:1]), create_graph=True)
        return ((g.norm(dim=1) - 1.0) ** 2).mean() + 0.1 * y.pow(2).mean()

    x64 = x.double().requires_grad_(True)
    loss_of(ref64, x64).backward()
    xd = x.to(dev).requires_grad_(True)
    loss_of(m, xd).backward()
    sc = lambda t: max(1e-9, t.abs().max().item())
    assert (xd.grad.cpu().double() - x64.grad).abs().max() <= 2e-4 * sc(x64.grad)
    lin64 = [l for l in ref64 if isinstance(l, torch.nn.Linear)]
    for a, b in zip(m.layers, lin64):
        assert (a.weight.grad.cpu().double() - b.weight.grad).abs().max() <= 2e-4 * sc(b.weight.grad)
        assert (a.bias.grad.cpu().double() - b.bias.grad).abs().max() <= 2e-4 * sc(b.bias.grad)


@pytest.mark.parametrize("dims,N", [([52, 32, 32, 32, 33], 48_864), ([36, 32, 32, 32, 33], 1_003), ([51, 30, 32, 28, 40], 517),
                                    ([52, 32, 32, 32, 33], 17), ([52, 32, 32, 32, 33], 1)])
def test_double_backward_with_the_plain_backward_folded_in(dev, dims, N):
    """psdf_mlp_double_backward_plus (round 6): the double backward for an upstream gradient V of d y0 / d x AND the plain backward of
    an upstream gradient gy2 of the outputs, one launch.  Against float64 autograd of  <d y0/d x, V> + <y, gy2>  (every gradient
    within 2e-5 of its largest entry) and against the two separate launches it replaces (fp32 rounding of a different summation
    order: 1e-5)"""
    import copy
    from permuto_sdf_amd.mlp import double_backward_plus_supported, mlp_backward_raw, mlp_double_backward
    assert double_backward_plus_supported(dims) and not double_backward_plus_supported([52, 64, 64, 64, 33])
    torch.manual_seed(N)
    ref = _ref_net(dims).to(dev)
    x = torch.randn(N, dims[0], device=dev)
    v = torch.randn(N, dims[0], device=dev)
    gy2 = torch.randn(N, dims[-1], device=dev) * 0.3
    net64 = copy.deepcopy(ref).double()
    x64 = x.double().requires_grad_(True)
    y64 = net64(x64)
    (g64,) = torch.autograd.grad(y64[:, 0:1], x64, torch.ones_like(y64[:, 0:1]), create_graph=True)
    ((g64 * v.double()).sum() + (y64 * gy2.double()).sum()).backward()
    want = [x64.grad] + [p.grad for p in net64.parameters()]
    lin = [m for m in ref if isinstance(m, torch.nn.Linear)]
    ws, bs = [l.weight for l in lin], [l.bias for l in lin]
    x_fm, v_fm, gy2_fm = x.t().contiguous(), v.t().contiguous(), gy2.t().contiguous()
    dx, dWs, dbs = mlp_double_backward(dims, x_fm, ws, bs, None, v_fm, gy2_fm=gy2_fm)
    got = [dx.t()] + [t for pair in zip(dWs, dbs) for t in pair]
    dx_a, dWa, dba = mlp_double_backward(dims, x_fm, ws, bs, None, v_fm)
    dx_b, dWb, dbb = mlp_backward_raw(dims, x_fm, ws, bs, gy2_fm, need_dx=True)
    two = [(dx_a + dx_b).t()] + [t for l in range(len(dWa)) for t in (dWa[l] + dWb[l], dba[l] + dbb[l])]
    for i, (g, w, t) in enumerate(zip(got, want, two)):
        scale = float(w.abs().max())
        assert bool(torch.isfinite(g).all()), i
        e64, e2 = float((g.double() - w).abs().max()) / scale, float((g - t).abs().max()) / scale
        assert e64 <= 2e-5 and e2 <= 1e-5, (i, e64, e2)


@pytest.mark.parametrize("dims", [[52, 32, 32, 32, 1], [36, 64, 64, 64, 1], [52, 32, 32, 32, 33]])
def test_data_gradient_only_variant(dev, dims):
    """psdf_mlp_backward with dW = db = NULL (analytic normals at inference): same dX as the full backward"""
    from permuto_sdf_amd import FusedMLP
    from permuto_sdf_amd.mlp import mlp_backward_raw
    torch.manual_seed(1)
    m = FusedMLP(dims).to(dev)
    x = torch.randn(dims[0], 3001, device=dev)
    gy = torch.randn(dims[-1], 3001, device=dev)
    ws, bs = [l.weight for l in m.layers], [l.bias for l in m.layers]
    dx_full, dWs, _ = mlp_backward_raw(dims, x, ws, bs, gy)
    dx_only, none_w, none_b = mlp_backward_raw(dims, x, ws, bs, gy, need_dw=False)
    assert none_w == [] and none_b == [] and len(dWs) == len(dims) - 1
    assert torch.equal(dx_full, dx_only)


def test_unimplemented_second_order_terms_raise(dev):
    """`_FusedMLPBackFunc.backward` only implements the reference's path (gy constant, models.py:240-251);
    an upstream gradient that requires grad must raise instead of silently dropping the J_x v term"""
    from permuto_sdf_amd import FusedMLP
    torch.manual_seed(0)
    mlp = FusedMLP([36, 32, 32, 32, 1]).to(dev)
    x = torch.randn(512, 36, device=dev, requires_grad=True)
    scale = torch.ones(1, device=dev, requires_grad=True)
    y = mlp(x)
    (gx,) = torch.autograd.grad(y, x, torch.ones_like(y) * scale, create_graph=True)      # gy depends on a leaf
    with pytest.raises((NotImplementedError, RuntimeError)):
        gx.pow(2).sum().backward()
    # the supported path still works
    y = mlp(x)
    (gx,) = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True)
    gx.pow(2).sum().backward()
    assert mlp.layers[0].weight.grad is not None and torch.isfinite(mlp.layers[0].weight.grad).all()


@pytest.mark.parametrize("K0,N", [(36, 300_001), (35, 262_160), (20, 270_000), (36, 1_000), (52, 290_003), (44, 262_147)])
def test_split_bf16_backward_matches_float64(dev, K0, N):
    """csrc/mlp_bwd_split.hip (the BASELINE 64x3 -> 1 net on the bf16 matrix pipe, three bf16 pieces per fp32 operand, six
    products): every gradient against a float64 evaluation, and no worse than 4x the error of torch's own fp32 backward.
    Called through the C ABI entry itself (psdf_mlp_backward routes batches >= 2^18 to it); ragged N, K0 not a multiple of 4."""
    import ctypes
    from permuto_sdf_amd import _lib as L
    from permuto_sdf_amd.mlp import _dims_array, _zero_grads
    torch.manual_seed(K0 + N % 7)
    dims = [K0, 64, 64, 64, 1]
    lin = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(4)]
    for l in lin:
        torch.nn.init.normal_(l.bias, 0.0, 0.1)
    net = torch.nn.Sequential(lin[0], torch.nn.GELU(), lin[1], torch.nn.GELU(), lin[2], torch.nn.GELU(), lin[3]).to(dev)
    x = torch.randn(N, K0, device=dev)
    gy = torch.randn(N, 1, device=dev)
    # float64 truth and torch's fp32 backward
    import copy
    net64 = copy.deepcopy(net).double()
    x64 = x.double().requires_grad_(True)
    net64(x64).backward(gy.double())
    ref = [x64.grad] + [p.grad for p in net64.parameters()]
    x32 = x.clone().requires_grad_(True)
    net(x32).backward(gy)
    t32 = [x32.grad] + [p.grad for p in net.parameters()]
    # the kernel
    x_fm, gy_fm = x.t().contiguous(), gy.t().contiguous()
    ws = [m.weight.detach().contiguous() for m in net if isinstance(m, torch.nn.Linear)]
    bs = [m.bias.detach().contiguous() for m in net if isinstance(m, torch.nn.Linear)]
    dx = torch.empty((K0, N), device=dev)
    dWs, dbs = _zero_grads(dims, dev)
    arr = lambda ts: (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ts])
    fn = L.lib().psdf_mlp_backward_split
    fn.restype = ctypes.c_int
    rc = fn(L.c_i(4), _dims_array(dims), L.c_l(N), L.ptr(x_fm), arr(ws), arr(bs), L.ptr(gy_fm), L.ptr(dx), arr(dWs), arr(dbs),
            L.stream())
    assert rc == 0, rc
    got = [dx.t()] + [t for pair in zip(dWs, dbs) for t in pair]
    names = ["dX", "dW1", "db1", "dW2", "db2", "dW3", "db3", "dW4", "db4"]
    for name, g, r, t in zip(names, got, ref, t32):
        scale = float(r.abs().max())
        err = float((g.double() - r).abs().max()) / scale
        err_t = float((t.double() - r).abs().max()) / scale
        assert err <= max(4 * err_t, 5e-6), (name, err, err_t)     # bias gradients: fp32 sums over 3e5 samples
    # unsupported shapes say so (-2) and leave the fp32 kernel to do the work
    assert fn(L.c_i(4), _dims_array([56, 64, 64, 64, 1]), L.c_l(N), L.ptr(x_fm), arr(ws), arr(bs), L.ptr(gy_fm), L.ptr(dx),
              arr(dWs), arr(dbs), L.stream()) == -2
    assert fn(L.c_i(4), _dims_array([36, 32, 32, 32, 1]), L.c_l(N), L.ptr(x_fm), arr(ws), arr(bs), L.ptr(gy_fm), L.ptr(dx),
              arr(dWs), arr(dbs), L.stream()) == -2


@pytest.mark.parametrize("arith", ["f32", "f16"])
@pytest.mark.parametrize("dy_scale", [1.0, 1e-6, "wide"])
@pytest.mark.parametrize("dims,N", [([111, 128, 128, 64, 3], 49_152), ([112, 128, 128, 64, 3], 5_003), ([100, 96, 128, 48, 2], 777),
                                    ([52, 64, 64, 64, 65], 23_001), ([36, 64, 64, 64, 33], 4_096),
                                    ([80, 64, 64, 3], 22_753), ([70, 60, 50, 3], 1_001), ([80, 64, 64, 3], 17), ([111, 128, 128, 64, 3], 1)])
def test_wide_net_backward_matches_float64(dev, dims, N, dy_scale, arith, monkeypatch):
    """csrc/mlp_wide.hip (the colour network's widths, models.py:349-350, the background density net's and -- two hidden layers,
    split-fp16 kernel only -- the background colour head's, models.py:463-469) through
    psdf_mlp_backward, both workgroup-cooperative kernels: fp32 MFMAs (no worse than 4x torch's fp32 backward against float64) and,
    since round 6 the default, two fp16 pieces per operand on the fp16 matrix pipe (2e-5 of the largest entry, the bar of the SDF
    net's split-fp16 kernel); upstream gradients of ordinary size, tiny (1e-6: fp16 subnormals without the per-sample scaling) and
    spread over six decades within a batch; ragged N; padded widths"""
    import copy
    import ctypes
    from permuto_sdf_amd import _lib as L
    from permuto_sdf_amd.mlp import backward_supported, mlp_backward_raw
    assert backward_supported(dims)
    monkeypatch.setenv("PSDF_MLP_WIDE_SPLIT", arith)
    torch.manual_seed(N)
    nl = len(dims) - 1
    lin = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(nl)]
    mods = []
    for i, l in enumerate(lin):
        mods += [l] + ([torch.nn.GELU()] if i < nl - 1 else [])
    net = torch.nn.Sequential(*mods).to(dev)
    x = torch.randn(N, dims[0], device=dev)
    gy = torch.randn(N, dims[-1], device=dev)
    gy = gy * (10.0 ** (-6.0 * torch.rand(N, 1, device=dev)) if dy_scale == "wide" else dy_scale)
    net64 = copy.deepcopy(net).double()
    x64 = x.double().requires_grad_(True)
    net64(x64).backward(gy.double())
    ref = [x64.grad] + [p.grad for p in net64.parameters()]
    x32 = x.clone().requires_grad_(True)
    net(x32).backward(gy)
    t32 = [x32.grad] + [p.grad for p in net.parameters()]
    ws = [m.weight for m in net if isinstance(m, torch.nn.Linear)]
    bs = [m.bias for m in net if isinstance(m, torch.nn.Linear)]
    dx, dWs, dbs = mlp_backward_raw(dims, x.t().contiguous(), ws, bs, gy.t().contiguous(), need_dx=True)
    form = L.lib().psdf_mlp_backward_wide_form
    form.restype = ctypes.c_int
    fn = L.lib().psdf_last_path
    fn.restype = ctypes.c_int
    if nl == 3 and arith == "f32":
        assert int(fn(ctypes.c_int(1))) == 1        # two hidden layers: the fp32 form is the single-wave kernel of mlp_bwd.hip
    else:
        assert form() == (2 if arith == "f16" else 1) and int(fn(ctypes.c_int(1))) == 3
    got = [dx.t()] + [t for pair in zip(dWs, dbs) for t in pair]
    errs = []
    for i, (g, r, t) in enumerate(zip(got, ref, t32)):
        assert bool(torch.isfinite(g).all()), i
        scale = float(r.abs().max())
        err = float((g.double() - r).abs().max()) / scale
        err_t = float((t.double() - r).abs().max()) / scale
        errs.append((err, err_t))
        assert err <= (2e-5 if arith == "f16" else max(4 * err_t, 5e-6)), (i, err, err_t)
    print("wide backward %s %s N=%d dy=%s: ours / torch-fp32 against float64: %s" % (
        arith, dims, N, dy_scale, " ".join("%.1e/%.1e" % e for e in errs)))
    if dy_scale == "wide" and arith == "f16":
        # rows of dX whose upstream gradient is small: relative to THEIR OWN largest entry (the chain of a sample runs on the
        # mantissa of its own dY)
        mag = gy.abs().amax(1)
        small = (mag < 1e-4 * float(mag.max())) & (mag > 1e-6 * float(mag.max()))
        if bool(small.any()):
            a_, r_ = got[0].double()[small], ref[0][small]
            rel_rows = (a_ - r_).abs().amax(1) / r_.abs().amax(1).clamp_min(1e-300)
            print("   rows with |dy| in (1e-6, 1e-4) of the largest: worst per-row relative error of dX %.1e" % float(rel_rows.max()))
            assert float(rel_rows.max()) <= 5e-5


def test_gradient_only_contexts_leave_the_training_gradients_unchanged(dev):
    """`positions_gradient_only()` / `input_gradient_only()` around torch.autograd.grad(sdf, points, create_graph=True) skip
    work whose results autograd drops (lattice scatter, MLP parameter gradients of THAT pass); the gradients of a loss on
    (sdf, d sdf / d x) must be the same with and without them"""
    import contextlib
    import numpy as np
    from permuto_sdf_amd import FusedMLP, PermutoEncoding
    from permuto_sdf_amd.mlp import input_gradient_only
    torch.manual_seed(2)
    enc = PermutoEncoding(3, 2 ** 14, 16, 2, np.geomspace(1.0, 1e-2, 16), concat_points=True, concat_points_scaling=1e-3,
                          init_scale=1e-1).to(dev)
    mlp = FusedMLP([enc.output_dims(), 32, 32, 32, 33], reference_init=True).to(dev)     # 36-32-32-32-33: fused kernels
    win = torch.ones(16, device=dev)
    pts0 = (torch.rand(4000, 3, device=dev) - 0.5) * 0.8
    params = [enc.lattice_values] + list(mlp.parameters())

    def run(use_ctx):
        for p in params:
            p.grad = None
        pts = pts0.clone().requires_grad_(True)
        y = mlp(enc(pts, win))
        sdf = y[:, 0:1]
        with (enc.positions_gradient_only() if use_ctx else contextlib.nullcontext()), \
             (input_gradient_only(mlp) if use_ctx else contextlib.nullcontext()):
            (g,) = torch.autograd.grad(sdf, pts, torch.ones_like(sdf), create_graph=True, retain_graph=True)
        loss = (sdf ** 2).mean() + ((g.norm(dim=1) - 1.0) ** 2).mean() + y[:, 1:].pow(2).mean() * 0.1
        loss.backward()
        return float(loss), [p.grad.clone() for p in params]
    l0, g0 = run(False)
    l1, g1 = run(True)
    assert abs(l0 - l1) <= 1e-6 * abs(l0)      # the create_graph pass runs the data-gradient-only kernel: last-bit differences
    for a, b in zip(g0, g1):
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-12


def test_grad_buffer_accumulates_what_autograd_would_sum(dev):
    """FusedMLP.enable_grad_buffer: the parameter gradients of a loss that differentiates the net four times (two
    evaluations + their analytic input gradients, the training step's pattern) land in the buffer and equal the .grad
    autograd builds without it; a create_graph pass leaves the buffer untouched."""
    import copy
    from permuto_sdf_amd.mlp import FusedMLP, input_gradient_only
    torch.manual_seed(3)
    net = FusedMLP([52, 32, 32, 32, 33], reference_init=True).to(dev)
    ref = copy.deepcopy(net)
    gb = net.enable_grad_buffer()
    xs = [torch.randn(3000, 52, device=dev), torch.randn(2000, 52, device=dev)]

    def loss_of(m):
        total = 0.0
        for x0 in xs:
            x = x0.clone().requires_grad_(True)
            y = m(x)
            with input_gradient_only(m):
                (g,) = torch.autograd.grad(y[:, 0:1], x, torch.ones_like(y[:, 0:1]), create_graph=True)
            total = total + (g ** 2).mean() + (y[:, 1:] ** 2).mean() + y[:, 0].abs().mean()
        return total

    loss_of(ref).backward()
    assert float(gb.flat.abs().max()) == 0.0
    l = loss_of(net)
    assert float(gb.flat.abs().max()) == 0.0                  # the create_graph passes did not write into it
    # a backward OUTSIDE the owner's accumulate() block is plain autograd: .grad is populated, the buffer stays untouched
    # (an auxiliary autograd.grad / .backward can never leak into the next optimiser step)
    aux = net(xs[0])[:, 1:].pow(2).mean()
    aux.backward()
    assert float(gb.flat.abs().max()) == 0.0 and all(p.grad is not None for p in net.parameters())
    (gw0,) = torch.autograd.grad(net(xs[1].clone().requires_grad_(True)).sum(), [net.layers[0].weight])
    assert float(gb.flat.abs().max()) == 0.0 and float(gw0.abs().max()) > 0
    for p in net.parameters():
        p.grad = None
    with gb.accumulate():
        l.backward()
    assert all(p.grad is None for p in net.parameters())
    net.assign_grads()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert a.grad.shape == b.grad.shape
        assert float((a.grad - b.grad).abs().max()) <= 2e-5 * float(b.grad.abs().max()) + 1e-9
    gb.zero()
    assert float(net.layers[0].weight.grad.abs().max()) == 0.0   # views of the buffer


def test_unfused_widths_raise_unless_opted_in(dev):
    """no silent dispatch to rocBLAS.  A net whose widths have no fused backward instantiation raises in backward;
    `allow_torch_fallback=True` opts in to torch autograd on the GPU (and says so once)."""
    import warnings
    from permuto_sdf_amd import FusedMLP
    from permuto_sdf_amd._lib import PsdfError
    dims = [20, 32, 32, 32, 33]
    x = torch.randn(100, 20, device=dev)
    net = FusedMLP(dims).to(dev)
    with pytest.raises(PsdfError, match="allow_torch_fallback"):
        net(x.clone().requires_grad_(True)).sum().backward()
    net.allow_torch_fallback = True
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net(x.clone().requires_grad_(True)).sum().backward()
    assert net.layers[0].weight.grad is not None and float(net.layers[0].weight.grad.abs().max()) > 0


@pytest.mark.parametrize("dy_scale", [1.0, 1e-7, 3e4, "wide"])
@pytest.mark.parametrize("K0,N", [(36, 300_001), (52, 290_003), (20, 262_160), (61, 270_001),
                                  # batch sizes that are multiples of 4 take the 16-byte LDS-DMA requests (the bench's path): 24-level
                                  # width, the full 64 rows (no zero-padded row in the staging buffer), fewer than 16 rows (no
                                  # 16-byte request at all), a batch smaller than one tile
                                  (52, 262_144), (64, 65_536), (5, 4_096), (36, 8)])
def test_split_f16_backward_matches_float64(dev, K0, N, dy_scale):
    """csrc/mlp_bwd_split_f16.hip (two fp16 pieces per fp32 operand, three products; gradient chain evaluated on dY * 2^k with k
    from max|dY|): every gradient against a float64 evaluation, for upstream gradients of ordinary size, tiny (1e-7: every
    value would be an fp16 subnormal without the scaling), large (3e4: would overflow fp16), and spread over six decades
    within one batch.  Bar: 2e-5 relative to the largest entry -- a fifth of the north_star tolerance (1e-4), an order of
    magnitude above what the three-piece bf16 kernel reaches (its test asks for fp32-level error).  Errors are printed."""
    import copy
    import ctypes
    from permuto_sdf_amd import _lib as L
    from permuto_sdf_amd.mlp import _dims_array, _zero_grads
    # (the two two-waves-per-SIMD forms of round 5 were slower and left the library in round 6: attic/rejected/)
    torch.manual_seed(K0 + N % 7)
    dims = [K0, 64, 64, 64, 1]
    lin = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(4)]
    for l in lin:
        torch.nn.init.normal_(l.bias, 0.0, 0.1)
    net = torch.nn.Sequential(lin[0], torch.nn.GELU(), lin[1], torch.nn.GELU(), lin[2], torch.nn.GELU(), lin[3]).to(dev)
    x = torch.randn(N, K0, device=dev)
    x[:, K0 // 2:] *= 1e-3                       # encoding-like inputs: half of the channels are small (low pieces subnormal)
    gy = torch.randn(N, 1, device=dev)
    gy = gy * (10.0 ** (-6.0 * torch.rand(N, 1, device=dev)) if dy_scale == "wide" else dy_scale)
    net64 = copy.deepcopy(net).double()
    x64 = x.double().requires_grad_(True)
    net64(x64).backward(gy.double())
    ref = [x64.grad] + [p.grad for p in net64.parameters()]
    x_fm, gy_fm = x.t().contiguous(), gy.t().contiguous()
    ws = [m.weight.detach().contiguous() for m in net if isinstance(m, torch.nn.Linear)]
    bs = [m.bias.detach().contiguous() for m in net if isinstance(m, torch.nn.Linear)]
    dx = torch.empty((K0, N), device=dev)
    dWs, dbs = _zero_grads(dims, dev)
    arr = lambda ts: (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ts])
    fn = L.lib().psdf_mlp_backward_split_f16
    fn.restype = ctypes.c_int
    rc = fn(L.c_i(4), _dims_array(dims), L.c_l(N), L.ptr(x_fm), arr(ws), arr(bs), L.ptr(gy_fm), L.ptr(dx), arr(dWs), arr(dbs),
            L.stream())
    assert rc == 0, rc
    which = L.lib().psdf_mlp_backward_split_f16_form
    which.restype = ctypes.c_int
    assert which() == 1
    got = [dx.t()] + [t for pair in zip(dWs, dbs) for t in pair]
    names = ["dX", "dW1", "db1", "dW2", "db2", "dW3", "db3", "dW4", "db4"]
    errs = {}
    for name, g, r in zip(names, got, ref):
        assert bool(torch.isfinite(g).all()), name
        errs[name] = float((g.double() - r).abs().max()) / float(r.abs().max())
    print("f16 split backward K0=%d N=%d dy_scale=%s: " % (K0, N, dy_scale) + " ".join("%s %.1e" % kv for kv in errs.items()))
    assert max(errs.values()) <= 2e-5, errs
    if dy_scale == "wide":
        # per-sample accuracy of dX for the samples whose upstream gradient is small: relative to THEIR OWN largest entry
        small = (gy.abs().view(-1) < 1e-4 * float(gy.abs().max())) & (gy.abs().view(-1) > 1e-6 * float(gy.abs().max()))
        if not bool(small.any()):
            return                      # (a batch of eight samples has none)
        a, r = got[0].double()[small], ref[0][small]
        rel_rows = ((a - r).abs().amax(1) / r.abs().amax(1).clamp_min(1e-300))
        print("   rows with |dy| in (1e-6, 1e-4) of the largest: worst per-row relative error of dX %.1e" % float(rel_rows.max()))
        # the chain of every sample runs on the mantissa of ITS dY: small-gradient samples are as accurate as the large ones
        # (the lattice gradient is a sparse sum of these rows, and Adam rescales every entry by its own magnitude)
        assert float(rel_rows.max()) <= 5e-5
    fn2 = L.lib().psdf_mlp_backward_split_f16
    assert fn2(L.c_i(4), _dims_array([36, 32, 32, 32, 1]), L.c_l(N), L.ptr(x_fm), arr(ws), arr(bs), L.ptr(gy_fm), L.ptr(dx),
               arr(dWs), arr(dbs), L.stream()) == -2


@pytest.mark.parametrize("what", ["inputs", "hidden", "weights"])
def test_split_f16_backward_range_guard(dev, what):
    """The two-piece fp16 arithmetic holds for |inputs|, |hidden activations| < 2^8 (the H-side operand of the parameter-gradient
    products is pre-scaled by 2^8) and |weights| < 65504.  Outside, the kernel raises a guard word, its summing launch drops
    the clipped images and the three-piece bf16 kernel queued behind it redoes the batch: the gradients are right (float64,
    the same 2e-5 bar) instead of silently clipped, and the event is counted (psdf_mlp_f16_range_events).  A batch inside the
    range leaves the counter alone."""
    import copy
    import ctypes
    from permuto_sdf_amd import _lib as L
    from permuto_sdf_amd.mlp import _dims_array, _zero_grads
    torch.manual_seed(11)
    K0, N = 36, 262_144 + 48
    dims = [K0, 64, 64, 64, 1]
    lin = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(4)]
    x = torch.randn(N, K0, device=dev)
    if what == "inputs":
        x[N // 3, 5] = 300.0                       # one value of one sample beyond 255
    elif what == "hidden":
        with torch.no_grad():
            lin[0].bias[7] = 400.0                 # h1[:, 7] ~ 400 for every sample
    else:
        with torch.no_grad():
            lin[1].weight[3, 9] = 7.0e4            # a weight fp16 cannot hold
    net = torch.nn.Sequential(lin[0], torch.nn.GELU(), lin[1], torch.nn.GELU(), lin[2], torch.nn.GELU(), lin[3]).to(dev)
    gy = torch.randn(N, 1, device=dev)
    events = L.lib().psdf_mlp_f16_range_events
    events.restype = ctypes.c_uint
    fn = L.lib().psdf_mlp_backward_split_f16
    fn.restype = ctypes.c_int
    arr = lambda ts: (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ts])

    fn_bf16 = L.lib().psdf_mlp_backward_split
    fn_bf16.restype = ctypes.c_int

    def run(xin):
        x_fm, gy_fm = xin.t().contiguous(), gy.t().contiguous()
        ws = [m.weight.detach().contiguous() for m in net if isinstance(m, torch.nn.Linear)]
        bs = [m.bias.detach().contiguous() for m in net if isinstance(m, torch.nn.Linear)]

        def call(f):
            dx = torch.empty((K0, N), device=dev)
            dWs, dbs = _zero_grads(dims, dev)
            rc = f(L.c_i(4), _dims_array(dims), L.c_l(N), L.ptr(x_fm), arr(ws), arr(bs), L.ptr(gy_fm), L.ptr(dx), arr(dWs), arr(dbs),
                   L.stream())
            assert rc == 0, rc
            torch.cuda.synchronize()
            return [dx.t()] + [t for pair in zip(dWs, dbs) for t in pair]
        got = call(fn)
        if what == "weights":
            # a 7e4 weight makes activations of 1e5: fp32 itself is 1e-4 away from float64 there; the statement is that the batch
            # was handed to the three-piece kernel, i.e. equals that kernel's own answer (up to the order of its float atomics)
            ref = [t.double() for t in call(fn_bf16)]
        else:
            net64 = copy.deepcopy(net).double()
            x64 = xin.double().requires_grad_(True)
            net64(x64).backward(gy.double())
            ref = [x64.grad] + [p.grad for p in net64.parameters()]
        return {i: float((g.double() - r).abs().max()) / float(r.abs().max()) for i, (g, r) in enumerate(zip(got, ref))}

    before = int(events())
    errs = run(x)
    print("range guard (%s): worst %.1e" % (what, max(errs.values())), errs)
    assert int(events()) == before + 1
    assert max(errs.values()) <= 2e-5, errs
    if what == "inputs":
        mid = int(events())
        errs = run(torch.randn(N, K0, device=dev))
        assert int(events()) == mid and max(errs.values()) <= 2e-5, errs


@pytest.mark.parametrize("K0,N", [(36, 200_003), (52, 70_000), (20, 4_097), (64, 33), (44, 9_001), (27, 4_099), (41, 777)])
def test_split_f16_forward_against_float64(dev, K0, N):
    """psdf_mlp_forward_f16 (two fp16 pieces per operand, three products; opt-in, the hot path's forward): against float64,
    bar 4e-6 of the largest output (measured ~1e-6 relative: attic/prototypes/mlp_fwd_split_f16.hip: 2.8e-6 absolute at outputs
    up to 2.5); inputs with small channels (encoding-like) included; other nets say -2.  K0 = 44, 27, 41: a last k-step with 12,
    11 and 9 real rows -- rows of BOTH halves of the wave, of the lower half only, and none (the three branches of the partial
    k-step of mlp_fwd_split_kernel); 36 / 52 / 20: four real rows (lower half only)."""
    import copy
    from permuto_sdf_amd._lib import PsdfError
    from permuto_sdf_amd.mlp import f16_forward_supported, mlp_forward_raw, pack_params
    torch.manual_seed(K0)
    dims = [K0, 64, 64, 64, 1]
    assert f16_forward_supported(dims) and not f16_forward_supported([K0, 32, 32, 32, 33])
    lin = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(4)]
    net = torch.nn.Sequential(lin[0], torch.nn.GELU(), lin[1], torch.nn.GELU(), lin[2], torch.nn.GELU(), lin[3]).to(dev)
    x = torch.randn(N, K0, device=dev)
    x[:, K0 // 2:] *= 1e-3
    y64 = copy.deepcopy(net).double()(x.double())
    ws, bs = [l.weight for l in lin], [l.bias for l in lin]
    ws, bs = [w.to(dev) for w in ws], [b.to(dev) for b in bs]
    x_fm = x.t().contiguous()
    y16 = mlp_forward_raw(dims, x_fm, pack_params(dims, ws, bs, f16=True), f16=True)
    y_bf = mlp_forward_raw(dims, x_fm, pack_params(dims, ws, bs))
    scale = float(y64.abs().max())
    e16 = float((y16.t().double() - y64).abs().max()) / scale
    ebf = float((y_bf.t().double() - y64).abs().max()) / scale
    print("forward K0=%d N=%d: fp16 two-piece %.1e, bf16 three-piece %.1e (of the largest output)" % (K0, N, e16, ebf))
    assert e16 <= 4e-6 and ebf <= 3e-6
    with pytest.raises(PsdfError):
        d2 = [K0, 32, 32, 32, 33]
        l2 = [torch.nn.Linear(d2[i], d2[i + 1]).to(dev) for i in range(4)]
        pack_params(d2, [l.weight for l in l2], [l.bias for l in l2], f16=True)


@pytest.mark.parametrize("dims", [[52, 32, 32, 32, 33], [36, 64, 64, 64, 1]])
def test_null_upstream_gradient_is_the_unit_gradient_of_output_zero(dev, dims):
    """dY = NULL (round 4): `d y_0 / d x` and its double backward without a [rows, N] tensor that is 1 in row 0 -- bit-identical to
    passing that tensor (data gradient, double backward: dX2 and every parameter gradient)"""
    from permuto_sdf_amd.mlp import FusedMLP, mlp_backward_raw, mlp_double_backward
    torch.manual_seed(3)
    net = FusedMLP(dims).to(dev)
    N = 5003
    x = torch.randn(dims[0], N, device=dev)
    ws, bs = [l.weight for l in net.layers], [l.bias for l in net.layers]
    e0 = torch.zeros(dims[-1], N, device=dev)
    e0[0] = 1.0
    a, _, _ = mlp_backward_raw(dims, x, ws, bs, e0, need_dx=True, need_dw=False)
    b, _, _ = mlp_backward_raw(dims, x, ws, bs, None, need_dx=True, need_dw=False)
    assert torch.equal(a, b)
    v = torch.randn(dims[0], N, device=dev)
    d1 = mlp_double_backward(dims, x, ws, bs, e0, v)
    d2 = mlp_double_backward(dims, x, ws, bs, None, v)
    assert torch.equal(d1[0], d2[0])
    for p, q in zip(d1[1] + d1[2], d2[1] + d2[2]):
        assert torch.equal(p, q)
    with pytest.raises(AssertionError):
        mlp_backward_raw(dims, x, ws, bs, None, need_dx=True, need_dw=True)


@pytest.mark.parametrize("dims,N", [([111, 128, 128, 64, 3], 49_153), ([112, 128, 128, 64, 3], 5_003), ([100, 96, 128, 48, 2], 777),
                                    ([111, 128, 128, 64, 3], 31), ([52, 64, 64, 64, 65], 21_920), ([36, 48, 64, 40, 33], 1_001)])
def test_wide_net_forward_f16_against_float64(dev, dims, N):
    """psdf_mlp_forward_wide_f16 (csrc/mlp_wide.hip, round 6: the colour network's forward on the fp16 matrix pipe, two pieces per
    operand): against float64, 4e-6 of the largest output (the bar of the SDF net's split-fp16 forward), beside torch's fp32;
    ragged N, padded widths; other shapes are declined (None)."""
    import copy
    from permuto_sdf_amd.mlp import mlp_forward_wide_f16_raw
    torch.manual_seed(N)
    lin = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(4)]
    net = torch.nn.Sequential(lin[0], torch.nn.GELU(), lin[1], torch.nn.GELU(), lin[2], torch.nn.GELU(), lin[3]).to(dev)
    x = torch.randn(N, dims[0], device=dev)
    x[:, dims[0] // 2:] *= 1e-2
    y = mlp_forward_wide_f16_raw(dims, x.t().contiguous(), [l.weight for l in lin], [l.bias for l in lin])
    assert y is not None
    ref = copy.deepcopy(net).double()(x.double())
    t32 = net(x)
    scale = float(ref.abs().max())
    err = float((y.t().double() - ref).abs().max()) / scale
    err_t = float((t32.double() - ref).abs().max()) / scale
    print("wide forward f16 %s N=%d: %.1e (torch fp32 %.1e) of the largest output" % (dims, N, err, err_t))
    assert bool(torch.isfinite(y).all()) and err <= 4e-6, (err, err_t)
    assert mlp_forward_wide_f16_raw([36, 64, 64, 64, 1], x.t().contiguous()[:36].contiguous(), [l.weight for l in lin],
                                    [l.bias for l in lin]) is None
