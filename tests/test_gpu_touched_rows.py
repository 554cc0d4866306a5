"""touched-rows AdamW.  Semantics, stated: the update is the reference's dense torch.optim.AdamW
(train_permuto_sdf.py:293-304: betas (0.9, 0.99), eps 1e-15, weight_decay 0 on the lattices) -- moments of rows a batch does
not touch KEEP DECAYING and keep moving the parameter, exactly as torch does -- and only the blocks of table rows that no
batch has ever touched (gradient and both moments exactly zero, where the dense update is the identity) are skipped.
Checked here: (1) the forward's touched map covers every row the backward / double backward writes; (2) several training
steps with changing batches are BIT-IDENTICAL to torch.optim.AdamW on a dense autograd gradient; (3) how much is skipped."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _enc(dev, L_=12, T=2 ** 14, seed=0):
    from permuto_sdf_amd import PermutoEncoding
    torch.manual_seed(seed)
    return PermutoEncoding(3, T, L_, 2, np.geomspace(1.0, 1e-3, L_), concat_points=True, concat_points_scaling=1e-3,
                           init_scale=1e-2).to(dev)


def _loss(enc, w1, pts, win):
    """first- and second-order use of the encoding, like the SDF net's eikonal term (models.py:236-251)"""
    pts = pts.clone().requires_grad_(True)
    sdf = torch.tanh(enc(pts, win) @ w1).sum(1, keepdim=True)
    (grad,) = torch.autograd.grad(sdf, pts, torch.ones_like(sdf), create_graph=True)
    return (sdf ** 2).mean() + ((grad.norm(dim=1) - 1.0) ** 2).mean() * 0.1


def test_touched_map_covers_every_written_row(dev):
    enc = _enc(dev)
    tr = enc.enable_touched_rows(block_rows_log2=5)
    win = torch.ones(12, device=dev)
    w1 = torch.randn(enc.output_dims(), 3, device=dev)
    pts = (torch.rand(3000, 3, device=dev) - 0.5) * 0.6
    with tr.accumulate():
        _loss(enc, w1, pts, win).backward()
    assert enc.lattice_values.grad is None                       # the gradient went to the persistent buffer
    g = tr.grad
    assert float(g.abs().max()) > 0
    rows = (g != 0).any(-1)                                      # [L, T]
    blocks = rows.view(12, -1, 32).any(-1)
    assert bool((tr.touched.bool() | ~blocks).all())             # written => touched
    frac = float(tr.touched.float().mean())
    assert 0.0 < frac < 0.9                                      # and the map is not trivially "everything"
    # no-grad forwards (occupancy refresh, importance sampling) do not mark
    before = tr.touched.clone()
    with torch.no_grad():
        enc((torch.rand(5000, 3, device=dev) - 0.5) * 1.9, win)
    assert torch.equal(before, tr.touched)


def test_block_adamw_is_bit_identical_to_dense_torch_adamw(dev):
    from permuto_sdf_amd.optim import FusedAdamW
    win = torch.ones(12, device=dev)
    ref, ours = _enc(dev, seed=3), _enc(dev, seed=3)
    assert torch.equal(ref.lattice_values, ours.lattice_values)
    w1 = torch.randn(ref.output_dims(), 3, device=dev)
    opt_ref = torch.optim.AdamW([ref.lattice_values], lr=1e-3, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.0)
    opt = FusedAdamW([ours.lattice_values], lr=1e-3, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.0)
    dense_fused = FusedAdamW([ref.lattice_values], lr=1e-3, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.0)
    tr = ours.enable_touched_rows()
    opt.attach(ours.lattice_values, tr)
    g = torch.Generator(device="cpu").manual_seed(5)
    # the reference semantic with the SAME gradients: dense FusedAdamW (== torch.optim.AdamW, tests/test_gpu_optim.py) fed
    # the buffer's gradient; then torch.optim.AdamW itself on an autograd gradient within rounding of the scatter order
    ref2 = _enc(dev, seed=3)
    opt_t = torch.optim.AdamW([ref2.lattice_values], lr=1e-3, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.0)
    skipped = []
    for it in range(6):
        # batches move: rows touched in one step are NOT touched in a later one, their moments must keep decaying
        centre = torch.tensor([[0.3 * np.cos(it), 0.3 * np.sin(it), 0.0]])
        pts = ((torch.rand(2000, 3, generator=g) - 0.5) * 0.25 + centre).to(dev)
        with tr.accumulate():
            _loss(ours, w1, pts, win).backward()
            _loss(ours, w1, pts * 0.5, win).backward()            # two backward calls accumulate into the one buffer
        grad = tr.grad.clone()
        ref.lattice_values.grad = grad.clone()
        dense_fused.step()
        ref2.lattice_values.grad = grad.clone()
        opt_t.step()
        skipped.append(1.0 - float((tr.touched | tr.active).float().mean()))
        opt.step()
        assert float(tr.grad.abs().max()) == 0.0 and int(tr.touched.sum()) == 0      # zero-fill fused into the update
        assert torch.equal(ours.lattice_values, ref.lattice_values), it               # bit-identical to the dense kernel
        assert torch.allclose(ours.lattice_values, ref2.lattice_values, rtol=0, atol=2e-7), it   # and to torch.optim.AdamW
    st, st_ref = opt.state[ours.lattice_values], dense_fused.state[ref.lattice_values]
    assert torch.equal(st["exp_avg"], st_ref["exp_avg"]) and torch.equal(st["exp_avg_sq"], st_ref["exp_avg_sq"])
    # rows touched only in step 0 still have (decayed, non-zero) moments: untouched rows are NOT frozen
    assert float(st["exp_avg"].abs().max()) > 0
    assert skipped[0] > 0.3 and skipped[-1] > 0.1, skipped       # a real share of the table is never read


def test_trainer_steps_with_and_without_touched_rows_agree(dev):
    """the cfg-4 trainer, same seeds: touched-rows path vs dense path.  The runs differ only in the accumulation order of the
    scatter-adds, which is run-to-run noise of the dense path as well; the first AdamW steps are sign-like (m / sqrt(v) = +-1),
    so an entry whose tiny gradient flips sign moves by ~lr per step.  The noise floor is therefore MEASURED (dense vs dense)
    and the touched-rows run must sit within it.  (The optimiser itself is checked bit for bit above.)"""
    from permuto_sdf_amd.train_step import SyntheticReel, Trainer
    reel = SyntheticReel(dev, nr_images=4, height=60, width=80)

    def run(flag):
        from permuto_sdf_amd.bridge import OccupancyGrid, RaySampler, VolumeRendering
        for cls in (OccupancyGrid, RaySampler, VolumeRendering):          # same jitter streams for every run
            cls._rng = type(cls._rng)()
        tr = Trainer(dev, seed=1, touched_rows=flag)
        tr.nr_rays = 128
        for _ in range(3):
            loss = tr.step(reel)
        return float(loss), tr.sdf.encoding.lattice_values.detach().clone(), tr.rgb.encoding.lattice_values.detach().clone()

    dense_a, dense_b, touched = run(False), run(False), run(True)

    def dist(x, y):
        d = [(a - b).abs() for a, b in zip(x[1:], y[1:])]
        return (abs(x[0] - y[0]) / abs(y[0]), max(float(t.max()) for t in d), max(float((t > 1e-6).float().mean()) for t in d))
    noise, got = dist(dense_a, dense_b), dist(touched, dense_a)
    assert got[0] <= 3 * noise[0] + 1e-3, (got, noise)
    assert got[1] <= 3 * noise[1] + 1e-4, (got, noise)
    assert got[2] <= 3 * noise[2] + 1e-3, (got, noise)


def test_backward_outside_accumulate_is_plain_autograd(dev):
    """the persistent buffer accumulates only inside the owner's `with tr.accumulate():`.  An auxiliary
    torch.autograd.grad(out, positions) or a stray .backward() behaves like plain autograd (dense gradient returned, buffer
    untouched), so nothing can leak into the next optimiser step; a gradient that did land in `.grad` is folded into the buffer
    by FusedAdamW.step instead of being lost."""
    from permuto_sdf_amd.optim import FusedAdamW
    enc = _enc(dev, seed=4)
    tr = enc.enable_touched_rows()
    win = torch.ones(12, device=dev)
    w1 = torch.randn(enc.output_dims(), 3, device=dev)
    pts = ((torch.rand(2000, 3, device=dev) - 0.5) * 0.6).requires_grad_(True)
    out = torch.tanh(enc(pts, win) @ w1).sum()
    (gp,) = torch.autograd.grad(out, pts, retain_graph=True)          # auxiliary: positions only
    assert float(tr.grad.abs().max()) == 0.0 and float(gp.abs().max()) > 0
    out.backward()                                                    # stray backward outside the block
    assert float(tr.grad.abs().max()) == 0.0
    dense = enc.lattice_values.grad.clone()
    assert float(dense.abs().max()) > 0
    # the same gradient through the buffer
    enc.lattice_values.grad = None
    with tr.accumulate():
        torch.tanh(enc(pts.detach(), win) @ w1).sum().backward()
    assert enc.lattice_values.grad is None
    assert float((tr.grad - dense).abs().max()) <= 2e-5 * float(dense.abs().max())
    # step(): a stray .grad is folded in, not lost
    opt = FusedAdamW([enc.lattice_values], lr=1e-3)
    opt.attach(enc.lattice_values, tr)
    enc.lattice_values.grad = dense.clone()
    before = enc.lattice_values.detach().clone()
    g_total = (tr.grad + dense).clone()
    opt.step()
    assert enc.lattice_values.grad is None and float(tr.grad.abs().max()) == 0.0
    moved = (enc.lattice_values.detach() != before)
    has_g = g_total != 0
    assert not bool((moved & ~has_g).any())                           # first step: only entries with a gradient move ...
    assert float((moved & has_g).sum()) >= 0.99 * float(has_g.sum())  # ... and (up to underflowing ones) all of them do


def test_block_adamw_clears_unmarked_gradients_and_rebuilds_active(dev):
    """ADVICE r2, csrc/optim.hip + optim.py: (1) a gradient written into an ACTIVE block whose touched byte is not set (a
    retained graph's backward after the step that consumed its marks) is applied once and cleared, not re-applied on every
    later step; (2) attach() after dense steps and load_state_dict() rebuild `active` from the moments, so that blocks with
    non-zero moments keep moving exactly as the dense update moves them."""
    from permuto_sdf_amd.optim import FusedAdamW
    win = torch.ones(12, device=dev)
    a, b = _enc(dev, seed=6), _enc(dev, seed=6)
    w1 = torch.randn(a.output_dims(), 3, device=dev)
    pts = (torch.rand(1500, 3, device=dev) - 0.5) * 0.3
    opt_a = FusedAdamW([a.lattice_values], lr=1e-3)
    opt_b = FusedAdamW([b.lattice_values], lr=1e-3)
    # two dense steps on both (no touched-rows state yet)
    for _ in range(2):
        a.lattice_values.grad = None
        _loss(a, w1, pts, win).backward()
        b.lattice_values.grad = a.lattice_values.grad.clone()     # the SAME gradient bits (float atomics are order dependent)
        opt_a.step()
        opt_b.step()
    assert torch.equal(a.lattice_values, b.lattice_values)
    b.lattice_values.grad = None                                      # (zero_grad: a stale .grad would be folded in again)
    tr = b.enable_touched_rows()
    opt_b.attach(b.lattice_values, tr)                                # after dense steps: active must come from the moments
    st = opt_b.state[b.lattice_values]
    want = ((st["exp_avg"] != 0) | (st["exp_avg_sq"] != 0)).view(tr.touched.numel(), tr.block_elems).any(1).view_as(tr.active)
    assert torch.equal(tr.active.bool(), want) and bool(want.any())
    # steps WITHOUT new gradients: the dense update keeps moving rows with decaying moments; the block update must follow
    for _ in range(2):
        a.lattice_values.grad = torch.zeros_like(a.lattice_values)
        opt_a.step()
        opt_b.step()
    assert torch.equal(a.lattice_values, b.lattice_values)
    # state_dict round trip into a fresh optimiser / TouchedRows
    c = _enc(dev, seed=6)
    with torch.no_grad():
        c.lattice_values.copy_(b.lattice_values)
    opt_c = FusedAdamW([c.lattice_values], lr=1e-3)
    tr_c = c.enable_touched_rows()
    opt_c.attach(c.lattice_values, tr_c)
    assert int(tr_c.active.sum()) == 0
    opt_c.load_state_dict(opt_b.state_dict())
    assert torch.equal(tr_c.active, tr.active)
    a.lattice_values.grad = torch.zeros_like(a.lattice_values)
    opt_a.step()
    opt_c.step()
    assert torch.equal(a.lattice_values, c.lattice_values)
    # (1) gradient in an active but unmarked block: applied once, then gone
    blk = int(torch.nonzero(tr_c.active.view(-1))[0])
    tr_c.grad.view(-1)[blk * tr_c.block_elems] = 1.0
    assert int(tr_c.touched.sum()) == 0
    opt_c.step()
    assert float(tr_c.grad.abs().max()) == 0.0
