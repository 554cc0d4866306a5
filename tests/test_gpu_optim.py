"""GPU: FusedAdamW (csrc/optim.hip, single-tensor and multi-tensor launches) == torch.optim.AdamW, the optimiser the
reference uses (permuto_sdf_py/train_permuto_sdf.py:293-304: betas (0.9, 0.99), eps 1e-15)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wd", [0.0, 0.1])
def test_fused_adamw_matches_torch(dev, wd):
    from permuto_sdf_amd.optim import FusedAdamW
    torch.manual_seed(0)
    shapes = [(3,), (64, 36), (64,), (1, 64), (1,), (300007,), (17, 5), (4, 4)]     # small (batched) and large tensors
    ref = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    ours = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    o_ref = torch.optim.AdamW(ref, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=wd)
    o_ours = FusedAdamW(ours, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=wd)
    for it in range(5):
        for a, b in zip(ref, ours):
            g = torch.randn_like(a)
            a.grad, b.grad = g.clone(), g.clone()
        if it == 2:                       # a parameter without gradient this step keeps its own step count
            ref[3].grad, ours[3].grad = None, None
        o_ref.step()
        o_ours.step()
    for a, b in zip(ref, ours):
        assert (a - b).abs().max() <= 2e-6 * max(1.0, float(a.abs().max()))
    # grad_scale = 1/world: equals scaling the gradients first
    p1, p2 = torch.nn.Parameter(torch.ones(100, device=dev)), torch.nn.Parameter(torch.ones(100, device=dev))
    g = torch.randn(100, device=dev)
    p1.grad, p2.grad = g * 0.25, g.clone()
    FusedAdamW([p1], lr=1e-2).step()
    FusedAdamW([p2], lr=1e-2).step(grad_scale=0.25)
    assert torch.allclose(p1, p2, atol=1e-7)


def test_compat_apex_fused_adam_matches_torch_adamw_on_the_references_groups(dev):
    """compat/apex: `apex.optimizers.FusedAdam` as train_permuto_sdf.py:293-301 builds it (named groups with their own weight
    decay, a group whose parameters have no gradient yet, weight decay switched on later, learning rates changed by a scheduler)
    == torch.optim.AdamW, the reference's other branch (:303)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "compat"))
    import apex
    torch.manual_seed(1)

    def make():
        torch.manual_seed(2)
        return [[torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes] for shapes in
                (((24, 4096, 2), (33, 32), (33,)), ((64, 52), (65,)), ((24, 2048, 2),), ((49, 3), (49, 3)))]
    names, wds = ("model_sdf", "model_bg", "model_rgb_only_encoding", "model_colorcal"), (0.0, 0.0, 0.0, 1e-1)
    ra, rb = make(), make()
    kw = dict(amsgrad=False, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.0, lr=1e-3)
    oa = torch.optim.AdamW([{"params": p, "weight_decay": w, "lr": 1e-3, "name": n} for p, w, n in zip(ra, wds, names)], **kw)
    ob = apex.optimizers.FusedAdam([{"params": p, "weight_decay": w, "lr": 1e-3, "name": n} for p, w, n in zip(rb, wds, names)], **kw)
    for it in range(6):
        oa.zero_grad()
        ob.zero_grad()
        for gi, (ga, gb) in enumerate(zip(ra, rb)):
            if gi == 1 and it < 2:            # the background net gets no gradient during the sphere-initialisation phase
                continue
            for a, b in zip(ga, gb):
                g = torch.randn_like(a) * (1e-3 if a.dim() == 3 else 1.0)
                a.grad, b.grad = g.clone(), g.clone()
        for o in (oa, ob):
            for group in o.param_groups:
                group["lr"] = 1e-3 * (it + 1) / 6                       # a warm-up scheduler at work
                if it >= 3 and group["name"] == "model_rgb_only_encoding":
                    group["weight_decay"] = 1.0                         # train_permuto_sdf.py:400-403
        oa.step()
        ob.step()
    for ga, gb in zip(ra, rb):
        for a, b in zip(ga, gb):
            assert (a - b).abs().max() <= 2e-6 * max(1.0, float(a.abs().max())), (tuple(a.shape), float((a - b).abs().max()))


def test_sharded_step_after_a_resume_clears_the_moments_it_does_not_own(dev):
    """load_state_dict() of a consolidated checkpoint leaves FULL moments on every rank; the first step(owned=...)
    must zero what lies outside the owned ranges (parallel.consolidated_state_dict sums over the ranks), and must not touch the
    parameter there."""
    from permuto_sdf_amd.optim import FusedAdamW
    torch.manual_seed(1)
    n = 1 << 17
    p = torch.nn.Parameter(torch.randn(n, device=dev))
    opt = FusedAdamW([p], lr=1e-2)
    p.grad = torch.randn(n, device=dev)
    opt.step()                                   # full moments everywhere (a replicated step = what a checkpoint holds)
    sd = opt.state_dict()
    q = torch.nn.Parameter(p.detach().clone())
    opt2 = FusedAdamW([q], lr=1e-2)
    opt2.load_state_dict(sd)
    before = q.detach().clone()
    q.grad = torch.randn(n, device=dev)
    lo, hi = n // 4, n // 2
    opt2.step(owned={q: [(lo, hi)]})
    st = opt2.state[q]
    for k in ("exp_avg", "exp_avg_sq"):
        assert float(st[k][:lo].abs().max()) == 0.0 and float(st[k][hi:].abs().max()) == 0.0
        assert float(st[k][lo:hi].abs().max()) > 0.0
    assert torch.equal(q.detach()[:lo], before[:lo]) and torch.equal(q.detach()[hi:], before[hi:])
    assert not torch.equal(q.detach()[lo:hi], before[lo:hi])
    assert opt2._sharded_ranges[q] == [(lo, hi)]
    # a second sharded step leaves the zeros alone and keeps updating the owned range
    q.grad = torch.randn(n, device=dev)
    opt2.step(owned={q: [(lo, hi)]})
    assert float(st["exp_avg"][:lo].abs().max()) == 0.0


def test_alignment_cache_of_small_parameters_sees_moved_moments(dev):
    """FusedAdamW batches small 16-byte aligned parameters into one launch with vector loads and remembers the alignment check per
    parameter (round 6: the training step is host bound).  Replaced moment tensors must be re-examined: aligned ones keep matching
    torch.optim.AdamW; MISALIGNED ones (a view into a flat buffer) must not reach the vector kernel -- the scalar launch refuses
    them loudly (status -1), as it always did."""
    import pytest
    from permuto_sdf_amd import _lib as L
    from permuto_sdf_amd.optim import FusedAdamW
    torch.manual_seed(3)
    a = torch.nn.Parameter(torch.randn(1000, device=dev))
    b = torch.nn.Parameter(a.detach().clone())
    ours, ref = FusedAdamW([a], lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.0), \
        torch.optim.AdamW([b], lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.0)
    for it in range(5):
        g = torch.randn(1000, device=dev)
        a.grad, b.grad = g.clone(), g.clone()
        if it == 2:      # new (aligned) moment tensors, as load_state_dict leaves them
            st = ours.state[a]
            st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()
        ours.step()
        ref.step()
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), it
    st = ours.state[a]
    flat = torch.zeros(1008, device=dev)
    view = flat[1:1001]
    assert view.data_ptr() % 16 != 0
    view.copy_(st["exp_avg"])
    st["exp_avg"] = view
    a.grad = torch.randn(1000, device=dev)
    with pytest.raises(L.PsdfError):
        ours.step()
