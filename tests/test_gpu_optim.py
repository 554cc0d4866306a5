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
