"""GPU: the full training step of BASELINE config 4 (permuto_sdf_amd/train_step.py) runs end to end -- sampling,
importance sampling, SDF with analytic gradient, colour and background networks, eikonal (double backward through the
encoding) and curvature losses, fused AdamW, occupancy refresh -- and learns a constant-colour reel."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_training_step_learns_constant_colour(dev):
    from permuto_sdf_amd.train_step import HyperParams, SyntheticReel, Trainer
    hp = HyperParams()
    hp.nr_rays = 256
    hp.target_nr_of_samples = 256 * 96
    tr = Trainer(dev, hp)
    reel = SyntheticReel(dev, nr_images=4, height=60, width=80)
    reel.rgb_reel[:] = torch.tensor([0.8, 0.3, 0.1], device=dev).view(1, 3, 1, 1)
    before = [p.detach().clone() for p in tr.params]
    losses = [float(tr.step(reel)) for _ in range(60)]
    assert all(l == l and abs(l) < 1e3 for l in losses), losses          # finite
    assert tr.last["nr_fg_samples"] > 0 and tr.last["nr_rays"] >= 64
    assert sum(losses[-10:]) / 10 < 0.6 * sum(losses[:5]) / 5, (losses[:5], losses[-10:])
    changed = sum(int((a - b.detach()).abs().max() > 0) for a, b in zip(before, tr.params))
    # all but the forced variance and the 4 inactive Lipschitz bounds (scale clamped at 1 -> zero gradient) moved
    assert changed >= len(before) - 5
    # touched-rows path: the lattice gradient lives in a persistent buffer that the optimiser clears as it consumes it
    t = tr.sdf.encoding.touched_rows
    assert tr.sdf.encoding.lattice_values.grad is None and float(t.grad.abs().max()) == 0.0 and int(t.touched.sum()) == 0
    frac_active = float(t.active.float().mean())
    assert 0.0 < frac_active < 1.0          # some row blocks have been updated, some were never touched (and never read)
    assert torch.isfinite(tr.sdf.encoding.lattice_values).all()


def test_checkpoint_round_trip_on_device(dev, tmp_path):
    from permuto_sdf_amd.train_step import HyperParams, SyntheticReel, Trainer
    hp = HyperParams()
    hp.nr_rays = 128
    hp.target_nr_of_samples = 128 * 96
    tr = Trainer(dev, hp)
    reel = SyntheticReel(dev, nr_images=2, height=40, width=60)
    for _ in range(3):
        tr.step(reel)
    tr.save_checkpoint(str(tmp_path))
    tr2 = Trainer(dev, hp, seed=5)
    tr2.load_checkpoint(str(tmp_path))
    for a, b in zip(tr.params, tr2.params):
        assert torch.equal(a, b)
    assert torch.equal(tr.grid.get_grid_occupancy(), tr2.grid.get_grid_occupancy())
    assert torch.equal(tr.grid.get_grid_values(), tr2.grid.get_grid_values())
