"""GPU: the fixed-shape sphere tracer (permuto_sdf_amd/sphere_trace.py) against a line-by-line restatement, in this
test, of the reference's mask-compacting loop (permuto_sdf_py/utils/sdf_utils.py:120-218) built on the drop-in API --
same kernels per point, so final points must be bit-identical; plus graph replay and a convergence check on an SDF
network fitted to a sphere."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import scene

pytestmark = pytest.mark.gpu


def fit_sphere_sdf(dev, r0=0.3, iters=300):
    """small encoded SDF fitted to |x| - r0 inside the unit cube (enough for sphere tracing to converge)."""
    from permuto_sdf_amd import FusedMLP, PermutoEncoding
    from permuto_sdf_amd.optim import FusedAdamW
    torch.manual_seed(0)
    enc = PermutoEncoding(3, 2 ** 16, 8, 2, np.geomspace(1.0, 0.02, 8), concat_points=True, concat_points_scaling=1.0,
                          init_scale=1e-3).to(dev)
    mlp = FusedMLP([enc.output_dims(), 64, 64, 64, 1]).to(dev)
    opt = FusedAdamW(list(enc.parameters())[:1] + list(mlp.parameters()), lr=5e-3)
    win = torch.ones(8, device=dev)
    for _ in range(iters):
        x = torch.rand(16384, 3, device=dev) - 0.5
        loss = ((mlp(enc(x, win)) - (x.norm(dim=1, keepdim=True) - r0)) ** 2).mean()
        for p in opt.param_groups[0]["params"]:
            p.grad = None
        loss.backward()
        opt.step()
    return enc, mlp, win, float(loss)


def reference_style_trace(n_iter, o, d, sdf_fn, mult, thr, grid, sphere):
    """restatement of sdf_utils.sphere_trace (occupancy branch) with boolean-mask compaction"""
    _, te, _, tx, _ = sphere.ray_intersection(o, d)
    rs = grid.compute_first_sample_start_of_occupied_regions(o, d, te, tx).compact_to_valid_samples()
    pos, dirs = rs.samples_pos, rs.samples_dirs
    voxel = 1.0 / grid.get_nr_voxels_per_dim()
    pos = pos + dirs * voxel * 0.5
    pts = pos.clone()
    conv = torch.zeros_like(pos)[:, 0:1].bool()
    for _ in range(n_iter):
        sel = torch.logical_not(conv)
        pu = pts[sel.repeat(1, 3)].view(-1, 3)
        du = dirs[sel.repeat(1, 3)].view(-1, 3)
        if pu.shape[0] == 0:
            break
        sdf = sdf_fn(pu)
        pu = pu + du * sdf * mult
        newly = sdf.abs() < thr
        conv[sel] = torch.logical_or(conv[sel], newly.view(-1))
        conv = conv.view(-1, 1)
        pu, within = grid.advance_sample_to_next_occupied_voxel(du.contiguous(), pu.contiguous())
        conv[sel] = torch.logical_or(conv[sel], torch.logical_not(within.view(-1)))
        conv = conv.view(-1, 1)
        pts[sel.repeat(1, 3)] = pu.view(-1)
    return pts, rs


def test_fixed_shape_trace_equals_reference_style_loop(dev):
    from permuto_sdf import OccupancyGrid, Sphere
    from permuto_sdf_amd.sphere_trace import SphereTracer
    enc, mlp, win, loss = fit_sphere_sdf(dev)
    assert loss < 1e-4
    port = O.Oracle("port")
    occ = scene.shell_occupancy(port, 64, r0=0.3, width=0.06, drop=0.0)
    grid = OccupancyGrid(64, 1.0, [0, 0, 0])
    grid.set_grid_occupancy(torch.from_numpy(occ).to(dev))
    sphere = Sphere(0.5, [0, 0, 0])
    on, dn = scene.make_rays(20000, seed=2, jitter_target=0.45)
    o, d = torch.from_numpy(on).to(dev), torch.from_numpy(dn).to(dev)

    def sdf_fn(p):
        with torch.no_grad():
            return mlp(enc(p.contiguous(), win))
    ref_pts, rs = reference_style_trace(15, o, d, sdf_fn, 0.9, 2e-4, grid, sphere)
    tracer = SphereTracer(enc, mlp, grid, sphere, win)
    pts, sdf, grads, conv = tracer.trace(o, d, 15, 0.9, 2e-4, True)
    cnt = (rs.ray_start_end_idx[:, 1] - rs.ray_start_end_idx[:, 0]).bool()      # rays that met an occupied voxel
    assert int(cnt.sum()) == ref_pts.shape[0] > 5000
    assert torch.equal(pts[cnt], ref_pts)                                          # bit-identical end points
    # converged rays that hit the sphere lie on it, and the analytic normal points outwards with unit length
    on_surface = cnt & (sdf.view(-1).abs() < 2e-4)
    assert int(on_surface.sum()) > 3000
    p = pts[on_surface]
    assert float((p.norm(dim=1) - 0.3).abs().max()) < 5e-3
    g = grads[on_surface]
    assert float((g.norm(dim=1) - 1).abs().median()) < 0.1
    assert float((torch.nn.functional.normalize(g, dim=1) * torch.nn.functional.normalize(p, dim=1)).sum(1).median()) > 0.98
    # analytic normal == autograd normal (reference: get_sdf_and_gradient, models.py:236-251)
    q = p[:2000].clone().requires_grad_(True)
    s = mlp(enc(q, win))
    (ga,) = torch.autograd.grad(s, q, torch.ones_like(s))
    assert float((ga - g[:2000]).abs().max()) < 1e-4 * max(1.0, float(ga.abs().max()))
    # hipGraph: capture once, replay on new rays written in place
    out = tracer.capture(o.clone(), d.clone(), nr_sphere_traces=15, sdf_multiplier=0.9, sdf_converged_tresh=2e-4)
    on2, dn2 = scene.make_rays(20000, seed=5, jitter_target=0.45)
    tracer._o.copy_(torch.from_numpy(on2).to(dev))
    tracer._d.copy_(torch.from_numpy(dn2).to(dev))
    pts_g = tracer.replay()[0].clone()
    pts_e = tracer.trace(tracer._o, tracer._d, 15, 0.9, 2e-4, True)[0]
    assert torch.equal(pts_g, pts_e)
