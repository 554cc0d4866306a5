"""BASELINE config 5: sphere-traced inference render, 1920x1080 rays, 15 iterations, hipGraph-captured.
Weights (no checkpoints offline): SPHERE-INITIALISED, as SURVEY 8(d) cfg 5 asks ("trained or sphere-init weights") -- the SDF
net is fitted to |p| - 0.3 with the reference's sphere-fit loss (permuto_sdf_utils.py:53-77, sdf term) for a few hundred
iterations at start-up, and the occupancy grid is the band |sdf| < 0.02 of THAT field (what update_with_sdf builds from a
trained model).  The trace keeps one slot per ray, but converged rays are masked out of the SDF evaluations and the marches
depend on the field, so the weights matter: a RANDOM-weight field (round 1's workload: every ray wanders out of the shell
and marches across the empty interior every iteration) is the worst case and is reported beside it.
Prints one JSON line: frames/s (graph replay) and eager ms for both."""
import json
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permuto_sdf import OccupancyGrid, Sphere  # noqa: E402
from permuto_sdf_amd import FusedMLP, PermutoEncoding  # noqa: E402
from permuto_sdf_amd.sphere_trace import SphereTracer  # noqa: E402


def sphere_fit(enc, mlp, dev, radius=0.3, iters=500, n=30000):
    """the reference's sphere initialisation, SDF term (train_permuto_sdf.py:327-328, permuto_sdf_utils.py:53-77)"""
    params = list(enc.parameters()) + list(mlp.parameters())
    opt = torch.optim.Adam(params, lr=1e-3)
    win = torch.ones(enc.nr_levels, device=dev)
    ball = Sphere(0.5, [0, 0, 0])
    loss = None
    for _ in range(iters):
        p = ball.rand_points_inside(n)
        y = mlp(enc(p, win))
        loss = (y[:, 0:1] - (p.norm(dim=1, keepdim=True) - radius)).abs().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    return float(loss)


def run(tr, o, d, kw, frames):
    for _ in range(3):
        tr.trace(o, d, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.trace(o, d, **kw)
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) * 100
    out = tr.capture(o, d, **kw)
    for _ in range(3):
        tr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        tr.replay()
    torch.cuda.synchronize()
    graph_ms = (time.perf_counter() - t0) * 1000 / frames
    return eager_ms, graph_ms, out


def main(W=1920, H=1080, levels=24, iters=15, frames=50):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    grid = OccupancyGrid(256, 1.0, [0, 0, 0])
    centres = grid.compute_grid_points(False)
    sphere = Sphere(0.5, [0, 0, 0])
    # pinhole camera at z=-1.2 looking at the origin
    ys, xs = torch.meshgrid(torch.linspace(-0.28, 0.28, H, device=dev), torch.linspace(-0.5, 0.5, W, device=dev), indexing="ij")
    d = torch.nn.functional.normalize(torch.stack([xs, ys, torch.ones_like(xs)], -1).view(-1, 3), dim=1)
    o = torch.tensor([0.0, 0.0, -1.2], device=dev).expand_as(d).contiguous()
    kw = dict(nr_sphere_traces=iters, sdf_multiplier=0.9, sdf_converged_tresh=2e-4, return_gradients=True)
    res = {"workload": "sphere_trace_%dx%d_%dit_L%d" % (W, H, iters, levels), "rays": W * H}
    for name in [w for w in ("sphere_init", "random") if w in os.environ.get("PSDF_TRACE_WEIGHTS", "sphere_init,random")]:
        enc = PermutoEncoding(3, 2 ** 18, levels, 2, np.geomspace(1.0, 1e-4, levels), concat_points=True,
                              concat_points_scaling=1e-3, init_scale=1e-2).to(dev)
        # reference SDF net (models.py:153-161); the random-weight case keeps round 1's initialisation (torch default)
        mlp = FusedMLP([enc.output_dims(), 32, 32, 32, 33], reference_init=(name == "sphere_init")).to(dev)
        if name == "sphere_init":
            res["sphere_fit_l1"] = round(sphere_fit(enc, mlp, dev), 5)
            with torch.no_grad():   # the band of the fitted field, Morton order = the order of the grid's own centres
                win = torch.ones(levels, device=dev)
                sdf = torch.cat([mlp(enc(centres[i:i + (1 << 21)], win))[:, 0] for i in range(0, centres.shape[0], 1 << 21)])
            grid.set_grid_occupancy((sdf.abs() < 0.02).contiguous())
        else:
            # round 1's workload: random weights, shell |r - 0.3| < 0.05
            grid.set_grid_occupancy(((centres.norm(dim=1) - 0.3).abs() < 0.05))
        tr = SphereTracer(enc, mlp, grid, sphere)
        tr.compact_marches = os.environ.get("PSDF_TRACE_COMPACT", "1") == "1"
        tr.coarse_mask_for_marches = os.environ.get("PSDF_TRACE_COARSE", "1") == "1"
        eager_ms, graph_ms, out = run(tr, o, d, kw, frames)
        conv = out[3]
        key = "" if name == "sphere_init" else "_random_weights"
        res["eager_ms_per_frame" + key] = round(eager_ms, 3)
        res["graph_ms_per_frame" + key] = round(graph_ms, 3)
        res["fps_graph" + key] = round(1000 / graph_ms, 2)
        res["converged_frac" + key] = round(float(conv.float().mean()), 4)
        res["occupied_frac" + key] = round(float(grid.get_grid_occupancy().float().mean()), 4)
    if "graph_ms_per_frame" in res:
        res["sdf_evals_per_s_upper"] = round(W * H * (iters + 1) / res["graph_ms_per_frame"] * 1000, 0)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
