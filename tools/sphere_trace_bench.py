"""BASELINE config 5: sphere-traced inference render, 1920x1080 rays, 15 iterations, hipGraph-captured.
Random-init lattice + an SDF head biased to a sphere-ish field (no checkpoints offline): timing does not depend on the
weights because the trace is fixed-shape (every ray evaluates the SDF every iteration).
Prints one JSON line: frames/s (graph replay), eager ms, and the per-frame kernel count."""
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


def main(W=1920, H=1080, levels=24, iters=15, frames=50):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    enc = PermutoEncoding(3, 2 ** 18, levels, 2, np.geomspace(1.0, 1e-4, levels), concat_points=True,
                          concat_points_scaling=1e-3, init_scale=1e-2).to(dev)
    mlp = FusedMLP([enc.output_dims(), 32, 32, 32, 33]).to(dev)      # reference SDF net (models.py:153-161)
    grid = OccupancyGrid(256, 1.0, [0, 0, 0])
    g = torch.linspace(-0.5, 0.5, 256, device=dev)
    # occupied shell around radius 0.3 (Morton order is what set_grid_occupancy expects: use the grid's own centres)
    c = grid.compute_grid_points(False)
    grid.set_grid_occupancy(((c.norm(dim=1) - 0.3).abs() < 0.05))
    sphere = Sphere(0.5, [0, 0, 0])
    # pinhole camera at z=-1.2 looking at the origin
    ys, xs = torch.meshgrid(torch.linspace(-0.28, 0.28, H, device=dev), torch.linspace(-0.5, 0.5, W, device=dev), indexing="ij")
    d = torch.nn.functional.normalize(torch.stack([xs, ys, torch.ones_like(xs)], -1).view(-1, 3), dim=1)
    o = torch.tensor([0.0, 0.0, -1.2], device=dev).expand_as(d).contiguous()
    tr = SphereTracer(enc, mlp, grid, sphere)
    kw = dict(nr_sphere_traces=iters, sdf_multiplier=0.9, sdf_converged_tresh=2e-4, return_gradients=True)
    for _ in range(3):
        tr.trace(o, d, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.trace(o, d, **kw)
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) * 100
    tr.capture(o, d, **kw)
    for _ in range(3):
        tr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        tr.replay()
    torch.cuda.synchronize()
    graph_ms = (time.perf_counter() - t0) * 1000 / frames
    print(json.dumps({"workload": "sphere_trace_%dx%d_%dit_L%d" % (W, H, iters, levels), "rays": W * H,
                      "eager_ms_per_frame": round(eager_ms, 3), "graph_ms_per_frame": round(graph_ms, 3),
                      "fps_graph": round(1000 / graph_ms, 2),
                      "sdf_evals_per_s": round(W * H * (iters + 1) / graph_ms * 1000, 0)}))


if __name__ == "__main__":
    main()
