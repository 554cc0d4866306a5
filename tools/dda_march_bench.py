"""Latency of OccupancyGrid.compute_samples_in_occupied_regions for the ray counts of a training step and of a render
chunk, with and without the coarse occupancy mask (csrc/sampling.hip, struct Occ).  GPU only."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permuto_sdf import OccupancyGrid, Sphere  # noqa: E402


def main():
    dev = torch.device("cuda")
    n = 256
    grid = OccupancyGrid(n, 1.0, [0, 0, 0])
    pts = grid.compute_grid_points(False)
    occ = ((pts.norm(dim=1) - 0.3).abs() < 0.02)
    grid.set_grid_occupancy(occ.contiguous())
    sph = Sphere(0.5, [0, 0, 0])
    out = {}
    for R in ([int(a) for a in sys.argv[1:]] or [705, 16384, 262144]):
        g = torch.Generator(device="cpu").manual_seed(1)
        o = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=1) * 1.5
        tgt = (torch.rand(R, 3, generator=g) - 0.5) * 0.6
        d = torch.nn.functional.normalize(tgt - o, dim=1)
        o, d = o.to(dev), d.to(dev)
        _, te, _, tx, _ = sph.ray_intersection(o, d)
        for mask in (True, False, True, False):
            OccupancyGrid.use_coarse_mask = mask
            for _ in range(5):
                rs = grid.compute_samples_in_occupied_regions(o, d, te, tx, 1e-4, 64, True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            K = 100 if R < 100000 else 20
            for _ in range(K):
                rs = grid.compute_samples_in_occupied_regions(o, d, te, tx, 1e-4, 64, True)
            torch.cuda.synchronize()
            out.setdefault("rays_%d" % R, []).append({"mask": mask, "us_per_call": round((time.perf_counter() - t0) / K * 1e6, 1),
                                                       "samples": int(rs.cur_nr_samples)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
