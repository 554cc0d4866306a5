"""Run-to-run spread of the lattice backward on a ray-like training batch, per level, for the binned and the plain path
(PSDF_ENC_QUEUE_MIN_N): float-atomic order should show up at the 1e-6 level on coarse levels; anything at 1e-3 is a lost or
doubled contribution."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permuto_sdf_amd import PermutoEncoding  # noqa: E402
from permuto_sdf_amd.encoding import encode_backward_raw, encode_double_backward_raw  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = PermutoEncoding(3, 2 ** 18, 24, 2, np.geomspace(1.0, 1e-4, 24), concat_points=True, concat_points_scaling=1e-3).to(dev)
win = torch.zeros(24, device=dev)
win[:8] = 1.0
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
o = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=1) * 1.5
d = torch.nn.functional.normalize((torch.rand(R, 3, device=dev) - 0.5) * 0.6 - o, dim=1)
t = torch.linspace(1.0, 2.0, 96, device=dev)
pts = (o[:, None, :] + t[None, :, None] * d[:, None, :]).reshape(-1, 3).contiguous()
N = pts.shape[0]
lat, sf, sh = enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach()
g = torch.randn(enc.output_dims(), N, device=dev)
u = torch.randn(N, 3, device=dev)
g2 = torch.randn_like(g)


def bwd():
    out = torch.zeros_like(lat)
    encode_backward_raw(enc.cfg, pts, lat, sf, sh, win, g, out, None)
    return out


def dbl():
    out = torch.zeros_like(lat)
    encode_double_backward_raw(enc.cfg, pts, lat, sf, sh, win, u, g, out, None, g2)
    return out


for name, fn in (("backward", bwd), ("merged double backward", dbl)):
    ref = fn().double()
    worst = torch.zeros(24, dtype=torch.float64, device=dev)
    for _ in range(30):
        x = fn().double()
        worst = torch.maximum(worst, (x - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1).clamp_min(1e-30))
    print(name, "N=%d" % N, "worst relative run-to-run difference per level:", " ".join("%.0e" % v for v in worst[:8].tolist()), flush=True)
