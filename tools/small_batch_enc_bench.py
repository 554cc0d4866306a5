"""Encoding kernels at the batch sizes of a cfg-4 training step (24 levels, 2^18 rows, ~49 K ray samples): GPU time per call of
the forward (with touched-block marking), the lattice backward, the position backward and the double backward."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permuto_sdf_amd import PermutoEncoding, _lib as L  # noqa: E402
from permuto_sdf_amd.encoding import encode_backward_raw, encode_double_backward_raw, encode_forward_raw  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = PermutoEncoding(3, 2 ** 18, 24, 2, np.geomspace(1.0, 1e-4, 24), concat_points=True, concat_points_scaling=1e-3).to(dev)
tr = enc.enable_touched_rows()
win = torch.ones(24, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for R in (11, 512, 2730):
    o = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=1) * 1.5
    d = torch.nn.functional.normalize((torch.rand(R, 3, device=dev) - 0.5) * 0.6 - o, dim=1)
    t = torch.linspace(1.0, 2.0, 96, device=dev)
    pts = (o[:, None, :] + t[None, :, None] * d[:, None, :]).reshape(-1, 3).contiguous()
    N = pts.shape[0]
    lat, sf, sh = enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach()
    g = torch.randn(enc.output_dims(), N, device=dev)
    gp = torch.zeros_like(pts)
    dd = torch.randn_like(pts)
    gg = torch.empty_like(g)
    r = {"fwd+mark": timed(lambda: encode_forward_raw(enc.cfg, pts, lat, sf, sh, win, touched=tr.touched)),
         "bwd lattice": timed(lambda: encode_backward_raw(enc.cfg, pts, lat, sf, sh, win, g, tr.grad, None)),
         "bwd lattice+pos": timed(lambda: encode_backward_raw(enc.cfg, pts, lat, sf, sh, win, g, tr.grad, gp)),
         "bwd pos": timed(lambda: encode_backward_raw(enc.cfg, pts, lat, sf, sh, win, g, None, gp)),
         "dbl bwd": timed(lambda: encode_double_backward_raw(enc.cfg, pts, lat, sf, sh, win, dd, g, tr.grad, gg))}
    print("N=%7d  " % N + "  ".join("%s %.1f us" % kv for kv in r.items()), flush=True)
