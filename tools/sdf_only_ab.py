"""SDF-only queries of the importance-sampling rounds (train_step.SdfNet.sdf_only): the fused encode -> MLP launch against the
unfused pair, GPU time per call at a training step's sizes: python tools/sdf_only_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permuto_sdf_amd.encoding import encode_forward_raw  # noqa: E402
from permuto_sdf_amd.fused import encode_mlp_forward_raw  # noqa: E402
from permuto_sdf_amd.mlp import mlp_forward_raw, pack_params  # noqa: E402
from permuto_sdf_amd.train_step import HyperParams, SdfNet  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = SdfNet(HyperParams()).to(dev)
lin = list(net.mlp_sdf.layers)
dims = [lin[0].in_features, 32, 32, 32, 1]
ws, bs = [l.weight for l in lin], [l.bias for l in lin]
ws[-1], bs[-1] = ws[-1][0:1].contiguous(), bs[-1][0:1].contiguous()
packed = pack_params(dims, ws, bs)
e = net.encoding


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    t.record()
    torch.cuda.synchronize()
    return s.elapsed_time(t) / reps * 1e3


for it in (0, 20000):
    win = net.window(it).contiguous()
    for N in (10960, 32576, 49152, 131072, 524288):
        pts = (torch.rand(N, 3, device=dev) - 0.5) * 0.9
        lat, sf, sh = e.lattice_values.detach(), e.scale_factor, e.random_shift_per_level.detach()
        a = timed(lambda: encode_mlp_forward_raw(e.cfg, pts, lat, sf, sh, win, dims, packed))
        b = timed(lambda: mlp_forward_raw(dims, encode_forward_raw(e.cfg, pts, lat, sf, sh, win), packed))
        y1, _ = encode_mlp_forward_raw(e.cfg, pts, lat, sf, sh, win, dims, packed)
        y2 = mlp_forward_raw(dims, encode_forward_raw(e.cfg, pts, lat, sf, sh, win), packed)
        print("it %5d N=%7d  fused %.1f us   encode + mlp %.1f us   max |diff| %.2e" % (it, N, a, b, float((y1.view(-1) - y2.view(-1)).abs().max())), flush=True)
