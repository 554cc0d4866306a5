"""Runs only the fused MLP backward of the BASELINE net (36-64-64-64-1, 2M samples) a few times: target for rocprofv3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permuto_sdf_amd import FusedMLP  # noqa: E402
from permuto_sdf_amd.mlp import mlp_backward_raw  # noqa: E402

dev = torch.device("cuda:0")
dims = [int(a) for a in (sys.argv[1].split("-") if len(sys.argv) > 1 else "36-64-64-64-1".split("-"))]
N = int(os.environ.get("PSDF_BENCH_N", 2 ** 21))
torch.manual_seed(0)
m = FusedMLP(dims).to(dev)
x = torch.randn(dims[0], N, device=dev)
gy = torch.ones(dims[-1], N, device=dev)
ws, bs = [l.weight for l in m.layers], [l.bias for l in m.layers]
for _ in range(3):
    mlp_backward_raw(dims, x, ws, bs, gy, need_dx=True)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    mlp_backward_raw(dims, x, ws, bs, gy, need_dx=True)
e.record()
torch.cuda.synchronize()
print("mlp_bwd %s: %.3f ms" % (dims, s.elapsed_time(e) / 10))
s.record()
for _ in range(10):
    mlp_backward_raw(dims, x, ws, bs, gy, need_dx=True, need_dw=False)
e.record()
torch.cuda.synchronize()
print("mlp_bwd %s, data gradient only: %.3f ms" % (dims, s.elapsed_time(e) / 10))
