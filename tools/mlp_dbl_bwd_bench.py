"""fused double backward of the MLP (csrc/mlp_bwd.hip) against the torch-autograd (rocBLAS) evaluation of the same VJP"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permuto_sdf_amd import FusedMLP
from permuto_sdf_amd.mlp import mlp_double_backward, _torch_gpu_double_backward
dev = torch.device("cuda:0")
for dims, N in (([52, 32, 32, 32, 33], 49152), ([52, 32, 32, 32, 33], 2 ** 21), ([36, 64, 64, 64, 1], 2 ** 21)):
    torch.manual_seed(0)
    m = FusedMLP(dims).to(dev)
    x = torch.randn(dims[0], N, device=dev); gy = torch.randn(dims[-1], N, device=dev); v = torch.randn(dims[0], N, device=dev)
    ws, bs = [l.weight for l in m.layers], [l.bias for l in m.layers]
    for name, fn in (("fused", mlp_double_backward), ("torch", _torch_gpu_double_backward)):
        if name == "torch" and N > 2 ** 20 and dims[1] == 64:
            pass
        for _ in range(2): fn(dims, x, ws, bs, gy, v)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): fn(dims, x, ws, bs, gy, v)
        e.record(); torch.cuda.synchronize()
        print("%s N=%d %s: %.3f ms" % (dims, N, name, s.elapsed_time(e) / 5), flush=True)
