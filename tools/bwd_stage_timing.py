"""Stage stamps of workgroup 0 of mlp_bwd_kernel (measurement build with -DPSDF_BWD_TIMING, see csrc/mlp_bwd.hip):
PSDF_LIB_PATH=permuto_sdf_amd/lib/variants/libpsdf_timing.so python tools/bwd_stage_timing.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permuto_sdf_amd import FusedMLP, _lib as L  # noqa: E402
from permuto_sdf_amd.mlp import _grad_views, mlp_backward_raw, mlp_double_backward  # noqa: E402

dev = torch.device("cuda:0")
dims = [51, 32, 32, 32, 33]
m = FusedMLP(dims).to(dev)
ws, bs = [l.weight for l in m.layers], [l.bias for l in m.layers]
_, dWs, dbs = _grad_views(dims, dev=dev)
out = (ctypes.c_ulonglong * 8)()
for N in (16, 1024, 49152):
    x = torch.randn(dims[0], N, device=dev)
    gy = torch.randn(dims[-1], N, device=dev)
    for dw in (True, False):
        for _ in range(3):
            mlp_backward_raw(dims, x, ws, bs, gy, need_dx=True, need_dw=dw, into=(dWs, dbs) if dw else None)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        mlp_backward_raw(dims, x, ws, bs, gy, need_dx=True, need_dw=dw, into=(dWs, dbs) if dw else None)
        e.record()
        torch.cuda.synchronize()
        L.lib().psdf_debug_bwd_timing(out)
        t = list(out)
        print("N=%6d dW=%d  events %.1f us | stage %.1f  tiles %.1f  zero %.1f  flush-lds %.1f  store %.1f us"
              % (N, dw, s.elapsed_time(e) * 1e3, *[(t[i + 1] - t[i]) / 100.0 for i in range(5)]), flush=True)
    v = torch.randn(dims[0], N, device=dev)
    for _ in range(3):
        mlp_double_backward(dims, x, ws, bs, gy, v, into=(dWs, dbs))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    mlp_double_backward(dims, x, ws, bs, gy, v, into=(dWs, dbs))
    e.record()
    torch.cuda.synchronize()
    L.lib().psdf_debug_bwd_timing(out)
    t = list(out)
    print("N=%6d double backward  events %.1f us | stage %.1f  tiles %.1f  zero %.1f  flush-lds %.1f  store %.1f us"
          % (N, s.elapsed_time(e) * 1e3, *[(t[i + 1] - t[i]) / 100.0 for i in range(5)]), flush=True)
