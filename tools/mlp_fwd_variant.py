"""Time psdf_mlp_forward from an alternative build of the library (A/B experiments on compile-time switches).
usage: python tools/mlp_fwd_variant.py [path/to/libpsdf_variant.so]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from permuto_sdf_amd import _lib

if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from permuto_sdf_amd import mlp

dims = [36, 64, 64, 64, 1]
N = 1 << 21
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(dims[0], N, device="cuda", generator=g)
ws = [torch.randn(dims[i + 1], dims[i], device="cuda", generator=g) * (2.0 / dims[i]) ** 0.5 for i in range(4)]
bs = [torch.randn(dims[i + 1], device="cuda", generator=g) * 0.1 for i in range(4)]
packed = mlp.pack_params(dims, ws, bs)
y = mlp.mlp_forward_raw(dims, x, packed)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    e0.record()
    for _ in range(20):
        mlp.mlp_forward_raw(dims, x, packed, out=y)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
print("%s: mlp_fwd %.4f ms  checksum %.6f" % (os.path.basename(_lib.LIB_PATH), best, float(y.double().sum())))
