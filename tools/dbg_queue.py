import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from permuto_sdf_amd import PermutoEncoding, _lib as L
from permuto_sdf_amd.encoding import _head, _tail, encode_backward_raw
dev = torch.device("cuda:0")
P, L_, T, N = 3, 16, 2 ** 18, 2 * 1024 * 1024
torch.manual_seed(0)
enc = PermutoEncoding(P, T, L_, 2, np.geomspace(1, 1e-4, L_), concat_points=True, concat_points_scaling=1e-3).to(dev)
pts = torch.rand(N, P, device=dev) - 0.5
g = torch.randn(enc.output_dims(), N, device=dev)
w = torch.ones(L_, device=dev)
cfg = enc.cfg
lat = enc.lattice_values.detach()
def q():
    gl = torch.zeros_like(lat)
    encode_backward_raw(cfg, pts, lat, enc.scale_factor, enc.random_shift_per_level.detach(), w, g, gl, None)
    return gl
def a():
    gl = torch.zeros_like(lat)
    L.call("psdf_encode_backward", *_head(cfg, N), L.ptr(pts), L.ptr(lat), L.ptr(enc.scale_factor),
           L.ptr(enc.random_shift_per_level.detach()), L.ptr(w), *_tail(cfg), L.ptr(g), L.ptr(gl), None, L.stream())
    return gl
gq, ga = q(), a()
torch.cuda.synchronize()
for l in range(L_):
    print(l, "max|a| %.3e  max|q-a| %.3e  sum a %.6e sum q %.6e" % (ga[l].abs().max(), (gq[l] - ga[l]).abs().max(), ga[l].double().sum(), gq[l].double().sum()))
for fn, name in ((q, "queue"), (a, "atomic")):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): fn()
    e.record(); torch.cuda.synchronize()
    print(name, s.elapsed_time(e) / 5, "ms")
