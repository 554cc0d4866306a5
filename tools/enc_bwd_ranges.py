"""Lattice backward of the bench batch as ONE launch pair vs k launch pairs over level ranges (each range: binning kernel +
reduce kernel right behind it, so that a range's queues are still in the 256-MB MALL when they are read back)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from permuto_sdf_amd import PermutoEncoding
from permuto_sdf_amd.encoding import _Cfg, encode_backward_raw
dev = torch.device("cuda:0")
rs, rgb, _ = bench.make_batch(dev, 7)
pts = rs.samples_pos
N = pts.shape[0]
L_ = 16
torch.manual_seed(0)
enc = PermutoEncoding(3, 2 ** 18, L_, 2, np.geomspace(1.0, 1e-4, L_), concat_points=True, concat_points_scaling=1e-3).to(dev)
g = torch.randn(enc.output_dims(), N, device=dev)
w = torch.ones(L_, device=dev)
lat = enc.lattice_values.detach()
sf, sh = enc.scale_factor, enc.random_shift_per_level.detach()
c = enc.cfg
def run(k):
    gl = torch.zeros_like(lat)
    bounds = k if isinstance(k, list) else [round(i * L_ / k) for i in range(k + 1)]
    for l0, l1 in zip(bounds[:-1], bounds[1:]):
        sub = _Cfg(3, c.capacity, l1 - l0, 2, False, 1.0)
        encode_backward_raw(sub, pts, lat[l0:l1], sf[l0:l1], sh[l0:l1], w[l0:l1], g[2 * l0:2 * l1], gl[l0:l1], None)
    return gl
ref = run(1)
for k in (1, 2, [0, 9, 16], [0, 11, 16], [0, 12, 16], [0, 6, 16], 4, 8, 16):
    out = run(k)
    err = float((out - ref).abs().max() / ref.abs().max())
    for _ in range(2): run(k)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): run(k)
    e.record(); torch.cuda.synchronize()
    print("%s level ranges: %.3f ms (incl. the 33.5 MB zero fill)   rel diff to 1 range %.1e" % (k, s.elapsed_time(e) / 5, err), flush=True)
