"""Per-level cost of the lattice backward on the bench batch (ray-ordered samples): one single-level encoding per scale."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from permuto_sdf_amd import PermutoEncoding
from permuto_sdf_amd.encoding import encode_backward_raw
dev = torch.device("cuda:0")
rs, rgb, _ = bench.make_batch(dev, 7)
pts = rs.samples_pos
N = pts.shape[0]
scales = np.geomspace(1.0, 1e-4, 16)
torch.manual_seed(0)
tot = 0
for l, sc in enumerate(scales):
    enc = PermutoEncoding(3, 2 ** 18, 1, 2, [sc], concat_points=False).to(dev)
    g = torch.randn(2, N, device=dev)
    w = torch.ones(1, device=dev)
    lat = enc.lattice_values.detach()
    def run():
        gl = torch.zeros_like(lat)
        encode_backward_raw(enc.cfg, pts, lat, enc.scale_factor, enc.random_shift_per_level.detach(), w, g, gl, None)
        return gl
    gl = run()
    nz = int((gl[0].abs().sum(1) > 0).sum())
    for _ in range(2): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    tot += ms
    print("level %2d scale %.2e: %.3f ms, touched rows %d" % (l, sc, ms, nz), flush=True)
print("sum %.3f ms" % tot)
