"""Round 5: the deal of the binning kernel's resident round over the levels after K steps of the bench's hot path
(psdf_encode_backward_level_shares), and the encode-backward time with and without it (PSDF_ENC_BWD_BALANCE=0)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from permuto_sdf_amd import _lib as L  # noqa: E402
from permuto_sdf_amd.hotpath import SdfHotPath  # noqa: E402

dev = torch.device("cuda:0")
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 16
hp = SdfHotPath(nr_levels=levels, hidden=64, out_channels=1, device=dev, seed=0)
rs, rgb, aux = bench.make_batch(dev, 7)
normals, gt = aux[4], aux[5]
fn = L.lib().psdf_encode_backward_level_shares
fn.restype = ctypes.c_int
buf = (ctypes.c_int * 64)()
for it in range(24):
    hp.step(rs, rgb, normals, gt)
    torch.cuda.synchronize()
    if it in (0, 1, 2, 4, 8, 16, 23):
        n = fn(buf, 64)
        print("step %2d shares %s (sum %d)" % (it, list(buf[:n]), sum(buf[:n])))
