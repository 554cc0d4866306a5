"""Round 4: timings of the encoding kernels that changed (HIP events, mean of 20 launches after 5 warm-ups): forward (the
XCD-pinned launch shape: PSDF_ENC_FWD_XCD=1), position-gradient-only backward (slabs + reduce; PSDF_ENC_POS_ATOMICS=1: the float-
atomic form), lattice backward (run-combine vote: A/B through PSDF_LIB_PATH) on the bench batch (16 384 rays x 128 samples) and at a
training step's size (49 152 samples, 24 levels).  One JSON line; the environment switches are read by the library at start-up."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from permuto_sdf_amd import PermutoEncoding  # noqa: E402
from permuto_sdf_amd.encoding import encode_backward_raw, encode_forward_raw  # noqa: E402


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return round(s.elapsed_time(e) / n, 4)


def main():
    dev = torch.device("cuda:0")
    rs, _, _ = bench.make_batch(dev, 7)
    out = {"env": {k: os.environ.get(k) for k in ("PSDF_ENC_POS_FUSED", "PSDF_ENC_POS_ATOMICS", "PSDF_LIB_PATH", "PSDF_AB_ONLY") if os.environ.get(k)}}
    only = os.environ.get("PSDF_AB_ONLY")      # one configuration (counter runs: a mean over launches of ONE size)
    for L_, pts, tag in ((16, rs.samples_pos, "2M_L16"), (24, rs.samples_pos, "2M_L24"), (24, rs.samples_pos[:49152].contiguous(), "49k_L24")):
        if only and tag != only:
            continue
        torch.manual_seed(0)
        enc = PermutoEncoding(3, 2 ** 18, L_, 2, np.geomspace(1.0, 1e-4, L_), concat_points=True, concat_points_scaling=1e-3,
                              init_scale=1e-2).to(dev)
        win = torch.ones(L_, device=dev)
        a = (enc.cfg, pts, enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), win)
        feat = encode_forward_raw(*a)
        g = torch.randn_like(feat)
        g_pos = torch.zeros_like(pts)
        g_lat = torch.zeros_like(enc.lattice_values)
        out[tag] = {"fwd_ms": timeit(lambda: encode_forward_raw(*a)),
                    "bwd_pos_ms": timeit(lambda: encode_backward_raw(*a, g, None, g_pos)),
                    "bwd_lattice_ms": timeit(lambda: encode_backward_raw(*a, g, g_lat, None)),
                    "bwd_lattice_and_pos_ms": timeit(lambda: encode_backward_raw(*a, g, g_lat, g_pos))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
