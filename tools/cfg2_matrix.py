"""BASELINE config 2 matrix: N = 2 097 152 points, L in {16, 24}, T = 2^18, F = 2; the BASELINE 64x3 -> 1 net
and the reference's 32x3 -> 33 net; forward, forward + backward(lattice), forward + backward(lattice + positions).
Points: uniform in the radius-0.5 ball (the config's definition) and ray-ordered samples (what the renderer feeds).
One JSON line per row; `python tools/cfg2_matrix.py > profiles/...`."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permuto_sdf_amd import FusedMLP, PermutoEncoding  # noqa: E402
from permuto_sdf_amd.encoding import encode_backward_raw, encode_forward_raw, morton_order  # noqa: E402
from permuto_sdf_amd.mlp import mlp_backward_raw, mlp_forward_raw, pack_params  # noqa: E402


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    dev = torch.device("cuda:0")
    N = 2 ** 21
    torch.manual_seed(0)
    ball = 0.5 * torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=1) * torch.rand(N, 1, device=dev) ** (1 / 3)
    ball_sorted = ball[morton_order(ball)].contiguous()
    o = torch.nn.functional.normalize(torch.randn(16384, 1, 3, device=dev), dim=2) * 0.5
    d = torch.nn.functional.normalize(-o + 0.2 * torch.randn(16384, 1, 3, device=dev), dim=2)
    rays = (o + d * torch.linspace(0, 1, 128, device=dev).view(1, 128, 1)).reshape(-1, 3).contiguous()
    for L_ in (16, 24):
        for net in ([64, 64, 64, 1], [32, 32, 32, 33]):
            enc = PermutoEncoding(3, 2 ** 18, L_, 2, np.geomspace(1.0, 1e-4, L_), concat_points=True,
                                  concat_points_scaling=1e-3, init_scale=1e-2).to(dev)
            mlp = FusedMLP([enc.output_dims()] + net).to(dev)
            ws, bs = [l.weight for l in mlp.layers], [l.bias for l in mlp.layers]
            win = torch.ones(L_, device=dev)
            lat = enc.lattice_values.detach()
            for name, x in (("ball", ball), ("ball, Morton-sorted by the caller", ball_sorted), ("rays", rays)):
                a = (enc.cfg, x, lat, enc.scale_factor, enc.random_shift_per_level.detach(), win)
                gy = torch.ones(net[-1], N, device=dev)

                def fwd():
                    feat = encode_forward_raw(*a)
                    return feat, mlp_forward_raw(mlp.dims, feat, pack_params(mlp.dims, ws, bs))

                def fwd_bwd(pos):
                    feat, _ = fwd()
                    d_feat, _, _ = mlp_backward_raw(mlp.dims, feat, ws, bs, gy, need_dx=True)
                    g_lat = torch.zeros_like(lat)
                    g_pos = torch.zeros_like(x) if pos else None
                    encode_backward_raw(*a, d_feat, g_lat, g_pos)

                t_f, t_fb, t_fbp = timeit(fwd), timeit(lambda: fwd_bwd(False)), timeit(lambda: fwd_bwd(True))
                print(json.dumps({"cfg": 2, "N": N, "L": L_, "net": "-".join(map(str, mlp.dims)), "points": name,
                                  "fwd_ms": round(t_f, 3), "fwd_bwd_lattice_ms": round(t_fb, 3),
                                  "fwd_bwd_lattice_pos_ms": round(t_fbp, 3),
                                  "fwd_Gsamples_s": round(N / t_f / 1e6, 3),
                                  "fwd_bwd_lattice_Gsamples_s": round(N / t_fb / 1e6, 3)}), flush=True)


if __name__ == "__main__":
    main()
