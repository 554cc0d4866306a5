"""cfg-2 forward matrix: encode, MLP, fused encode->MLP (with / without the feature by-product) at 2M points."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permuto_sdf_amd import FusedMLP, PermutoEncoding  # noqa: E402
from permuto_sdf_amd.encoding import encode_forward_raw  # noqa: E402
from permuto_sdf_amd.fused import encode_mlp_forward_raw  # noqa: E402
from permuto_sdf_amd.mlp import mlp_forward_raw, pack_params  # noqa: E402


def timeit(fn, n=20):
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
    rnd = torch.randn(N, 3, device=dev)
    ball = 0.5 * torch.nn.functional.normalize(rnd, dim=1) * torch.rand(N, 1, device=dev) ** (1 / 3)
    # ray-ordered: 16384 rays x 128 samples marching through the ball
    o = torch.nn.functional.normalize(torch.randn(16384, 1, 3, device=dev), dim=2) * 0.5
    d = torch.nn.functional.normalize(-o + 0.2 * torch.randn(16384, 1, 3, device=dev), dim=2)
    rays = (o + d * torch.linspace(0, 1, 128, device=dev).view(1, 128, 1)).reshape(-1, 3).contiguous()
    rows = []
    for L_ in (16, 24):
        for net in ([64, 64, 64, 1], [32, 32, 32, 33], [32, 32, 32, 1]):
            enc = PermutoEncoding(3, 2 ** 18, L_, 2, np.geomspace(1.0, 1e-4, L_), concat_points=True,
                                  concat_points_scaling=1e-3, init_scale=1e-2).to(dev)
            mlp = FusedMLP([enc.output_dims()] + net).to(dev)
            packed = pack_params(mlp.dims, [l.weight for l in mlp.layers], [l.bias for l in mlp.layers])
            win = torch.ones(L_, device=dev)
            for name, x in (("ball", ball), ("rays", rays)):
                a = (enc.cfg, x, enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), win)
                feat = encode_forward_raw(*a)
                r = {"L": L_, "net": "-".join(map(str, mlp.dims)), "points": name,
                     "encode_ms": timeit(lambda: encode_forward_raw(*a)),
                     "mlp_ms": timeit(lambda: mlp_forward_raw(mlp.dims, feat, packed)),
                     "fused_ms": timeit(lambda: encode_mlp_forward_raw(*a, mlp.dims, packed)),
                     "fused_feat_ms": timeit(lambda: encode_mlp_forward_raw(*a, mlp.dims, packed, want_feat=True))}
                r = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}
                r["fused_Gsamples_s"] = round(N / r["fused_ms"] / 1e6, 3)
                rows.append(r)
                print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
