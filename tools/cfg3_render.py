"""BASELINE config 3: full volume render of a 512x512 image through the drop-in API --
Sphere.ray_intersection -> OccupancyGrid.compute_samples_in_occupied_regions (256^3 grid, shell occupancy, <= 128
samples/ray, no jitter) -> SDF network -> NeuS weights -> integrate.  Chunked at 16 384 rays exactly like the
reference's run_net_in_chunks (train_permuto_sdf.py:172-187) because of its 2 097 152-sample pool
(src/OccupancyGrid.cu:216), and unchunked (one pool for the whole image) as the native path allows."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permuto_sdf import OccupancyGrid, Sphere, VolumeRendering  # noqa: E402
from permuto_sdf_amd import FusedMLP, PermutoEncoding  # noqa: E402
from permuto_sdf_amd.encoding import encode_forward_raw  # noqa: E402
from permuto_sdf_amd.mlp import mlp_forward_raw, pack_params  # noqa: E402


def main(W=512, H=512, levels=24):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    enc = PermutoEncoding(3, 2 ** 18, levels, 2, np.geomspace(1.0, 1e-4, levels), concat_points=True,
                          concat_points_scaling=1e-3, init_scale=1e-2).to(dev)
    mlp = FusedMLP([enc.output_dims(), 64, 64, 64, 1]).to(dev)
    with torch.no_grad():
        mlp.layers[-1].bias.fill_(0.05)
    # the hot path's forward arithmetic (hotpath.py): two fp16 pieces per operand for the BASELINE net unless PSDF_MLP_FWD_SPLIT=bf16
    from permuto_sdf_amd.mlp import f16_forward_supported
    f16 = os.environ.get("PSDF_MLP_FWD_SPLIT", "f16") != "bf16" and f16_forward_supported(mlp.dims)
    packed = pack_params(mlp.dims, [l.weight for l in mlp.layers], [l.bias for l in mlp.layers], f16=f16)
    win = torch.ones(levels, device=dev)
    grid = OccupancyGrid(256, 1.0, [0, 0, 0])
    c = grid.compute_grid_points(False)
    grid.set_grid_occupancy(((c.norm(dim=1) - 0.3).abs() < 0.02))
    sphere = Sphere(0.5, [0, 0, 0])
    ys, xs = torch.meshgrid(torch.arange(H, device=dev) + 0.5, torch.arange(W, device=dev) + 0.5, indexing="ij")
    d = torch.nn.functional.normalize(torch.stack([(xs - W / 2) / 512.0, (ys - H / 2) / 512.0, torch.ones_like(xs)], -1).view(-1, 3), dim=1)
    o = torch.tensor([0.0, 0.0, -1.5], device=dev).expand_as(d).contiguous()
    rgb_const = torch.rand(1, 3, device=dev)

    def render(o, d):
        _, te, _, tx, _ = sphere.ray_intersection(o, d)
        rs = grid.compute_samples_in_occupied_regions(o, d, te, tx, 1e-4, 128, False).compact_to_valid_samples()
        M = rs.samples_pos.shape[0]
        feat = encode_forward_raw(enc.cfg, rs.samples_pos, enc.lattice_values.detach(), enc.scale_factor,
                                  enc.random_shift_per_level.detach(), win)
        sdf = mlp_forward_raw(mlp.dims, feat, packed, f16=f16)
        alpha = VolumeRendering.sdf2alpha(rs, sdf.view(-1, 1), 512.0, True, 1.0)
        T, _ = VolumeRendering.cumprod_alpha2transmittance(rs, 1.0 - alpha + 1e-7)
        w = alpha * T
        rgb = rgb_const.expand(M, 3).contiguous()
        return VolumeRendering.integrate_with_weights(rs, rgb, w), M

    def image(chunk):
        tot = 0
        outs = []
        for i in range(0, o.shape[0], chunk):
            img, M = render(o[i:i + chunk], d[i:i + chunk])
            outs.append(img)
            tot += M
        return torch.cat(outs), tot

    res = {}
    for name, chunk in (("chunked_16384_rays", 16384), ("one_pool", W * H)):
        grid.max_nr_samples = OccupancyGrid.POOL if chunk == 16384 else W * H * 128   # reference pool / whole image
        for _ in range(2):
            img, tot = image(chunk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            img, tot = image(chunk)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 200
        res[name] = {"ms_per_image": round(ms, 3), "samples": tot, "Msamples_per_s": round(tot / ms / 1e3, 1),
                     "Mrays_per_s": round(W * H / ms / 1e3, 2)}
    print(json.dumps({"cfg": 3, "image": "%dx%d" % (W, H), "L": levels, "net": "-".join(map(str, mlp.dims)),
                      "mlp_forward": "two fp16 pieces" if f16 else "three bf16 pieces", **res}))


if __name__ == "__main__":
    main()
