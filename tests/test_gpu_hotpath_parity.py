"""GPU: ORACLE PARITY OF THE CONFIGURATION THAT IS TIMED.  bench.py times SdfHotPath.step at 2 097 152 samples, where the
library dispatches to `mlp_fwd_split_kernel`, `mlp_bwd_split_kernel` (N >= 2^18, csrc/mlp_bwd.hip) and the queue-mode encode
backward `encode_bwd_kernel<..,queue>` + `encode_bwd_reduce_kernel` (N >= 2^18, csrc/encode.hip queue_plan).  These tests run
that dispatch -- asserted through psdf_last_path(), not assumed -- against the CPU oracle chain (oracle/hotpath_oracle.py:
oracle/permuto_oracle.py + unmodified torch.nn + oracle/neus_oracle.py, torch autograd):

  * the whole step at 2 048 rays x 128 = 2^18 samples, 16 and 24 levels, T = 2^18: sdf, radiance, loss, lattice gradient (global
    and per level) and all eight MLP parameter gradients;
  * SURVEY.md 8(d) cfg 2 literally: the 2 097 152-sample batch of BASELINE.json configs[1] goes through the kernels in one piece
    and a 65 536-sample subset of it is compared with the oracle -- forward values directly; gradients by feeding an upstream
    gradient that is zero outside the subset, so that the full-size backward launches produce exactly the subset's gradient.

Tolerance: 1e-4 relative to the largest entry (the north_star bar for fp32 SDF / radiance), written at each assert.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import hotpath_oracle as ho
from oracle import permuto_oracle as po

pytestmark = pytest.mark.gpu

TOL = 1e-4


def last_path(family):
    from permuto_sdf_amd import _lib as L
    fn = L.lib().psdf_last_path
    fn.restype = ctypes.c_int
    return int(fn(ctypes.c_int(family)))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def mlp_params_cpu(mlp):
    return [l.weight.detach().cpu().clone() for l in mlp.layers], [l.bias.detach().cpu().clone() for l in mlp.layers]


@pytest.mark.parametrize("nr_levels", [16, 24])
def test_step_on_the_benchmarked_kernels_matches_oracle(dev, nr_levels):
    import bench
    from permuto_sdf import VolumeRendering
    from permuto_sdf_amd.hotpath import SdfHotPath
    R, n = 2048, 128
    hp = SdfHotPath(nr_levels=nr_levels, hidden=64, out_channels=1, capacity=2 ** 18, device=dev, seed=5)
    rs, rgb, aux = bench.make_batch(dev, 21, nr_rays=R, per_ray=n)
    normals, gt = aux[4], aux[5]
    N = rs.samples_pos.shape[0]
    assert N == 1 << 18
    pred, saved, out = hp.step(rs, rgb, normals, gt, reduce=False, optimizer_step=False)
    torch.cuda.synchronize()
    # the kernels bench.py times ran -- not their small-batch siblings
    assert last_path(2) in (2, 3), "MLP forward did not run mlp_fwd_split_kernel (2: bf16 pieces, 3: fp16 pieces)"
    assert last_path(1) in (2, 4), "MLP backward did not run a split-operand kernel (2: bf16 pieces, 4: fp16 pieces)"
    assert last_path(0) == 2, "encode backward did not run the queue-mode binning + reduce kernels"

    ws, bs = mlp_params_cpu(hp.mlp)
    ref = ho.reference_step(rs.samples_pos.cpu(), rs.samples_dirs.cpu(), normals.cpu(), rs.samples_dt.cpu(), rgb.cpu(), gt.cpu(),
                            R, n, hp.enc.lattice_values.detach().cpu(), hp.enc.scale_per_level,
                            hp.enc.random_shift_per_level.detach().cpu(), torch.ones(nr_levels), ws, bs, hp.inv_s.cpu(),
                            hp.cos_anneal_ratio, reference_compat=VolumeRendering.reference_compat)
    errs = {"features": rel(saved["feat"].t(), ref["feat"]),
            "sdf": rel(saved["sdf"].view(-1, 1), ref["sdf"]),
            "radiance": rel(pred, ref["pred"]),
            "loss": abs(float(out["loss"]) - ref["loss"]) / abs(ref["loss"]),
            "lattice_grad": rel(out["grads"][0], ref["g_lattice"])}
    for l in range(4):
        errs["dW%d" % l] = rel(out["grads"][1 + 2 * l], ref["g_weights"][l])
        errs["db%d" % l] = rel(out["grads"][2 + 2 * l], ref["g_biases"][l])
    print("L=%d, 2^18 samples, split / queue kernels: " % nr_levels + " ".join("%s %.1e" % kv for kv in errs.items()))
    assert float(ref["g_lattice"].abs().max()) > 0 and all(float(g.abs().max()) > 0 for g in ref["g_weights"])
    assert max(errs.values()) < TOL, errs
    # per level: a coarse level's large entries must not hide a wrong fine level
    g, gr = out["grads"][0].cpu(), ref["g_lattice"]
    for l in range(nr_levels):
        s = float(gr[l].abs().max())
        assert s > 0 and float((g[l] - gr[l]).abs().max()) <= TOL * s, (l, s)


def test_cfg2_full_batch_subset_matches_oracle(dev):
    """SURVEY.md 8(d) cfg 2: N = 2 097 152 points, 16 levels, 36-64-64-64-1 -- 'parity vs oracle on a 65 536-point subset'."""
    import bench
    from permuto_sdf_amd.encoding import encode_backward_raw, encode_forward_raw
    from permuto_sdf_amd.hotpath import SdfHotPath
    from permuto_sdf_amd.mlp import mlp_backward_raw, mlp_forward_raw, pack_params
    L_ = 16
    hp = SdfHotPath(nr_levels=L_, hidden=64, out_channels=1, capacity=2 ** 18, device=dev, seed=9)
    rs, _, _ = bench.make_batch(dev, 33)                         # the bench batch: 16 384 rays x 128
    pos = rs.samples_pos
    N = pos.shape[0]
    assert N == 2097152
    sub = torch.arange(5, N, 32, device=dev)                     # 65 536 samples spread over every ray
    assert sub.numel() == 65536
    cfg, enc, mlp = hp.enc.cfg, hp.enc, hp.mlp
    win = torch.linspace(0.3, 1.0, L_, device=dev)
    ws_d, bs_d = [l.weight for l in mlp.layers], [l.bias for l in mlp.layers]
    # ---- full-size launches
    feat = encode_forward_raw(cfg, pos, enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), win)
    sdf = mlp_forward_raw(mlp.dims, feat, pack_params(mlp.dims, ws_d, bs_d))
    g = torch.Generator().manual_seed(4)
    dy_sub = torch.randn(sub.numel(), generator=g)
    dY = torch.zeros(1, N, device=dev)
    dY[0, sub] = dy_sub.to(dev)
    d_feat, dWs, dbs = mlp_backward_raw(mlp.dims, feat, ws_d, bs_d, dY, need_dx=True)
    g_lat = torch.zeros_like(enc.lattice_values)
    encode_backward_raw(cfg, pos, enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), win,
                        d_feat, g_lat, None)
    torch.cuda.synchronize()
    assert last_path(2) in (2, 3) and last_path(0) == 2 and last_path(1) in (2, 4)
    # ---- oracle on the subset only
    lat = enc.lattice_values.detach().cpu().clone().requires_grad_(True)
    f_ref = po.encode(pos[sub].cpu(), lat, enc.scale_per_level, enc.random_shift_per_level.detach().cpu(), win.cpu(), True, 1e-3)
    ws, bs = mlp_params_cpu(mlp)
    lins = []
    mods = []
    for i, (w, b) in enumerate(zip(ws, bs)):
        lin = torch.nn.Linear(w.shape[1], w.shape[0])
        lin.weight.data.copy_(w)
        lin.bias.data.copy_(b)
        lins.append(lin)
        mods += [lin] + ([torch.nn.GELU()] if i < 3 else [])
    f_leaf = f_ref.detach().clone().requires_grad_(True)
    y_ref = torch.nn.Sequential(*mods)(f_leaf)
    y_ref.backward(dy_sub.view(-1, 1))
    f_ref.backward(f_leaf.grad)
    errs = {"features": rel(feat[:, sub].t(), f_ref), "sdf": rel(sdf[0, sub].view(-1, 1), y_ref),
            "d_features": rel(d_feat[:, sub].t(), f_leaf.grad), "lattice_grad": rel(g_lat, lat.grad)}
    for l in range(4):
        errs["dW%d" % l] = rel(dWs[l], lins[l].weight.grad)
        errs["db%d" % l] = rel(dbs[l], lins[l].bias.grad)
    print("cfg 2, 65 536-sample subset of the 2 097 152 batch: " + " ".join("%s %.1e" % kv for kv in errs.items()))
    assert max(errs.values()) < TOL, errs
    for l in range(L_):
        s = float(lat.grad[l].abs().max())
        assert s > 0 and float((g_lat[l].cpu() - lat.grad[l]).abs().max()) <= TOL * s, (l, s)
    # samples outside the subset received a zero upstream gradient: their data gradient is exactly zero
    mask = torch.ones(N, dtype=torch.bool, device=dev)
    mask[sub] = False
    assert float(d_feat[:, mask].abs().max()) == 0.0


@pytest.mark.parametrize("L_", [16, 24])
@pytest.mark.parametrize("dy_kind", ["normal", "six_decades_and_an_outlier"])
def test_cfg2_full_batch_dense_gradient_against_float64(dev, dy_kind, L_):
    """The error budget of the TIMED arithmetic with every one of the 2 097 152 samples carrying a non-zero upstream gradient
    (the subset test above zeroes dY outside 65 536 samples): features -> split-fp16 MLP forward -> split-fp16 MLP backward
    (dX, dW, db: accumulated over the whole batch) -> encode backward (lattice gradient), against a float64 evaluation of the
    same chain on the GPU (torch float64 autograd through the unmodified Linear/GELU stack in chunks; float64 scatter of the
    float64 feature gradient with the oracle's vertex rows and barycentric weights).  Bar: 5e-5 of the largest entry of each
    gradient -- half the north_star tolerance (1e-4) -- so that the two-piece arithmetic keeps a margin at the benchmark size;
    the measured margins are printed."""
    import bench
    from permuto_sdf_amd.encoding import encode_backward_raw, encode_forward_raw
    from permuto_sdf_amd.hotpath import SdfHotPath
    from permuto_sdf_amd.mlp import mlp_backward_raw, mlp_forward_raw, pack_params
    # L_ = 16: the headline's net (36-64-64-64-1); 24: the reference's level count (52-64-64-64-1, the four-input-tile
    # instantiations of both split-fp16 kernels), which the bench line quotes as an extra row
    F_, T_ = 2, 2 ** 18
    hp = SdfHotPath(nr_levels=L_, hidden=64, out_channels=1, capacity=T_, device=dev, seed=9)
    rs, _, _ = bench.make_batch(dev, 34)
    pos = rs.samples_pos
    N = pos.shape[0]
    assert N == 2097152
    cfg, enc, mlp = hp.enc.cfg, hp.enc, hp.mlp
    with torch.no_grad():      # a lattice with visible magnitudes (the initialisation's 1e-5 would leave the features ~ 0)
        enc.lattice_values.copy_(torch.randn(enc.lattice_values.shape, generator=torch.Generator().manual_seed(3)).to(dev) * 0.1)
    win = torch.ones(L_, device=dev)
    ws_d, bs_d = [l.weight for l in mlp.layers], [l.bias for l in mlp.layers]
    g = torch.Generator().manual_seed(8)
    dy = torch.randn(N, generator=g)
    if dy_kind != "normal":
        dy = dy * 10.0 ** (torch.rand(N, generator=g) * 6.0 - 3.0)
        dy[123457] = 50.0 * float(dy.abs().max())
    assert int((dy == 0).sum()) == 0
    dY = dy.view(1, N).to(dev)
    # ---- the timed kernels, full size, in one piece
    feat = encode_forward_raw(cfg, pos, enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), win)
    from permuto_sdf_amd.mlp import f16_forward_supported
    f16 = hp.fwd_f16 and f16_forward_supported(mlp.dims)          # what SdfHotPath.forward (the bench) runs
    sdf = mlp_forward_raw(mlp.dims, feat, pack_params(mlp.dims, ws_d, bs_d, f16=f16), f16=f16)
    d_feat, dWs, dbs = mlp_backward_raw(mlp.dims, feat, ws_d, bs_d, dY, need_dx=True)
    g_lat = torch.zeros_like(enc.lattice_values)
    encode_backward_raw(cfg, pos, enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), win,
                        d_feat, g_lat, None)
    torch.cuda.synchronize()
    assert last_path(2) in (2, 3) and last_path(0) == 2 and last_path(1) in (2, 4)
    # ---- float64 on the GPU, chunked
    mods = []
    for i, l in enumerate(mlp.layers):
        lin = torch.nn.Linear(l.weight.shape[1], l.weight.shape[0]).to(dev).double()
        lin.weight.data.copy_(l.weight.detach().double())
        lin.bias.data.copy_(l.bias.detach().double())
        mods += [lin] + ([torch.nn.GELU()] if i < len(mlp.layers) - 1 else [])
    net = torch.nn.Sequential(*mods)
    lins = [m for m in mods if isinstance(m, torch.nn.Linear)]
    # the SAME net evaluated the way the reference evaluates it: unmodified torch.nn in fp32 on this GPU (rocBLAS), measured
    # against the same float64 -- the calibration of "fp32-equivalent" (round 6): printed beside ours, and ours must stay within
    # a small factor of it
    import copy
    net32 = copy.deepcopy(net).float()
    lins32 = [m for m in net32 if isinstance(m, torch.nn.Linear)]
    t_sdf = t_dx = 0.0
    g64 = torch.zeros(L_, T_, F_, dtype=torch.float64, device=dev)
    sf = po.scale_factors(enc.scale_per_level, 3)
    shifts = enc.random_shift_per_level.detach()
    e_sdf = e_dx = e_rows = 0.0
    dx_max = 0.0
    CH = 1 << 17
    for c0 in range(0, N, CH):
        sl = slice(c0, c0 + CH)
        x = feat[:, sl].t().double().requires_grad_(True)
        y = net(x)
        y.backward(dY[:, sl].t().double())
        e_sdf = max(e_sdf, float((sdf[0, sl].double() - y[:, 0]).abs().max()))
        e_dx = max(e_dx, float((d_feat[:, sl].t().double() - x.grad).abs().max()))
        dx_max = max(dx_max, float(x.grad.abs().max()))
        x32 = feat[:, sl].t().clone().requires_grad_(True)
        y32 = net32(x32)
        y32.backward(dY[:, sl].t())
        t_sdf = max(t_sdf, float((y32[:, 0].double() - y[:, 0]).abs().max()))
        t_dx = max(t_dx, float((x32.grad.double() - x.grad).abs().max()))
        pc = pos[sl]            # the restatement's elementwise fp32 arithmetic, evaluated on the GPU (2 M points x 16 levels)
        for l in range(L_):
            rem0, rank, bary = po.simplex(pc, shifts[l], sf[l])
            idx = po.vertex_indices(rem0, rank, T_)
            bw = bary[:, :4].double()
            gl = x.grad[:, l * F_:(l + 1) * F_]
            # float64 scatter WITHOUT atomics (a coarse level sends the whole chunk to a handful of rows: float64 atomics on one
            # address serialise for seconds): sort by row, running sum, differences at the row boundaries
            rows_all = idx.t().reshape(-1)                                             # [4 n]: vertex 0 of every sample, then 1, ...
            vals_all = torch.cat([gl * bw[:, r:r + 1] for r in range(4)], 0)           # [4 n, F]
            order = torch.argsort(rows_all)
            rs_, cs_ = rows_all[order], vals_all[order].cumsum(0)
            uniq, counts = torch.unique_consecutive(rs_, return_counts=True)
            ends = counts.cumsum(0) - 1
            seg = cs_[ends]
            seg[1:] = seg[1:] - cs_[ends[:-1]]
            g64[l][uniq] += seg
            if c0 == 0:     # the rows / weights used for the float64 scatter are the kernel's: they reproduce its features
                f_chk = sum(enc.lattice_values[l].detach().double().index_select(0, idx[:, r]) * bw[:, r:r + 1] for r in range(4))
                e_rows = max(e_rows, float((f_chk - feat[l * F_:(l + 1) * F_, sl].t().double()).abs().max()))
    errs = {"sdf": e_sdf / float(sdf.abs().max()), "d_features": e_dx / dx_max,
            "lattice_grad": float((g_lat.double() - g64).abs().max() / g64.abs().max())}
    for i, lin in enumerate(lins):
        errs["dW%d" % i] = float((dWs[i].double() - lin.weight.grad).abs().max() / lin.weight.grad.abs().max())
        errs["db%d" % i] = float((dbs[i].double() - lin.bias.grad).abs().max() / lin.bias.grad.abs().max())
    t32 = {"sdf": t_sdf / float(sdf.abs().max()), "d_features": t_dx / dx_max}
    for i, (lin, l32) in enumerate(zip(lins, lins32)):
        t32["dW%d" % i] = float((l32.weight.grad.double() - lin.weight.grad).abs().max() / lin.weight.grad.abs().max())
        t32["db%d" % i] = float((l32.bias.grad.double() - lin.bias.grad).abs().max() / lin.bias.grad.abs().max())
    per_level = [float((g_lat[l].double() - g64[l]).abs().max() / g64[l].abs().max()) for l in range(L_)]
    print("   torch.nn fp32 (rocBLAS) on the same inputs vs the same float64: %s" % " ".join("%s %.1e" % kv for kv in t32.items()))
    print("   ratio ours / torch-fp32: %s" % " ".join("%s %.1f" % (k, errs[k] / max(t32[k], 1e-12)) for k in t32))
    assert e_rows <= 2e-6 * float(feat.abs().max()), e_rows
    print("cfg 2 (L = %d), all 2 097 152 samples carry gradient (%s), kernels (fwd %d, bwd %d) vs float64: %s; lattice per level max %.1e"
          % (L_, dy_kind, last_path(2), last_path(1), " ".join("%s %.1e" % kv for kv in errs.items()), max(per_level)))
    assert max(errs.values()) < 5e-5, errs
    assert max(per_level) < TOL, per_level
