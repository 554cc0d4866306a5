"""The call patterns of the reference's Python (permuto_sdf_py), replayed against the drop-in WITHOUT the reference tree
(it does not exist on the GPU box).  tools/run_reference_on_gpu.py runs the real thing (log under profiles/); this file
pins the hazards that only execution finds:

  * `torch.set_default_tensor_type(torch.cuda.FloatTensor)` (train_permuto_sdf.py:71): bare torch.tensor/ones/rand land
    on the GPU while the drop-in is in use (subprocess, so the setting cannot leak into other tests);
  * non-contiguous inputs (the reference's accessors are stride-aware; the raw-pointer C ABI must be fed contiguous copies);
  * attributes of a RaySamplesPacked overwritten after production (sdf_utils.py:216 assigns samples_pos; here also
    ray_start_end_idx) -> the container's "already packed" knowledge must be dropped, counts recomputed;
  * sphere_trace's loop (sdf_utils.py:120-218): boolean-mask gather, `advance_sample_to_next_occupied_voxel` writing into
    its argument, scatter back, `samples_pos = pts`, integrate on the traced container;
  * `initialize_with_one_sample_per_ray` (src/RaySamplesPacked.cu:97-122): int32 ranges on the GPU, members
    updated, the container usable by per-ray kernels;
  * importance sampling's sequence (sdf_utils.py:383-423): set_sdf with a VIEW, in-place normalisation, remove_sdf.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(scope="module")
def port():
    return O.Oracle("port")


@pytest.fixture(scope="module")
def world(port, dev):
    from permuto_sdf import OccupancyGrid, Sphere
    n = 64
    occ = scene.shell_occupancy(port, n)
    o, d = scene.make_rays(1500, seed=11)
    grid = OccupancyGrid(n, 1.0, [0, 0, 0])
    grid.set_grid_occupancy(T(occ, dev))
    return dict(n=n, occ=occ, grid=grid, sphere=Sphere(0.5, [0, 0, 0]), o=o, d=d)


DEFAULT_TENSOR_TYPE_SCRIPT = r'''
import sys, math
sys.path.insert(0, %(root)r)
import numpy as np, torch
torch.manual_seed(0)
torch.set_default_tensor_type(torch.cuda.FloatTensor)            # train_permuto_sdf.py:71
import permutohedral_encoding as permuto_enc
from permuto_sdf import OccupancyGrid, Sphere, VolumeRendering, RaySampler, PermutoSDF
assert torch.ones(1).is_cuda and torch.tensor(1.0).is_cuda       # what the reference relies on
enc = permuto_enc.PermutoEncoding(3, 2 ** 14, 8, 2, np.geomspace(1.0, 1e-3, 8), appply_random_shift_per_level=True,
                                  concat_points=True, concat_points_scaling=1e-3).to("cuda")
mlp = torch.nn.Sequential(torch.nn.Linear(enc.output_dims(), 32), torch.nn.GELU(), torch.nn.Linear(32, 32), torch.nn.GELU(),
                          torch.nn.Linear(32, 1 + 4)).to("cuda")
c2f = permuto_enc.Coarse2Fine(8)
aabb = Sphere(0.5, [0, 0, 0])
assert aabb.m_center_tensor.is_cuda
grid = OccupancyGrid(32, 1.0, [0, 0, 0])
pts = aabb.rand_points_inside(nr_points=4096)
assert pts.is_cuda and float(pts.norm(dim=1).max()) <= 0.5 + 1e-6
# models.py:236-251: sdf + gradient with create_graph, then a loss on the gradient (double backward through the encoding)
with torch.set_grad_enabled(True):
    pts.requires_grad_(True)
    window = c2f(0.65)
    y = mlp(enc(pts, window.view(-1)))
    sdf = y[:, 0:1]
    (g,) = torch.autograd.grad(sdf, pts, torch.ones_like(sdf, requires_grad=False), create_graph=True, retain_graph=True)
    loss = ((sdf - (pts.norm(dim=-1, keepdim=True) - 0.3)) ** 2).mean() * 3e3 + ((g.norm(dim=-1) - 1.0) ** 2).mean() * 5e1
loss.backward()
lat = [p for n, p in enc.named_parameters() if "lattice_values" in n][0]
assert lat.grad is not None and torch.isfinite(lat.grad).all() and float(lat.grad.abs().max()) > 0
assert all(torch.isfinite(p.grad).all() for p in mlp.parameters())
# rays with bare constructors, samplers, compositing: everything stays on the GPU
o = torch.tensor([[0.0, 0.0, -1.5]]).repeat(256, 1) + torch.rand(256, 3) * 0.05
d = torch.nn.functional.normalize(-o + torch.randn(256, 3) * 0.1, dim=1)
_, te, _, tx, hit = aabb.ray_intersection(o, d)
fg = grid.compute_samples_in_occupied_regions(o, d, te, tx, 1e-4, 32, True).compact_to_valid_samples()
bg = RaySampler.compute_samples_bg(o, d, tx, 16, aabb.m_radius, aabb.m_center_tensor, True, False)
assert fg.samples_pos.is_cuda and fg.samples_pos.shape[0] > 0 and bg.samples_pos_4d.shape == (256 * 16, 4)
alpha = VolumeRendering.sdf2alpha(fg, fg.samples_pos.norm(dim=1, keepdim=True) - 0.3, 512, True, 1.0).clip(0.0, 1.0)
Tm, bgT = VolumeRendering.cumprod_alpha2transmittance(fg, 1 - alpha + 1e-7)
assert Tm.is_cuda and bgT.shape == (256, 1)
sh = PermutoSDF.spherical_harmonics(fg.samples_dirs, 5)
assert sh.shape[1] == 25
inv_s = torch.exp(torch.tensor(1.0) * 0.3 * 10.0).clip(1e-6, 1e6)             # SingleVarianceNetwork forward
cen, idx = grid.compute_random_sample_of_grid_points(2048, True)
grid.update_with_sdf_random_sample(idx, cen.norm(dim=1, keepdim=True) - 0.3, inv_s.view(-1), 1e-4)
print("DEFAULT_TENSOR_TYPE_OK")
'''


def test_cuda_default_tensor_type(dev):
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "-c", DEFAULT_TENSOR_TYPE_SCRIPT % {"root": ROOT}], capture_output=True, text=True,
                       env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and "DEFAULT_TENSOR_TYPE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_non_contiguous_inputs_equal_contiguous(world, dev):
    from permuto_sdf import PermutoSDF, RaySampler
    from permutohedral_encoding import PermutoEncoding
    o, d = T(world["o"], dev), T(world["d"], dev)
    wide = torch.cat([o, d, o], 1)                       # [R, 9]: column slices are strided views
    o_nc, d_nc = wide[:, 0:3], wide[:, 3:6]
    o_t = o.t().contiguous().t()                         # column-major storage
    assert not o_nc.is_contiguous() and not o_t.is_contiguous()
    sph, grid = world["sphere"], world["grid"]
    ref = sph.ray_intersection(o, d)
    for a, b in zip(sph.ray_intersection(o_nc, d_nc), ref):
        assert torch.equal(a, b)
    for a, b in zip(sph.ray_intersection(o_t, d_nc), ref):
        assert torch.equal(a, b)
    te, tx = ref[1], ref[3]
    te_nc = torch.cat([te, tx], 1)[:, 0:1]
    s0 = grid.compute_samples_in_occupied_regions(o, d, te, tx, 1e-4, 48, False).compact_to_valid_samples()
    s1 = grid.compute_samples_in_occupied_regions(o_nc, d_nc, te_nc, tx, 1e-4, 48, False).compact_to_valid_samples()
    assert torch.equal(s0.samples_pos, s1.samples_pos) and torch.equal(s0.ray_start_end_idx, s1.ray_start_end_idx)
    b0 = RaySampler.compute_samples_bg(o, d, tx, 8, 0.5, sph.m_center_tensor, False, False)
    b1 = RaySampler.compute_samples_bg(o_t, d_nc, tx, 8, 0.5, sph.m_center_tensor, False, False)
    assert torch.equal(b0.samples_pos_4d, b1.samples_pos_4d)
    assert torch.equal(PermutoSDF.spherical_harmonics(d, 4), PermutoSDF.spherical_harmonics(d_nc, 4))
    # encoding: strided points, and a strided upstream gradient (a column slice of a wider gradient)
    torch.manual_seed(1)
    enc = PermutoEncoding(3, 2 ** 12, 6, 2, np.geomspace(1.0, 1e-2, 6), concat_points=True, concat_points_scaling=1.0).to(dev)
    with torch.no_grad():
        enc.lattice_values.normal_()
    p = (o * 0.3).clone()
    p_nc = torch.cat([p, p], 1)[:, 3:6]
    assert torch.equal(enc(p), enc(p_nc))
    up = torch.randn(p.shape[0], enc.output_dims() + 5, device=dev)
    outs = []
    for pts, g in ((p, up[:, 2:2 + enc.output_dims()].contiguous()), (p_nc, up[:, 2:2 + enc.output_dims()])):
        enc.lattice_values.grad = None
        pts = pts.detach().requires_grad_(True)
        enc(pts).backward(g)
        outs.append((enc.lattice_values.grad.clone(), pts.grad.clone()))
    # same kernels on the same values: the scatter order of the lattice gradient may differ between launches
    for a, b in zip(outs[0], outs[1]):
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max())


def test_attribute_overwrite_invalidates_packed_shortcut(world, dev):
    sph, grid = world["sphere"], world["grid"]
    o, d = T(world["o"], dev), T(world["d"], dev)
    _, te, _, tx, _ = sph.ray_intersection(o, d)
    pool = grid.compute_samples_in_occupied_regions(o, d, te, tx, 1e-4, 48, False)
    assert pool._exact
    full = pool.compact_to_valid_samples()
    n_full = full.samples_pos.shape[0]
    assert n_full == pool.compute_exact_nr_samples() == int((full.ray_start_end_idx[:, 1] - full.ray_start_end_idx[:, 0]).sum())
    # overwriting sample tensors of the same shape keeps the packing (sdf_utils.py:216)
    full.samples_pos = full.samples_pos + 0.0
    assert full._exact
    # Python drops every second ray by re-assigning the ranges: the container is no longer dense
    se = pool.ray_start_end_idx.clone()
    se[::2] = 0
    pool.ray_start_end_idx = se
    assert not pool._exact
    expect = int((se[:, 1] - se[:, 0]).sum())
    assert pool.compute_exact_nr_samples() == expect
    half = pool.compact_to_valid_samples()
    assert half.samples_pos.shape[0] == expect < n_full
    # the surviving rays carry exactly their old samples, re-packed densely in ray order
    hs, fs = half.ray_start_end_idx.cpu().numpy(), full.ray_start_end_idx.cpu().numpy()
    hp, fp = half.samples_pos.cpu().numpy(), full.samples_pos.cpu().numpy()
    cursor = 0
    for r in range(1, hs.shape[0], 2):
        cnt = fs[r, 1] - fs[r, 0]
        assert hs[r, 1] - hs[r, 0] == cnt and hs[r, 0] == cursor
        assert np.array_equal(hp[hs[r, 0]:hs[r, 1]], fp[fs[r, 0]:fs[r, 1]])
        cursor += cnt
    assert (hs[::2, 1] - hs[::2, 0] == 0).all()


def test_initialize_with_one_sample_per_ray(world, dev):
    """src/RaySamplesPacked.cu:97-122 done right: cuda int32 ranges [i, i+1), members updated"""
    from permuto_sdf import RaySamplesPacked, VolumeRendering
    o, d = T(world["o"], dev), T(world["d"], dev)
    R = o.shape[0]
    entry = world["sphere"].ray_intersection(o, d)[0]
    rs = RaySamplesPacked(R, R)
    rs.initialize_with_one_sample_per_ray(entry, d)
    assert rs.samples_pos.data_ptr() == entry.data_ptr() and rs.samples_dirs.data_ptr() == d.data_ptr()   # re-pointed, no copy
    se = rs.ray_start_end_idx
    assert se.is_cuda and se.dtype == torch.int32 and tuple(se.shape) == (R, 2)
    assert torch.equal(se[:, 0].cpu(), torch.arange(R, dtype=torch.int32)) and torch.equal(se[:, 1], se[:, 0] + 1)
    assert rs.max_nr_samples == R and rs.rays_have_equal_nr_of_samples and rs.fixed_nr_of_samples_per_ray == 1
    assert not rs.has_sdf and rs.compute_exact_nr_samples() == R
    assert tuple(rs.samples_z.shape) == (R, 1) and tuple(rs.samples_dt.shape) == (R, 1)
    # usable by the per-ray kernels: integrating a per-sample value with unit weights returns it (one sample per ray)
    vals = torch.rand(R, 3, device=dev)
    out = VolumeRendering.integrate_with_weights(rs, vals, torch.ones(R, 1, device=dev))
    assert torch.equal(out, vals)
    s, per_sample = VolumeRendering.sum_over_each_ray(rs, vals[:, 0:1].contiguous())
    assert torch.equal(s, vals[:, 0:1]) and torch.equal(per_sample, vals[:, 0:1])
    assert torch.equal(RaySamplesPacked.compute_per_sample_ray_idx(se, R).cpu(), torch.arange(R, dtype=torch.int32))
    same = rs.compact_to_valid_samples()
    assert same.samples_pos.shape[0] == R


def test_sphere_trace_loop_pattern_against_oracle(world, port, dev):
    """sdf_utils.py:120-218 with an analytic SDF, the oracle running the same loop on the CPU (advance: bit-exact kernel)"""
    from permuto_sdf import VolumeRendering
    sph, grid = world["sphere"], world["grid"]
    o, d = T(world["o"], dev), T(world["d"], dev)
    _, te, _, tx, _ = sph.ray_intersection(o, d)
    rs = grid.compute_first_sample_start_of_occupied_regions(o, d, te, tx).compact_to_valid_samples()
    pos, dirs = rs.samples_pos, rs.samples_dirs
    voxel = 1.0 / grid.get_nr_voxels_per_dim()
    pos = pos + dirs * voxel * 0.5
    pts = pos.clone()
    conv = torch.zeros_like(pos)[:, 0:1].bool()
    # oracle side
    gridnp = (world["n"], 1.0, [0, 0, 0], world["occ"])
    _, te_n, _, tx_n, _ = port.sphere_intersect(0.5, [0, 0, 0], world["o"], world["d"])
    fh = port.compact(port.first_hit_samples(world["o"], world["d"], te_n, tx_n, 1 << 21, gridnp))
    pos_n, dirs_n = fh.pos, fh.dirs
    assert np.array_equal(rs.samples_pos.cpu().numpy(), pos_n)
    pts_n = (pos_n + dirs_n * np.float32(voxel) * np.float32(0.5)).astype(np.float32)
    assert np.array_equal(pts.cpu().numpy(), pts_n)
    conv_n = np.zeros((pts_n.shape[0], 1), bool)

    def sdf_t(p):
        return p.norm(dim=1, keepdim=True) - 0.3

    for it in range(6):
        sel = torch.logical_not(conv)
        pu = pts[sel.repeat(1, 3)].view(-1, 3)
        du = dirs[sel.repeat(1, 3)].view(-1, 3)
        if pu.shape[0] == 0:
            break
        s = sdf_t(pu)
        pu = pu + du * s * 0.9
        conv[sel] = torch.logical_or(conv[sel], (s.abs() < 2e-4).view(-1))
        conv = conv.view(-1, 1)
        pu_before = pu
        pu, inb = grid.advance_sample_to_next_occupied_voxel(du, pu)
        assert pu.data_ptr() == pu_before.data_ptr()                    # in place, like src/OccupancyGrid.cu:311
        conv[sel] = torch.logical_or(conv[sel], torch.logical_not(inb.view(-1)))
        conv = conv.view(-1, 1)
        pts[sel.repeat(1, 3)] = pu.view(-1)
        # ---- the same iteration on the CPU: the torch ops above on host tensors + the oracle's advance
        seln = ~conv_n[:, 0]
        pun, dun = torch.from_numpy(pts_n[seln]), torch.from_numpy(dirs_n[seln])
        # the SDF values are the GPU's (torch's norm differs in the last bit between devices; the network is not what
        # this test is about): everything after them -- step, convergence test, voxel march -- is replayed on the CPU
        assert pun.shape[0] == s.shape[0]
        sn = s.cpu()
        pun = (pun + dun * sn * 0.9).numpy()
        newly = (sn.abs() < 2e-4).numpy()[:, 0]
        adv, inbn = port.advance_samples(dun.numpy(), pun, gridnp)
        conv_n[seln, 0] |= newly | ~inbn.reshape(-1).astype(bool)
        pts_n[seln] = adv
        assert np.array_equal(pts.cpu().numpy(), pts_n), it
        assert np.array_equal(conv.cpu().numpy(), conv_n), it
    rs.samples_pos = pts                                                # sdf_utils.py:216
    w = torch.ones_like(pts)[:, 0:1]
    w[conv.logical_not()] = 0.0
    integ = VolumeRendering.integrate_with_weights(rs, pts, w)
    assert integ.shape == (o.shape[0], 3)
    per_ray_idx = rs.ray_start_end_idx[:, 0].long()
    has = (rs.ray_start_end_idx[:, 1] - rs.ray_start_end_idx[:, 0]) == 1
    assert torch.equal(integ[has], (pts * w)[per_ray_idx[has]])
    assert torch.equal(integ[~has], torch.zeros_like(integ[~has]))


def test_importance_sampling_sequence_with_views(world, port, dev):
    """sdf_utils.py:383-423: set_sdf(view of a wider network output), in-place `weights /= ...`, remove_sdf, two rounds"""
    from permuto_sdf import VolumeRendering
    sph, grid = world["sphere"], world["grid"]
    o, d = T(world["o"], dev), T(world["d"], dev)
    _, te, _, tx, _ = sph.ray_intersection(o, d)
    rs = grid.compute_samples_in_occupied_regions(o, d, te, tx, 1e-4, 32, False).compact_to_valid_samples()

    def net(p):      # [N, 1+4] like sdf_and_feat; the SDF is a column VIEW (models.py:190)
        return torch.cat([p.norm(dim=1, keepdim=True) - 0.3, p, p[:, 0:1]], 1)

    sdf = net(rs.samples_pos)[:, 0:1]
    assert not sdf.is_contiguous()
    rs.set_sdf(sdf)
    assert rs.has_sdf
    for mult, last in ((1.0, False), (2.0, True)):
        alpha = VolumeRendering.sdf2alpha(rs, rs.samples_sdf, 512, True, mult).clip(0.0, 1.0)
        Tm, _ = VolumeRendering.cumprod_alpha2transmittance(rs, 1 - alpha + 1e-7)
        w = alpha * Tm
        _, per = VolumeRendering.sum_over_each_ray(rs, w)
        per = torch.clamp(per, min=1e-6)
        w /= per
        cdf = VolumeRendering.compute_cdf(rs, w)
        imp = VolumeRendering.importance_sample(o, d, rs, cdf, 16, False)
        if not last:
            imp.set_sdf(net(imp.samples_pos)[:, 0:1])
        else:
            rs.remove_sdf()
        n_before = rs.samples_pos.shape[0]
        comb = VolumeRendering.combine_uniform_samples_with_imp(o, d, tx, rs, imp)
        # round 4: the merged total follows from the march's per-ray counts, which were on the host once (a ray holds 0 or >= 3
        # samples, a round adds 16 to every non-empty one): the compaction below does not sync -- and must get the same container
        assert comb._known_total is not None and comb._known_total == int(comb.cur_nr_samples.item())
        rs = comb.compact_to_valid_samples()
        se = rs.ray_start_end_idx
        cnt = (se[:, 1] - se[:, 0])
        assert rs.samples_pos.shape[0] == int(cnt.sum()) and rs.samples_pos.shape[0] >= n_before
        z = rs.samples_z.view(-1)
        ridx = rs.compute_per_sample_ray_idx(se, z.shape[0]).long()
        same = ridx[1:] == ridx[:-1]
        assert bool(((z[1:] >= z[:-1]) | ~same).all())                      # merged samples stay sorted inside each ray
        if not last:
            # the merged container carries the SDF of both sources at the merged positions
            assert torch.allclose(rs.samples_sdf, rs.samples_pos.norm(dim=1, keepdim=True) - 0.3, atol=2e-6)


def test_compaction_of_a_whole_image_reads_the_total_not_the_counts(world, dev, monkeypatch):
    """above bridge._COUNTS_ON_HOST_MAX_RAYS rays the compaction of a march reads the 4-byte total (per-ray counts of a 512x512
    image are a 1 MB copy and a threaded host reduction: +12 ms per image, measured); the container is the same either way, only
    the host-side knowledge that lets the NEXT merge skip its sync is absent"""
    from permuto_sdf_amd import bridge
    sph, grid = world["sphere"], world["grid"]
    o, d = T(world["o"], dev), T(world["d"], dev)
    _, te, _, tx, _ = sph.ray_intersection(o, d)
    a = grid.compute_samples_in_occupied_regions(o, d, te, tx, 1e-4, 32, False)
    assert a._ray_counts is not None and o.shape[0] <= bridge._COUNTS_ON_HOST_MAX_RAYS
    ca = a.compact_to_valid_samples()
    assert ca._host_nonempty == int((a._ray_counts > 0).sum().item())
    monkeypatch.setattr(bridge, "_COUNTS_ON_HOST_MAX_RAYS", o.shape[0] - 1)
    b = grid.compute_samples_in_occupied_regions(o, d, te, tx, 1e-4, 32, False)
    cb = b.compact_to_valid_samples()
    assert cb._host_nonempty is None and cb._known_total is None
    assert cb.samples_pos.shape == ca.samples_pos.shape and torch.equal(cb.samples_pos, ca.samples_pos)
    assert torch.equal(cb.ray_start_end_idx, ca.ray_start_end_idx) and torch.equal(cb.samples_z, ca.samples_z)
    assert int(cb.cur_nr_samples.item()) == int(ca.cur_nr_samples.item()) == ca.samples_pos.shape[0]


def test_fused_evaluators_behind_reference_style_models(dev, monkeypatch):
    """PSDF_FUSE_REFERENCE_MLPS=1 (permuto_sdf_amd/reference_fusion.py): a model class built like the reference's `SDF`
    (models.py:132-203: encoding constructed first inside __init__, then a Linear/GELU nn.Sequential, forward = encoding ->
    mlp) gets its Sequential swapped for the fused evaluator at the first forward -- same Parameters, same state_dict keys --
    and sdf, its input gradient (create_graph=True, models.py:236-251) and every parameter gradient of an eikonal-style loss
    equal the unfused torch evaluation within 1e-4.  A reference-style LipshitzMLP owner (`RGB`) is swapped as well."""
    import numpy as np
    import permutohedral_encoding as permuto_enc
    from permuto_sdf_amd import reference_fusion as RF
    from permuto_sdf_amd.mlp import LipshitzMLP as OurLipshitz

    class SDF(torch.nn.Module):        # the name is what the owner detection looks for
        def __init__(self):
            super().__init__()
            self.encoding = permuto_enc.PermutoEncoding(3, 2 ** 14, 24, 2, np.geomspace(1.0, 1e-3, 24), appply_random_shift_per_level=True,
                                                        concat_points=True, concat_points_scaling=1e-3)
            self.mlp_sdf = torch.nn.Sequential(torch.nn.Linear(self.encoding.output_dims(), 32), torch.nn.GELU(), torch.nn.Linear(32, 32),
                                               torch.nn.GELU(), torch.nn.Linear(32, 32), torch.nn.GELU(), torch.nn.Linear(32, 33))
            self.c2f = permuto_enc.Coarse2Fine(24)

        def forward(self, points):
            return self.mlp_sdf(self.encoding(points, self.c2f(0.9).view(-1)))

    def run(model, pts0):
        for p in model.parameters():
            p.grad = None
        pts = pts0.clone().requires_grad_(True)
        y = model(pts)
        sdf = y[:, 0:1]
        (g,) = torch.autograd.grad(sdf, pts, torch.ones_like(sdf), create_graph=True)
        loss = sdf.abs().mean() + ((g.norm(dim=1) - 1.0) ** 2).mean() + 0.1 * y[:, 1:].pow(2).mean()
        loss.backward()
        return y.detach().clone(), g.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    torch.manual_seed(0)
    pts = (torch.rand(6000, 3, device=dev) - 0.5) * 0.8
    monkeypatch.delenv("PSDF_FUSE_REFERENCE_MLPS", raising=False)
    plain = SDF().to(dev)
    with torch.no_grad():
        plain.encoding.lattice_values.mul_(1e3)           # visible feature magnitudes
    y0, g0, grads0 = run(plain, pts)
    assert type(plain.mlp_sdf) is torch.nn.Sequential
    monkeypatch.setenv("PSDF_FUSE_REFERENCE_MLPS", "1")
    fused = SDF().to(dev)
    keys = list(fused.state_dict().keys())
    fused.load_state_dict(plain.state_dict())
    params_before = {k: id(p) for k, p in fused.named_parameters()}
    y1, g1, grads1 = run(fused, pts)
    assert isinstance(fused.mlp_sdf, RF.FusedSequential) and fused.mlp_sdf.fused
    assert list(fused.state_dict().keys()) == keys                                   # checkpoints are untouched
    assert {k: id(p) for k, p in fused.named_parameters()} == params_before          # an optimiser built before keeps working
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert rel(y1, y0) < 1e-4 and rel(g1, g0) < 1e-4, (rel(y1, y0), rel(g1, g0))
    assert set(grads1) == set(grads0)
    for k in grads0:
        assert rel(grads1[k], grads0[k]) < 1e-4, (k, rel(grads1[k], grads0[k]))

    # LipshitzMLP owner: a module that looks like the reference's class is replaced by ours, Parameters shared
    class LipshitzMLP(torch.nn.Module):
        def __init__(self, cin, outs, last_layer_linear):
            super().__init__()
            self.last_layer_linear = last_layer_linear
            self.layers = torch.nn.ModuleList()
            for c in outs:
                self.layers.append(torch.nn.Linear(cin, c))
                cin = c
            self.weights_per_layer = torch.nn.ParameterList([l.weight for l in self.layers])
            self.biases_per_layer = torch.nn.ParameterList([l.bias for l in self.layers])
            self.lipshitz_bound_per_layer = torch.nn.ParameterList(
                [torch.nn.Parameter(torch.ones(1) * l.weight.abs().sum(1).max().item() * 0.5) for l in self.layers])

        def forward(self, x):
            for i, l in enumerate(self.layers):
                c = torch.nn.functional.softplus(self.lipshitz_bound_per_layer[i])
                w = l.weight * torch.clamp(c / l.weight.abs().sum(1), max=1.0)[:, None]
                x = torch.nn.functional.linear(x, w, l.bias)
                if i < len(self.layers) - 1:
                    x = torch.nn.functional.gelu(x)
            return x

    class RGB(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.mlp = LipshitzMLP(111, [128, 128, 64, 3], True)

    net = RGB().to(dev)
    x = torch.randn(4000, 111, device=dev)
    ref_out = net.mlp(x)
    ref_out.pow(2).sum().backward()
    ref_grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    for p in net.parameters():
        p.grad = None
    keys = list(net.state_dict().keys())
    assert RF.fuse_model(net) == ["mlp"] and isinstance(net.mlp, OurLipshitz)
    assert list(net.state_dict().keys()) == keys
    out = net.mlp(x)
    out.pow(2).sum().backward()
    assert rel(out.detach(), ref_out.detach()) < 1e-4
    for k, p in net.named_parameters():
        assert rel(p.grad, ref_grads[k]) < 2e-4, k
