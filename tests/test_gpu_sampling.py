"""GPU parity (through the drop-in `permuto_sdf` API -> C ABI -> HIP) vs the CPU oracle for the sample-generation
rows: OccupancyGrid, RaySampler, Sphere, RaySamplesPacked, spherical harmonics, rays from reel.
Bar: bit-exact for everything built from + - * / sqrt floor (voxel indices, per-ray counts, sample positions);
1e-6 where device libm / rsqrt differ from the host's (stated per assertion)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import scene

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def bits_equal(t, a):
    t = t.detach().cpu().numpy()
    assert t.shape == a.shape, (t.shape, a.shape)
    if a.dtype == np.float32:
        assert np.array_equal(t.view(np.uint32), a.view(np.uint32)), float(np.abs(t - a).max())
    else:
        assert np.array_equal(t, a)


@pytest.fixture(scope="module")
def port():
    return O.Oracle("port")


# (n, extent, translation): the two unit grids take the kernels' compile-time specialised path (csrc/sampling.hip,
# GridFast), the third one -- extent != 1, shifted -- the generic path with the reference's divisions
@pytest.fixture(scope="module", params=[(64, 1.0, (0.0, 0.0, 0.0)), (256, 1.0, (0.0, 0.0, 0.0)), (64, 1.2, (0.04, -0.03, 0.02))],
                ids=["n64", "n256", "n64-extent1.2-shifted"])
def world(request, port, dev):
    from permuto_sdf import OccupancyGrid, Sphere
    n, extent, tr = request.param
    occ = scene.shell_occupancy(port, n, extent, tr)
    o, d = scene.make_rays(2000, seed=3)
    sph = Sphere(0.5, [0, 0, 0])
    grid = OccupancyGrid(n, extent, list(tr))
    grid.set_grid_occupancy(T(occ, dev))
    _, te, _, tx, _ = port.sphere_intersect(0.5, [0, 0, 0], o, d)
    return dict(n=n, occ=occ, gridnp=(n, extent, list(tr), occ), grid=grid, sphere=sph, o=o, d=d, te=te, tx=tx)


def test_sphere_intersection(port, dev):
    from permuto_sdf import Sphere
    o, d = scene.make_rays(5000, seed=9, jitter_target=0.8)
    c = [0.05, -0.02, 0.01]
    ref = port.sphere_intersect(0.5, c, o, d)
    out = Sphere(0.5, c).ray_intersection(T(o, dev), T(d, dev))
    for a, b in zip(out, ref):
        bits_equal(a, b)
    assert out[4].dtype == torch.bool
    # ray through the centre: t0 = |o| - r, t1 = |o| + r  (known answer)
    o1 = torch.tensor([[0.0, 0.0, -2.0]], device=dev)
    d1 = torch.tensor([[0.0, 0.0, 1.0]], device=dev)
    _, t0, _, t1, hit = Sphere(0.5, [0, 0, 0]).ray_intersection(o1, d1)
    assert float(t0) == 1.5 and float(t1) == 2.5 and bool(hit)
    # empty input
    e = Sphere(0.5, [0, 0, 0]).ray_intersection(torch.zeros(0, 3, device=dev), torch.zeros(0, 3, device=dev))
    assert e[0].shape == (0, 3) and e[4].shape == (0, 1)


def test_sphere_points_and_inside(port, dev):
    from permuto_sdf import Sphere
    s = Sphere(0.5, [0, 0, 0])
    torch.manual_seed(0)
    pts = s.rand_points_inside(30000)
    assert pts.shape == (30000, 3)
    assert float(pts.norm(dim=1).max()) <= 0.5 * (1 + 1e-5)
    assert bool(s.check_point_inside_primitive(pts).all())
    # kernel arithmetic vs oracle on explicit uniforms (device sin/cos/acos/pow differ from glibc by a few ulp)
    rng = np.random.default_rng(2)
    phi, ct, u = rng.uniform(0, 6.28, 4000), rng.uniform(-1, 1, 4000), rng.uniform(0, 1, 4000)
    ref = port.rand_points_inside(0.5, phi, ct, u)
    from permuto_sdf_amd import _lib as L
    out = torch.empty(4000, 3, device=dev)
    tp, tc, tu = (T(a.astype(np.float32), dev) for a in (phi, ct, u))    # keep the device buffers alive over the call
    L.call("psdf_sphere_rand_points_inside", L.c_i(4000), L.c_f(0.5), L.ptr(tp), L.ptr(tc), L.ptr(tu), L.ptr(out), L.stream())
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-6


def test_grid_points_and_morton(port, dev):
    from permuto_sdf import OccupancyGrid
    g = OccupancyGrid(32, 1.0, [0.1, -0.2, 0.05])
    assert g.get_nr_voxels() == 32 ** 3 and g.get_nr_voxels_per_dim() == 32
    bits_equal(g.compute_grid_points(False), port.grid_points(32, 1.0, [0.1, -0.2, 0.05]))
    # jittered points use the process-global generator: replay its state in the oracle
    st = (OccupancyGrid._rng.state, OccupancyGrid._rng.inc)
    bits_equal(g.compute_grid_points(True), port.grid_points(32, 1.0, [0.1, -0.2, 0.05], randomize=True, rng=st))
    assert OccupancyGrid._rng.state != st[0]          # advanced by 2^32 on the host, like the reference
    st = (OccupancyGrid._rng.state, OccupancyGrid._rng.inc)
    pts, idx = g.compute_random_sample_of_grid_points(5000, True)
    assert idx.dtype == torch.int32 and int(idx.min()) >= 0 and int(idx.max()) < 32 ** 3
    bits_equal(pts, port.grid_points(32, 1.0, [0.1, -0.2, 0.05], idx.cpu().numpy(), True, rng=st))
    # Morton round trip: centre of voxel i maps back to voxel i (occupancy one-hot probe)
    g2 = OccupancyGrid(16, 1.0, [0, 0, 0])
    centres = g2.compute_grid_points(False)
    for probe in (0, 1, 2, 4, 77, 16 ** 3 - 1):
        occ = torch.zeros(16 ** 3, dtype=torch.bool, device=dev)
        occ[probe] = True
        g2.set_grid_occupancy(occ)
        hit = g2.check_occupancy(centres).view(-1)
        assert int(hit.sum()) == 1 and bool(hit[probe])
    with pytest.raises(ValueError):
        OccupancyGrid(48, 1.0, [0, 0, 0])


def test_grid_updates(port, dev):
    from permuto_sdf import OccupancyGrid
    n = 32
    rng = np.random.default_rng(1)
    vals = rng.uniform(0, 2, n ** 3).astype(np.float32)
    occ = rng.uniform(size=n ** 3) > 0.5
    dens = rng.uniform(0, 3, (n ** 3, 1)).astype(np.float32)

    def fresh():
        g = OccupancyGrid(n, 1.0, [0, 0, 0])
        g.set_grid_values(T(vals, dev).clone())
        g.set_grid_occupancy(T(occ, dev).clone())
        return g

    g = fresh()
    g.update_with_density(T(dens, dev), 0.95, 0.5)
    rv, ro = port.update_with_density(vals, occ, dens, 0.95, 0.5)
    bits_equal(g.get_grid_values(), rv)
    bits_equal(g.get_grid_occupancy(), ro)
    idx = np.unique(rng.integers(0, n ** 3, 6000)).astype(np.int32)       # unique: duplicates race in the reference
    g = fresh()
    g.update_with_density_random_sample(T(idx, dev), T(dens[:len(idx)], dev), 0.9, 0.7)
    rv, ro = port.update_with_density(vals, occ, dens[:len(idx)], 0.9, 0.7, idx)
    bits_equal(g.get_grid_values(), rv)
    bits_equal(g.get_grid_occupancy(), ro)
    sdf = rng.normal(0, 0.05, (n ** 3, 1)).astype(np.float32)
    for inv_s in (64.0, 512.0):
        g = fresh()
        g.update_with_sdf(T(sdf, dev), inv_s, 0.0, 1e-4)
        rv, ro = port.update_with_sdf(vals, occ, sdf, n, 1.0, inv_s, 1e-4)
        bits_equal(g.get_grid_values(), rv)
        # occupancy threshold goes through exp/pow: allow the device libm to flip voxels within 1e-5 of the threshold
        mism = (g.get_grid_occupancy().cpu().numpy() != ro).mean()
        assert mism < 1e-4, mism
        g = fresh()
        g.update_with_sdf_random_sample(T(idx, dev), T(sdf[:len(idx)], dev), torch.tensor([inv_s], device=dev), 1e-4)
        rv, ro = port.update_with_sdf(vals, occ, sdf[:len(idx)], n, 1.0, inv_s, 1e-4, idx)
        bits_equal(g.get_grid_values(), rv)
        assert (g.get_grid_occupancy().cpu().numpy() != ro).mean() < 1e-4


def test_check_occupancy_and_advance(port, world, dev):
    w = world
    pts = np.random.default_rng(5).uniform(-0.6, 0.6, (20000, 3)).astype(np.float32)   # includes out-of-grid points
    # coordinates that exercise the float -> uint32 conversion of pos_to_idx (one saturating hardware instruction on the GPU,
    # explicit compares in the oracle): NaN, infinities, values around 2^32 / n, negative zero, far outside either way
    special = np.array([np.nan, np.inf, -np.inf, -0.0, 0.0, 1e30, -1e30, 2.0 ** 32, 2.0 ** 31, 16777216.0, -16777216.0, 3.9, 4.1,
                        1e-30, -1e-30, 0.49999997, 0.5, 0.50000006], dtype=np.float32)
    sp = np.stack(np.meshgrid(special, special[:6], special[:4], indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    pts[:sp.shape[0]] = sp
    bits_equal(w["grid"].check_occupancy(T(pts, dev)), port.check_occupancy(*w["gridnp"][:3], w["occ"], pts))
    dirs = w["d"][:1000]
    start = np.clip(w["o"][:1000] + w["te"][:1000] * dirs, -0.49, 0.49).astype(np.float32)
    ref_pos, ref_in = port.advance_samples(dirs, start, w["gridnp"])
    p = T(start, dev)
    new_pos, within = w["grid"].advance_sample_to_next_occupied_voxel(T(dirs, dev), p)
    bits_equal(new_pos, ref_pos)
    bits_equal(within, ref_in)
    assert new_pos.data_ptr() == p.data_ptr()      # in place, like the reference


def compare_packed(rs, ref, exact=True):
    """rs: RaySamplesPacked (already compact / ray ordered); ref: compacted oracle Samples."""
    n = ref.total()
    assert rs.compute_exact_nr_samples() == n
    c = rs.compact_to_valid_samples()
    assert c.samples_pos.shape[0] == n
    bits_equal(c.ray_start_end_idx, ref.start_end)
    bits_equal(c.ray_fixed_dt, ref.fixed_dt)
    for name, arr in (("samples_pos", ref.pos), ("samples_dirs", ref.dirs), ("samples_z", ref.z), ("samples_dt", ref.dt)):
        bits_equal(getattr(c, name), arr[:n])
    return c


def _march_form():
    import ctypes
    from permuto_sdf_amd import _lib as L
    fn = L.lib().psdf_march_form
    fn.restype = ctypes.c_int
    return int(fn())


@pytest.mark.parametrize("form", ["thread", "quad"])
@pytest.mark.parametrize("jitter", [False, True])
def test_compute_samples_in_occupied_regions(port, world, dev, jitter, form, monkeypatch):
    """both forms of the march (csrc/sampling.hip: a thread per ray; four lanes per ray with the recorded first walk, round 6)
    against the oracle, bit for bit; the generic (non-unit) grid has no quad form and runs the thread kernel either way"""
    from permuto_sdf import OccupancyGrid
    w = world
    monkeypatch.setenv("PSDF_MARCH_FORM", form)
    st = (OccupancyGrid._rng.state, OccupancyGrid._rng.inc)
    rs = w["grid"].compute_samples_in_occupied_regions(T(w["o"], dev), T(w["d"], dev), T(w["te"], dev), T(w["tx"], dev),
                                                       1e-3, 64, jitter)
    assert _march_form() == (2 if (form == "quad" and w["gridnp"][1] == 1.0) else 1)
    ref = port.march_samples(w["o"], w["d"], w["te"], w["tx"], 1e-3, 64, 1 << 21, grid=w["gridnp"], jitter=jitter, rng=st)
    ref_c = port.compact(ref)
    assert ref_c.total() > 10000
    counts = ref_c.counts()
    assert ((counts == 0) | (counts >= 3)).all()             # invariant: a ray has 0 or >= 3 samples
    c = compare_packed(rs, ref_c)
    ridx = c.compute_per_sample_ray_idx(c.ray_start_end_idx, c.samples_pos.shape[0])
    bits_equal(ridx, port.per_sample_ray_idx(ref_c.start_end, ref_c.total()))


def test_generic_compaction_with_holes(port, world, dev):
    """compact_to_valid_samples on a container with holes (the reference's pool layout, produced by the oracle)."""
    from permuto_sdf import RaySamplesPacked
    w = world
    ref = port.march_samples(w["o"], w["d"], w["te"], w["tx"], 1e-3, 64, 1 << 17, grid=w["gridnp"])
    rs = RaySamplesPacked(len(w["o"]), 1 << 17)
    rs.samples_pos.copy_(T(ref.pos, dev)); rs.samples_dirs.copy_(T(ref.dirs, dev))
    rs.samples_z.copy_(T(ref.z, dev)); rs.samples_dt.copy_(T(ref.dt, dev)); rs.samples_sdf.copy_(T(ref.sdf, dev))
    rs.samples_pos_4d.zero_()
    rs.ray_fixed_dt.copy_(T(ref.fixed_dt, dev)); rs.ray_start_end_idx.copy_(T(ref.start_end, dev))
    compare_packed(rs, port.compact(ref))


@pytest.mark.parametrize("jitter", [False, True])
def test_ray_sampler_fg_bg(port, world, dev, jitter):
    from permuto_sdf import RaySampler
    w = world
    o, d, te, tx = (T(w[k], dev) for k in ("o", "d", "te", "tx"))
    st = (RaySampler._rng.state, RaySampler._rng.inc)
    rs = RaySampler.compute_samples_fg(o, d, te, tx, 1e-2, 48, 0.5, torch.zeros(3, device=dev), jitter)
    ref = port.compact(port.march_samples(w["o"], w["d"], w["te"], w["tx"], 1e-2, 48, len(w["o"]) * 48, jitter=jitter, rng=st))
    compare_packed(rs, ref)
    # 32: the training configuration; 24: rays do not fill a workgroup evenly; 300: more samples per ray than a workgroup
    # has threads (the thread-per-ray kernel)
    for contract, per_ray in ((False, 32), (True, 32), (True, 24), (False, 300)):
        st = (RaySampler._rng.state, RaySampler._rng.inc)
        bg = RaySampler.compute_samples_bg(o, d, tx, per_ray, 0.5, [0, 0, 0], jitter, contract)
        rb = port.samples_bg(w["o"], w["d"], w["tx"], per_ray, 0.5, [0, 0, 0], jitter, contract, rng=st)
        assert bg.rays_have_equal_nr_of_samples and bg.fixed_nr_of_samples_per_ray == per_ray
        bits_equal(bg.samples_z, rb.z)
        bits_equal(bg.samples_dt, rb.dt)
        bits_equal(bg.samples_pos, rb.pos)
        bits_equal(bg.samples_dirs, rb.dirs)
        bits_equal(bg.ray_start_end_idx, rb.start_end)
        # 4-D point: direction uses the device reciprocal square root (1 ulp) -> 1e-6
        assert np.abs(bg.samples_pos_4d.cpu().numpy() - rb.pos4).max() < 1e-6


def test_first_hit(port, world, dev):
    w = world
    rs = w["grid"].compute_first_sample_start_of_occupied_regions(T(w["o"], dev), T(w["d"], dev), T(w["te"], dev), T(w["tx"], dev))
    ref = port.compact(port.first_hit_samples(w["o"], w["d"], w["te"], w["tx"], 1 << 21, w["gridnp"]))
    assert ref.total() > 100
    compare_packed(rs, ref)


def test_spherical_harmonics(port, dev):
    from permuto_sdf import PermutoSDF
    _, d = scene.make_rays(5000, seed=11)
    for deg in range(1, 8):
        out = PermutoSDF.spherical_harmonics(T(d, dev), deg)
        assert out.shape == (5000, deg * deg)
        bits_equal(out, port.spherical_harmonics(d, deg))
    assert abs(float(out[0, 0]) - 0.28209479) < 1e-7         # band-0 constant (known answer)
    with pytest.raises(ValueError):
        PermutoSDF.spherical_harmonics(T(d, dev), 8)


def test_random_rays_from_reel(port, dev):
    from permuto_sdf import PermutoSDF
    rng = np.random.default_rng(2)
    I, H, W = 4, 30, 40

    class Reel:
        pass
    reel = Reel()
    rgb = rng.uniform(size=(I, 3, H, W)).astype(np.float32)
    mask = (rng.uniform(size=(I, 1, H, W)) > 0.3).astype(np.float32)
    K = np.tile(np.array([[30, 0, 20], [0, 31, 15], [0, 0, 1]], np.float32), (I, 1, 1))
    tf = np.tile(np.eye(4, dtype=np.float32), (I, 1, 1))
    tf[:, :3, :3] = np.linalg.qr(rng.normal(size=(I, 3, 3)))[0]
    tf[:, :3, 3] = rng.normal(size=(I, 3))
    reel.rgb_reel, reel.mask_reel, reel.K_reel, reel.tf_world_cam_reel = (T(a, dev) for a in (rgb, mask, K, tf))
    for has_mask in (True, False):
        reel.has_mask = has_mask
        torch.manual_seed(5)
        o, d, gt, gm, img = PermutoSDF.random_rays_from_reel(reel, 3000)
        torch.manual_seed(5)   # the launcher draws pixel indices first, then image indices
        pix = torch.randint(0, H * W, (3000,), dtype=torch.int32, device=dev)
        img2 = torch.randint(0, I, (3000,), dtype=torch.int32, device=dev)
        assert torch.equal(img, img2)
        ro, rd, rgt, rgm = port.random_rays_from_reel(rgb, mask, K, tf, pix.cpu().numpy(), img.cpu().numpy(), has_mask)
        bits_equal(o, ro)
        bits_equal(gt, rgt)
        bits_equal(gm, rgm)
        assert np.abs(d.cpu().numpy() - rd).max() < 1e-6       # normalisation through the device rsqrt
        assert np.abs(d.norm(dim=1).cpu().numpy() - 1).max() < 1e-6


def test_against_committed_golden_vectors(dev):
    """HIP path vs tests/golden/ref_vectors.npz (outputs of the REFERENCE's own kernels, generated by
    tests/golden/make_golden.py where /root/reference exists).  Inputs are regenerated from the same seeds."""
    import os
    from permuto_sdf import OccupancyGrid, PermutoSDF, RaySampler, Sphere, VolumeRendering as VR
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_vectors.npz"))
    port = O.Oracle("port")
    n = 32
    occ = scene.shell_occupancy(port, n, seed=11)
    o, d = scene.make_rays(160, seed=21)
    to, td = T(o, dev), T(d, dev)
    p0, te, p1, tx, hit = Sphere(0.5, [0, 0, 0]).ray_intersection(to, td)
    for t, k in ((p0, "sph_p0"), (te, "sph_t0"), (p1, "sph_p1"), (tx, "sph_t1"), (hit, "sph_hit")):
        bits_equal(t, gold[k])
    g = OccupancyGrid(n, 1.0, [0, 0, 0])
    g.set_grid_occupancy(T(occ, dev))
    from permuto_sdf_amd.bridge import Pcg32
    for jit in (False, True):
        OccupancyGrid._rng = Pcg32()
        c = g.compute_samples_in_occupied_regions(to, td, te, tx, 2e-3, 48, jit).compact_to_valid_samples()
        k = "march%d_" % jit
        bits_equal(c.ray_start_end_idx, gold[k + "se"])
        nn = gold[k + "se"][:, 1].max()
        bits_equal(c.samples_z, gold[k + "z"][:nn])
        bits_equal(c.samples_pos, gold[k + "pos"][:nn])
        bits_equal(c.samples_dt, gold[k + "dt"][:nn])
        bits_equal(c.ray_fixed_dt, gold[k + "fdt"])
    RaySampler._rng = Pcg32()
    bg = RaySampler.compute_samples_bg(to, td, tx, 16, 0.5, [0, 0, 0], True, True)
    bits_equal(bg.samples_z, gold["bg_z"])
    bits_equal(bg.samples_pos, gold["bg_p3"])
    assert np.abs(bg.samples_pos_4d.cpu().numpy() - gold["bg_p4"]).max() < 1e-6
    bits_equal(PermutoSDF.spherical_harmonics(td, 5), gold["sh5"])
    bits_equal(PermutoSDF.spherical_harmonics(td, 7), gold["sh7"])
    pts = np.random.default_rng(31)
    fh = g.compute_first_sample_start_of_occupied_regions(to, td, te, tx).compact_to_valid_samples()
    bits_equal(fh.ray_start_end_idx, gold["fh_se"])
    bits_equal(fh.samples_z, gold["fh_z"][:gold["fh_se"][:, 1].max()])
    # compositing: rebuild the packed samples and the random tensors exactly as tests/golden/cases.py does
    OccupancyGrid._rng = Pcg32()
    c = g.compute_samples_in_occupied_regions(to, td, te, tx, 2e-3, 48, False).compact_to_valid_samples()
    M = c.samples_pos.shape[0]
    rng = np.random.default_rng(31)
    for _ in range(1):   # replay the generator draws of cases.run_all up to the compositing block
        rng.uniform(0, 2, n ** 3); rng.uniform(size=n ** 3); rng.normal(0, 0.05, n ** 3); rng.integers(0, n ** 3, 500)
        rng.uniform(-0.499, 0.499, (500, 3))
    rgb = rng.uniform(size=(M, 3)).astype(np.float32)
    sigma = rng.uniform(0, 60, (M, 1)).astype(np.float32)
    sdf = (scene.analytic_sdf(c.samples_pos.cpu().numpy()) + rng.normal(0, 2e-3, (M, 1))).astype(np.float32)
    pred, depth, bgT, w = VR.volume_render_nerf(c, T(rgb, dev), T(sigma, dev), tx, False)
    assert np.abs(pred.cpu().numpy() - gold["nerf_pred"]).max() < 2e-5
    assert np.abs(bgT.cpu().numpy() - gold["nerf_bg"]).max() < 2e-5
    alpha = VR.sdf2alpha(c, T(sdf, dev), 512.0, True, 1.0)
    assert np.abs(alpha.cpu().numpy() - gold["alpha"]).max() < 2e-6
    om = 1 - alpha.clamp(0, 1) + 1e-7
    Tr, bg2 = VR.cumprod_alpha2transmittance(c, om)
    assert np.abs(Tr.cpu().numpy() - gold["T"]).max() < 2e-6
    assert np.abs(VR.integrate_with_weights(c, T(rgb, dev), alpha.clamp(0, 1) * Tr).cpu().numpy() - gold["integ"]).max() < 2e-5


# ---------------------------------------------------------------------------------------------- coarse occupancy mask
def test_coarse_mask_words_and_marches_unchanged(world, dev):
    """The coarse mask (one bit per 8x8x8 Morton block, csrc/sampling.hip struct Occ) is what it says, and the four DDA
    entry points return the SAME bits with and without it (they are compared with the oracle, mask on, elsewhere in
    this file)."""
    from permuto_sdf import OccupancyGrid
    from permuto_sdf_amd import _lib as L
    g = world["grid"]
    n = world["n"]
    words = L.lib().psdf_occupancy_coarse_words(n)
    assert words == n ** 3 // (512 * 32)
    assert L.lib().psdf_occupancy_coarse_words(16) == 0          # too small for a word of blocks: no mask
    assert g._coarse(100) is None                                   # small batches march without it
    mask = g._coarse(1 << 20).cpu().numpy().view(np.uint32)
    blocks = world["occ"].reshape(-1, 512).any(axis=1).reshape(-1, 32)
    ref = (blocks.astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)
    assert np.array_equal(mask, ref)
    assert 0 < blocks.mean() < 0.6                                 # the shell leaves most blocks empty

    o, d, te, tx = (T(world[k], dev) for k in ("o", "d", "te", "tx"))

    def run():
        out = []
        for jitter in (False, True):
            OccupancyGrid._rng.__init__()
            rs = g.compute_samples_in_occupied_regions(o, d, te, tx, 1e-3, 64, jitter)
            m = int(rs.cur_nr_samples)      # the pool beyond the samples is uninitialised memory
            out += [rs.samples_z[:m].clone(), rs.samples_pos[:m].clone(), rs.samples_dt[:m].clone(), rs.ray_start_end_idx.clone()]
        rs = g.compute_first_sample_start_of_occupied_regions(o, d, te, tx)
        m = int(rs.cur_nr_samples)
        out += [rs.samples_pos[:m].clone(), rs.samples_z[:m].clone(), rs.ray_start_end_idx.clone()]
        p = (o + d * te).contiguous()
        p2, within = g.advance_sample_to_next_occupied_voxel(d, p)
        out += [p2.clone(), within.clone()]
        return out

    keep = OccupancyGrid.COARSE_MIN_RAYS
    try:
        OccupancyGrid.COARSE_MIN_RAYS = 0                           # the 2000 rays of this world march with the mask ...
        a = run()
        OccupancyGrid.use_coarse_mask = False                       # ... and without
        b = run()
    finally:
        OccupancyGrid.use_coarse_mask = True
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x.shape == y.shape and torch.equal(x, y)

    # the mask follows the occupancy: a grid edited in place is re-read on the next call
    occ = g.get_grid_occupancy()
    saved = occ.clone()
    try:
        occ.zero_()
        rs = g.compute_samples_in_occupied_regions(o, d, te, tx, 1e-3, 64, False)
        assert int(rs.cur_nr_samples) == 0
        occ.copy_(saved)
        rs = g.compute_samples_in_occupied_regions(o, d, te, tx, 1e-3, 64, False)
        assert torch.equal(rs.samples_z[:int(rs.cur_nr_samples)], a[0])
    finally:
        occ.copy_(saved)
        OccupancyGrid.COARSE_MIN_RAYS = keep


@pytest.mark.parametrize("n,R,per_ray,min_dist", [(256, 727, 64, 1e-4), (256, 20000, 64, 1e-4), (512, 3000, 128, 1e-4),
                                                   (128, 5000, 32, 5e-3), (1024, 600, 64, 1e-4)])
@pytest.mark.parametrize("jitter", [False, True])
def test_quad_march_equals_thread_march(dev, n, R, per_ray, min_dist, jitter, monkeypatch):
    """march_quad_kernel == march_kernel (which the tests above hold against the oracle), bit for bit, where the quad form's
    special cases live: a training step's ray count; more rays than one round of workgroups; n = 512 and 1024, where a walk is
    longer than the 512-step record (several chunks, the second march then probes memory); a coarse grid with few long steps;
    rays that START outside the grid (coordinates beyond it: the walk ends at once) and axis-parallel directions (a zero
    component: safe_inverse)."""
    from permuto_sdf import OccupancyGrid, Sphere
    g = torch.Generator().manual_seed(n + R)
    grid = OccupancyGrid(n, 1.0, [0, 0, 0])
    pts = grid.compute_grid_points(False)
    r = pts.norm(dim=1)
    occ = ((r - 0.3).abs() < 0.03) | ((r - 0.12).abs() < 0.01) | (torch.rand(pts.shape[0], generator=g).to(dev) < 0.002)
    grid.set_grid_occupancy(occ.contiguous())
    o = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=1) * 1.5
    d = torch.nn.functional.normalize((torch.rand(R, 3, generator=g) - 0.5) * 0.7 - o, dim=1)
    d[:8] = torch.tensor([[0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, -1.0]]).repeat(2, 1)   # axis-parallel
    o[:8] = -1.5 * d[:8] + torch.tensor([0.013, -0.021, 0.007]) * (1 - d[:8].abs())
    o, d = o.to(dev).contiguous(), d.to(dev).contiguous()
    _, te, _, tx, _ = Sphere(0.5, [0, 0, 0]).ray_intersection(o, d)
    te[8:16] = -3.0                   # these rays start far outside the grid: the walk must end at its first step
    out = {}
    for form in ("thread", "quad"):
        monkeypatch.setenv("PSDF_MARCH_FORM", form)
        st = (OccupancyGrid._rng.state, OccupancyGrid._rng.inc)
        rs = grid.compute_samples_in_occupied_regions(o, d, te, tx, min_dist, per_ray, jitter)
        assert _march_form() == (2 if form == "quad" else 1)
        OccupancyGrid._rng.state, OccupancyGrid._rng.inc = st          # the same jitter stream for both
        c = rs.compact_to_valid_samples()
        out[form] = [t.clone() for t in (c.ray_start_end_idx, c.samples_z, c.samples_pos, c.samples_dt, c.ray_fixed_dt,
                                         c.samples_dirs)]
    assert out["thread"][1].shape[0] > 3 * R
    for a, b in zip(out["thread"], out["quad"]):
        assert a.shape == b.shape
        if a.dtype == torch.float32:      # bit patterns (NaN spacing of rays without samples included)
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        else:
            assert torch.equal(a, b)
