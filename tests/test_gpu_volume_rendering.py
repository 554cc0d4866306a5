"""GPU parity of the compositing rows (VolumeRendering statics, fwd + bwd) vs the CPU oracle.
fp32 tolerances: 1e-5 relative to the per-tensor scale for scans / reductions (wave-tree vs serial summation
order), 1e-6 where only exp/libm differ; bit-exact for importance-sample indices / merge order."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import scene

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def close(t, a, rel=1e-5):
    t = t.detach().cpu().numpy()
    assert t.shape == a.shape, (t.shape, a.shape)
    err = np.abs(t - a).max() if a.size else 0.0
    assert err <= rel * max(1e-12, np.abs(a).max()) + 1e-12, (err, np.abs(a).max())


def bits_equal(t, a):
    t = t.detach().cpu().numpy()
    assert t.shape == a.shape
    assert np.array_equal(t.view(np.uint32), a.view(np.uint32)), float(np.abs(t - a).max())


@pytest.fixture(scope="module")
def port():
    return O.Oracle("port")


def to_packed(s, dev):
    from permuto_sdf import RaySamplesPacked
    n = len(s.z)
    rs = RaySamplesPacked(s.R, n)
    rs.samples_pos, rs.samples_dirs = T(s.pos, dev), T(s.dirs, dev)
    rs.samples_z, rs.samples_dt, rs.samples_sdf = T(s.z, dev), T(s.dt, dev), T(s.sdf, dev)
    rs.ray_fixed_dt, rs.ray_start_end_idx = T(s.fixed_dt, dev), T(s.start_end, dev)
    rs.rays_have_equal_nr_of_samples, rs.fixed_nr_of_samples_per_ray = s.equal, s.fixed
    rs.has_sdf = s.has_sdf
    rs.cur_nr_samples.fill_(n)
    rs._exact = True
    return rs


@pytest.fixture(scope="module", params=["packed", "equal160"])
def data(request, port, dev):
    """'packed': variable-length rays from the occupancy marcher (incl. empty rays);
    'equal160': equal-count rays of 160 samples (exercises the 64-lane chunk carry and the equal-count path)."""
    rng = np.random.default_rng(4)
    if request.param == "packed":
        occ = scene.shell_occupancy(port, 64)
        o, d = scene.make_rays(1500, seed=3)
        _, te, _, tx, _ = port.sphere_intersect(0.5, [0, 0, 0], o, d)
        s = port.compact(port.march_samples(o, d, te, tx, 2e-3, 100, 1 << 18, grid=(64, 1.0, [0, 0, 0], occ)))
    else:
        R, n = 300, 160
        o, d = scene.make_rays(R, seed=5)
        _, te, _, tx, _ = port.sphere_intersect(0.5, [0, 0, 0], o, d)
        s = O.Samples(R, R * n)
        s.equal, s.fixed = True, n
        zz = te + (tx - te) * (np.arange(n, dtype=np.float32)[None, :] + 0.5) / n
        s.z = zz.reshape(-1, 1).astype(np.float32)
        s.dt = np.full((R * n, 1), 1.0, np.float32) * ((tx - te) / n).repeat(n, 1).reshape(-1, 1).astype(np.float32)
        s.pos = (o[:, None, :] + zz[:, :, None] * d[:, None, :]).reshape(-1, 3).astype(np.float32)
        s.dirs = np.repeat(d, n, 0)
        s.fixed_dt = ((tx - te) / n).astype(np.float32)
        s.start_end = np.stack([np.arange(R) * n, np.arange(R) * n + n], 1).astype(np.int32)
    M = len(s.z)
    s.sdf = (scene.analytic_sdf(s.pos) + rng.normal(0, 2e-3, (M, 1))).astype(np.float32)
    s.has_sdf = True
    return dict(s=s, rs=to_packed(s, dev), o=o, d=d, tx=tx, M=M, rng=rng,
                rgb=rng.uniform(size=(M, 3)).astype(np.float32), sigma=rng.uniform(0, 60, (M, 1)).astype(np.float32))


def test_nerf_render_fwd_bwd(port, data, dev):
    from permuto_sdf import VolumeRendering as VR
    s, rs = data["s"], data["rs"]
    ref = port.volume_render_nerf(s, data["rgb"], data["sigma"])
    out = VR.volume_render_nerf(rs, T(data["rgb"], dev), T(data["sigma"], dev), T(data["tx"], dev), False)
    for a, b in zip(out, ref):
        close(a, b, 2e-5)
    # invariant: sum of weights + background transmittance == 1 (up to the T < 1e-4 early out)
    wsum, _ = port.sum_over_each_ray(s, ref[3])
    assert np.abs(wsum + ref[2] - 1)[s.counts() > 0].max() < 2e-4
    gp = data["rng"].normal(size=(s.R, 3)).astype(np.float32)
    gb = data["rng"].normal(size=(s.R, 1)).astype(np.float32)
    rg = port.volume_render_nerf_backward(s, gp, gb, ref[0], ref[2], data["rgb"], data["sigma"])
    og = VR.volume_render_nerf_backward(T(gp, dev), T(gb, dev), torch.zeros(data["M"], 1, device=dev), out[0], rs,
                                        T(data["rgb"], dev), T(data["sigma"], dev), T(data["tx"], dev), False, out[2])
    close(og[0], rg[0], 2e-5)
    close(og[1], rg[1], 5e-5)


def test_dt_alpha_transmittance_chain(port, data, dev):
    from permuto_sdf import VolumeRendering as VR
    s, rs = data["s"], data["rs"]
    for use in (True, False):
        bits_equal(VR.compute_dt(rs, T(data["tx"], dev), use), port.compute_dt(s, data["tx"], use))
    for dyn, inv_s, mult in ((True, 512.0, 1.0), (True, 512.0, 2.0), (False, 300.0, 1.0)):
        a = VR.sdf2alpha(rs, T(s.sdf, dev), inv_s, dyn, mult)
        assert np.abs(a.cpu().numpy() - port.sdf2alpha(s, s.sdf, inv_s, dyn, mult)).max() < 2e-6
    alpha = np.clip(port.sdf2alpha(s, s.sdf, 512.0, True, 1.0), 0, 1)
    om = (1 - alpha + 1e-7).astype(np.float32)
    Tr, bg = port.cumprod(s, om)
    oT, obg = VR.cumprod_alpha2transmittance(rs, T(om, dev))
    close(oT, Tr, 2e-6)
    close(obg, bg, 2e-6)
    if not s.equal:
        assert np.all(bg[s.counts() == 0] == 1.0)            # empty rays keep full background transmittance
    # known answer: constant factor a on a ray -> T_i = a^i
    w = (alpha * Tr).astype(np.float32)
    for C in (1, 2, 3, 32):
        v = data["rng"].normal(size=(data["M"], C)).astype(np.float32)
        r1, r2 = port.sum_over_each_ray(s, v)
        o1, o2 = VR.sum_over_each_ray(rs, T(v, dev))
        close(o1, r1, 2e-5)
        close(o2, r2, 2e-5)
        if C <= 3:
            g1 = data["rng"].normal(size=(s.R, C)).astype(np.float32)
            g2 = data["rng"].normal(size=(data["M"], C)).astype(np.float32)
            bits_equal(VR.sum_over_each_ray_backward(T(g1, dev), T(g2, dev), rs, T(v, dev)),
                       port.sum_over_each_ray_backward(s, g1, g2, v))
    with pytest.raises(ValueError):
        VR.sum_over_each_ray(rs, torch.zeros(data["M"], 5, device=dev))
    close(VR.integrate_with_weights(rs, T(data["rgb"], dev), T(w, dev)), port.integrate(s, data["rgb"], w), 2e-5)
    gp = data["rng"].normal(size=(s.R, 3)).astype(np.float32)
    for compat in (True, False):
        VR.reference_compat = compat
        og = VR.integrate_with_weights_backward(T(gp, dev), rs, T(data["rgb"], dev), T(w, dev), None)
        rg = port.integrate_backward(s, gp, data["rgb"], w, compat)
        bits_equal(og[0], rg[0])
        bits_equal(og[1], rg[1])
    VR.reference_compat = True
    for inv in (False, True):
        close(VR.cumsum_over_each_ray(rs, T(w, dev), inv), port.cumsum(s, w, inv), 2e-5)
    gT = data["rng"].normal(size=Tr.shape).astype(np.float32)
    gb = data["rng"].normal(size=bg.shape).astype(np.float32)
    cs = port.cumsum(s, (gT * Tr).astype(np.float32), True)
    bits_equal(VR.cumprod_alpha2transmittance_backward(T(gT, dev), T(gb, dev), rs, T(om, dev), T(Tr, dev), T(bg, dev), T(cs, dev)),
               port.cumprod_backward(s, gT, gb, om, Tr, bg, cs))


def test_constant_alpha_known_answer(dev):
    from permuto_sdf import RaySamplesPacked, VolumeRendering as VR
    R, n, a = 3, 150, 0.97
    rs = RaySamplesPacked(R, R * n)
    rs.rays_have_equal_nr_of_samples, rs.fixed_nr_of_samples_per_ray = True, n
    Tr, bg = VR.cumprod_alpha2transmittance(rs, torch.full((R * n, 1), a, device=dev))
    expect = a ** np.arange(n)
    assert np.abs(Tr.view(R, n)[1].cpu().numpy() - expect).max() < 1e-5
    assert abs(float(bg[2]) - a ** (n - 1)) < 1e-5


def test_autograd_functions_of_the_reference(port, data, dev):
    """The reference's four autograd.Functions (volume_rendering_funcs.py) are thin wrappers over these statics;
    check the fwd/bwd pairs are mutually consistent with a finite-difference directional derivative."""
    from permuto_sdf import VolumeRendering as VR
    s, rs = data["s"], data["rs"]
    M = data["M"]
    torch.manual_seed(0)
    a = (0.6 + 0.39 * torch.rand(M, 1, device=dev, dtype=torch.float32))
    gT = torch.randn(M, 1, device=dev)
    gb = torch.randn(s.R, 1, device=dev)

    def f(x):
        Tr, bg = VR.cumprod_alpha2transmittance(rs, x)
        return float((Tr.double() * gT.double()).sum() + (bg.double() * gb.double()).sum())
    Tr, bg = VR.cumprod_alpha2transmittance(rs, a)
    cs = VR.cumsum_over_each_ray(rs, gT * Tr, True)
    g = VR.cumprod_alpha2transmittance_backward(gT, gb, rs, a, Tr, bg, cs)
    v = torch.randn(M, 1, device=dev)
    eps = 1e-3
    fd = (f(a + eps * v) - f(a - eps * v)) / (2 * eps)
    an = float((g.double() * v.double()).sum())
    assert abs(fd - an) <= 2e-2 * max(1.0, abs(an)), (fd, an)


@pytest.mark.parametrize("jitter", [False, True])
def test_importance_sampling_and_merge(port, data, dev, jitter):
    from permuto_sdf import VolumeRendering as VR
    s, rs = data["s"], data["rs"]
    alpha = np.clip(port.sdf2alpha(s, s.sdf, 512.0, True, 1.0), 0, 1)
    Tr, _ = port.cumprod(s, (1 - alpha + 1e-7).astype(np.float32))
    w = (alpha * Tr).astype(np.float32)
    _, ws = port.sum_over_each_ray(s, w)
    wn = (w / np.clip(ws, 1e-6, None)).astype(np.float32)
    cdf = port.compute_cdf(s, wn)
    close(VR.compute_cdf(rs, T(wn, dev)), cdf, 2e-5)
    last = cdf[s.start_end[s.counts() > 0, 1] - 1]
    st = (VR._rng.state, VR._rng.inc)
    imp = VR.importance_sample(T(data["o"], dev), T(data["d"], dev), rs, T(cdf, dev), 16, jitter)
    ri = port.importance_sample(s, data["o"], data["d"], cdf, 16, jitter, rng=st)
    assert imp.rays_have_equal_nr_of_samples and imp.fixed_nr_of_samples_per_ray == 16
    bits_equal(imp.samples_z, ri.z)
    bits_equal(imp.samples_pos, ri.pos)
    bits_equal(imp.samples_dirs, ri.dirs)
    ri.sdf = scene.analytic_sdf(ri.pos)
    ri.has_sdf = True
    imp.set_sdf(T(ri.sdf, dev))
    comb = VR.combine_uniform_samples_with_imp(T(data["o"], dev), T(data["d"], dev), T(data["tx"], dev), rs, imp)
    rc = port.compact(port.combine(s, ri, data["o"], data["d"], data["tx"]))
    n = rc.total()
    assert comb.has_sdf and comb.compute_exact_nr_samples() == n
    c = comb.compact_to_valid_samples()
    bits_equal(c.ray_start_end_idx, rc.start_end)
    for name, arr in (("samples_pos", rc.pos), ("samples_dirs", rc.dirs), ("samples_z", rc.z), ("samples_dt", rc.dt),
                      ("samples_sdf", rc.sdf)):
        bits_equal(getattr(c, name), arr[:n])
    z = rc.z[:n, 0]
    for st_, en in rc.start_end[rc.counts() > 0][:200]:
        assert np.all(np.diff(z[st_:en]) >= 0)              # merged samples are sorted along each ray


@pytest.mark.parametrize("case", ["ties", "unsorted_importance", "unsorted_uniform"])
def test_merge_ties_and_unsorted_inputs(port, dev, case):
    """combine_uniform_samples_with_imp is a serial two-way merge per ray in the reference (importance sample first on ties);
    the kernel places every element by rank in parallel when both lists are sorted and walks the lists like the reference when
    they are not.  Both branches, and z values that collide exactly, against the oracle bit for bit."""
    from permuto_sdf import RaySamplesPacked, VolumeRendering as VR
    rng = np.random.default_rng(11)
    R, n_imp = 37, 16
    o, d = scene.make_rays(R, seed=9)
    _, te, _, tx, _ = port.sphere_intersect(0.5, [0, 0, 0], o, d)
    counts = rng.integers(0, 90, R)
    counts[:3] = (0, 1, 2)
    start = np.concatenate([[0], np.cumsum(counts)])
    M = int(start[-1])
    s = O.Samples(R, M)
    s.equal, s.fixed = False, 0
    s.start_end = np.stack([start[:-1], start[1:]], 1).astype(np.int32)
    grid = 64.0 if case == "ties" else 4096.0       # z on a coarse grid of values: many exact collisions between the two lists
    z = np.zeros((M, 1), np.float32)
    zi = np.zeros((R * n_imp, 1), np.float32)
    for r in range(R):
        a, b = float(te[r, 0]), float(tx[r, 0])
        zu = np.sort(np.round(rng.uniform(a, b, counts[r]) * grid) / grid).astype(np.float32)
        zr = np.sort(np.round(rng.uniform(a, b, n_imp) * grid) / grid).astype(np.float32)
        if case == "unsorted_uniform" and counts[r] > 4 and r % 2 == 0:
            zu[[1, 3]] = zu[[3, 1]]
        if case == "unsorted_importance" and r % 3 == 0:
            zr[[2, 9]] = zr[[9, 2]]
        z[start[r]:start[r + 1], 0] = zu
        zi[r * n_imp:(r + 1) * n_imp, 0] = zr
    ridx = np.repeat(np.arange(R), counts)
    s.z = z
    s.dt = np.full((M, 1), 1e-2, np.float32)
    s.pos = (o[ridx] + z * d[ridx]).astype(np.float32)
    s.dirs = d[ridx].astype(np.float32)
    s.fixed_dt = rng.uniform(5e-3, 2e-2, (R, 1)).astype(np.float32)
    s.sdf = rng.normal(0, 1, (M, 1)).astype(np.float32)
    s.has_sdf = True
    ri = O.Samples(R, R * n_imp)
    ri.equal, ri.fixed = True, n_imp
    iidx = np.repeat(np.arange(R), n_imp)
    ri.z, ri.pos, ri.dirs = zi, (o[iidx] + zi * d[iidx]).astype(np.float32), d[iidx].astype(np.float32)
    ri.sdf = rng.normal(0, 1, (R * n_imp, 1)).astype(np.float32)
    ri.has_sdf = True
    ri.start_end = np.stack([np.arange(R) * n_imp, np.arange(R) * n_imp + n_imp], 1).astype(np.int32)
    ri.dt = np.zeros((R * n_imp, 1), np.float32)
    ri.fixed_dt = np.zeros((R, 1), np.float32)
    if case == "ties":
        assert len(np.intersect1d(z, zi)) > 20
    rs = to_packed(s, dev)
    imp = RaySamplesPacked(R, R * n_imp)
    imp.rays_have_equal_nr_of_samples, imp.fixed_nr_of_samples_per_ray = True, n_imp
    imp.samples_z, imp.samples_pos, imp.samples_dirs = T(ri.z, dev), T(ri.pos, dev), T(ri.dirs, dev)
    imp.set_sdf(T(ri.sdf, dev))
    comb = VR.combine_uniform_samples_with_imp(T(o, dev), T(d, dev), T(tx, dev), rs, imp)
    rc = port.compact(port.combine(s, ri, o, d, tx))
    n = rc.total()
    assert n > 0 and comb.compute_exact_nr_samples() == n
    c = comb.compact_to_valid_samples()
    bits_equal(c.ray_start_end_idx, rc.start_end)
    for name, arr in (("samples_pos", rc.pos), ("samples_dirs", rc.dirs), ("samples_z", rc.z), ("samples_dt", rc.dt),
                      ("samples_sdf", rc.sdf)):
        bits_equal(getattr(c, name), arr[:n])
    bits_equal(c.ray_fixed_dt, rc.fixed_dt)


@pytest.mark.parametrize("mult", [1.0, 2.0])
def test_sdf_importance_cdf_equals_the_operator_chain(data, dev, mult):
    """VolumeRendering.sdf_importance_cdf (one launch; the trainers' sampling loop) == the nine launches of
    importance_sampling_sdf_model's operator chain (sdf_utils.py:403-417), bit for bit: rays of 0, 1, 2, 3 .. 160 samples,
    packed and equal-count containers, a NaN in the SDF."""
    from permuto_sdf import VolumeRendering as VR
    rs = data["rs"]
    sdf = rs.samples_sdf.clone()
    if data["M"] > 500:
        sdf[437] = float("nan")

    def chain():
        alpha = VR.sdf2alpha(rs, sdf, 512.0, True, mult).clip(0.0, 1.0)
        Tr, _ = VR.cumprod_alpha2transmittance(rs, 1 - alpha + 1e-7)
        w = alpha * Tr
        _, per_sample = VR.sum_over_each_ray(rs, w)
        return VR.compute_cdf(rs, w / torch.clamp(per_sample, min=1e-6))
    a, b = chain(), VR.sdf_importance_cdf(rs, sdf, 512.0, True, mult)
    assert a.shape == b.shape
    nan = torch.isnan(a)
    assert torch.equal(nan, torch.isnan(b))                  # the ray with the NaN is NaN in both (sign / payload bits are not compared)
    if data["M"] > 500:
        assert bool(nan.any()) and not bool(nan.all())
    assert torch.equal(a[~nan].view(torch.int32), b[~nan].view(torch.int32)), float((a[~nan] - b[~nan]).abs().max())
    ok = torch.isfinite(a)
    assert float(a[ok].max()) <= 1.0 + 1e-5 and float(a[ok].min()) >= 0.0
