"""Pins the C restatement oracle (oracle/psdf_oracle.c) against the REFERENCE's own kernel headers compiled for
the CPU (oracle/_ref, built from /root/reference by `make -C oracle ref`): every function, bit for bit.
Skipped where the reference build is not available (then tests/test_oracle_golden.py still pins the port against
the vectors that were generated from that build)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import scene

pytestmark = pytest.mark.skipif(not (O.build(ref=True) and O.have_ref()), reason="reference CPU build not available")


@pytest.fixture(scope="module")
def libs():
    return O.Oracle("port"), O.Oracle("ref")


def eq(a, b):
    assert a.shape == b.shape and a.dtype == b.dtype
    if a.dtype == np.float32:
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), float(np.abs(a - b).max())
    else:
        assert np.array_equal(a, b)


def eq_samples(a, b):
    for k in ("pos", "dirs", "z", "dt", "fixed_dt", "start_end", "cur"):
        eq(getattr(a, k), getattr(b, k))


@pytest.fixture(scope="module")
def world(libs):
    port, ref = libs
    n = 64
    occ = scene.shell_occupancy(port, n)
    o, d = scene.make_rays(600, seed=3)
    _, te, _, tx, hit = port.sphere_intersect(0.5, [0, 0, 0], o, d)
    return dict(n=n, grid=(n, 1.0, [0, 0, 0], occ), o=o, d=d, te=te, tx=tx)


def test_scalars(libs):
    port, ref = libs
    for x, y, z in [(1, 0, 0), (0, 1, 0), (0, 0, 1), (255, 17, 93), (1023, 1023, 1023)]:
        assert port.morton3D(x, y, z) == ref.morton3D(x, y, z)
    for v in (0, 1, 7, 0x12345678, 0x3FFFFFFF):
        assert port.morton3D_invert(v) == ref.morton3D_invert(v)
    for adv in (0, 5, 1 << 32):
        for a, b in zip(port.pcg32(16, adv), ref.pcg32(16, adv)):
            assert np.array_equal(a, b)


def test_grid_points_and_updates(libs):
    port, ref = libs
    n = 32
    eq(port.grid_points(n, 1.0, [0.1, -0.2, 0.05]), ref.grid_points(n, 1.0, [0.1, -0.2, 0.05]))
    eq(port.grid_points(n, 2.0, [0, 0, 0], randomize=True), ref.grid_points(n, 2.0, [0, 0, 0], randomize=True))
    idx = np.random.default_rng(0).integers(0, n ** 3, 5000).astype(np.int32)
    eq(port.grid_points(n, 1.0, [0, 0, 0], idx, True), ref.grid_points(n, 1.0, [0, 0, 0], idx, True))
    rng = np.random.default_rng(1)
    vals, occ = rng.uniform(0, 2, n ** 3).astype(np.float32), rng.uniform(size=n ** 3) > 0.5
    dens = rng.uniform(0, 3, n ** 3).astype(np.float32)
    for a, b in zip(port.update_with_density(vals, occ, dens, 0.95, 0.5), ref.update_with_density(vals, occ, dens, 0.95, 0.5)):
        eq(a, b)
    uidx = np.unique(idx)
    for a, b in zip(port.update_with_density(vals, occ, dens[:len(uidx)], 0.9, 0.7, uidx),
                    ref.update_with_density(vals, occ, dens[:len(uidx)], 0.9, 0.7, uidx)):
        eq(a, b)
    sdf = rng.normal(0, 0.05, n ** 3).astype(np.float32)
    for inv_s in (64.0, 512.0):
        for a, b in zip(port.update_with_sdf(vals, occ, sdf, n, 1.0, inv_s, 1e-4), ref.update_with_sdf(vals, occ, sdf, n, 1.0, inv_s, 1e-4)):
            eq(a, b)
        for a, b in zip(port.update_with_sdf(vals, occ, sdf[:len(uidx)], n, 1.0, inv_s, 1e-4, uidx),
                        ref.update_with_sdf(vals, occ, sdf[:len(uidx)], n, 1.0, inv_s, 1e-4, uidx)):
            eq(a, b)


def test_check_occupancy_and_advance(libs, world):
    port, ref = libs
    pts = np.random.default_rng(5).uniform(-0.499, 0.499, (4000, 3)).astype(np.float32)
    eq(port.check_occupancy(*world["grid"][:3], world["grid"][3], pts), ref.check_occupancy(*world["grid"][:3], world["grid"][3], pts))
    dirs = world["d"][:400]
    start = (world["o"][:400] + world["te"][:400] * dirs).astype(np.float32)
    start = np.clip(start, -0.49, 0.49).astype(np.float32)
    for a, b in zip(port.advance_samples(dirs, start, world["grid"]), ref.advance_samples(dirs, start, world["grid"])):
        eq(a, b)


@pytest.mark.parametrize("jitter", [False, True])
def test_march_and_compact(libs, world, jitter):
    port, ref = libs
    w = world
    a = port.march_samples(w["o"], w["d"], w["te"], w["tx"], 1e-3, 64, 1 << 16, grid=w["grid"], jitter=jitter)
    b = ref.march_samples(w["o"], w["d"], w["te"], w["tx"], 1e-3, 64, 1 << 16, grid=w["grid"], jitter=jitter)
    eq_samples(a, b)
    assert a.total() > 1000
    ca, cb = port.compact(a), ref.compact(b)
    eq_samples(ca, cb)
    eq(port.per_sample_ray_idx(ca.start_end, ca.total()), ref.per_sample_ray_idx(cb.start_end, cb.total()))
    # pool overflow semantics
    a2 = port.march_samples(w["o"], w["d"], w["te"], w["tx"], 1e-3, 64, 3000, grid=w["grid"], jitter=jitter)
    b2 = ref.march_samples(w["o"], w["d"], w["te"], w["tx"], 1e-3, 64, 3000, grid=w["grid"], jitter=jitter)
    eq(a2.start_end, b2.start_end)
    eq(a2.cur, b2.cur)
    eq(a2.z[:3000], b2.z[:3000])


@pytest.mark.parametrize("jitter", [False, True])
def test_fg_bg_first_hit(libs, world, jitter):
    port, ref = libs
    w = world
    eq_samples(port.march_samples(w["o"], w["d"], w["te"], w["tx"], 1e-2, 48, 1 << 16, jitter=jitter),
               ref.march_samples(w["o"], w["d"], w["te"], w["tx"], 1e-2, 48, 1 << 16, jitter=jitter))
    for contract in (False, True):
        a = port.samples_bg(w["o"], w["d"], w["tx"], 32, 0.5, [0, 0, 0], jitter, contract)
        b = ref.samples_bg(w["o"], w["d"], w["tx"], 32, 0.5, [0, 0, 0], jitter, contract)
        eq_samples(a, b)
        eq(a.pos4, b.pos4)
    eq_samples(port.first_hit_samples(w["o"], w["d"], w["te"], w["tx"], 1 << 12, w["grid"]),
               ref.first_hit_samples(w["o"], w["d"], w["te"], w["tx"], 1 << 12, w["grid"]))


def test_sphere_sh_reel(libs):
    port, ref = libs
    o, d = scene.make_rays(3000, seed=9, jitter_target=0.8)
    for a, b in zip(port.sphere_intersect(0.5, [0.05, -0.02, 0.01], o, d), ref.sphere_intersect(0.5, [0.05, -0.02, 0.01], o, d)):
        eq(a, b)
    rng = np.random.default_rng(2)
    phi, ct, u = rng.uniform(0, 6.28, 2000), rng.uniform(-1, 1, 2000), rng.uniform(0, 1, 2000)
    eq(port.rand_points_inside(0.5, phi, ct, u), ref.rand_points_inside(0.5, phi, ct, u))
    for deg in range(1, 8):
        eq(port.spherical_harmonics(d, deg), ref.spherical_harmonics(d, deg))
    I, H, W = 3, 12, 20
    rgb, mask = rng.uniform(size=(I, 3, H, W)).astype(np.float32), (rng.uniform(size=(I, 1, H, W)) > 0.3).astype(np.float32)
    K = np.tile(np.array([[30, 0, 10], [0, 31, 6], [0, 0, 1]], np.float32), (I, 1, 1))
    tf = np.tile(np.eye(4, dtype=np.float32), (I, 1, 1))
    tf[:, :3, :3] = np.linalg.qr(rng.normal(size=(I, 3, 3)))[0]
    tf[:, :3, 3] = rng.normal(size=(I, 3))
    pix, img = rng.integers(0, H * W, 500), rng.integers(0, I, 500)
    for hm in (True, False):
        for a, b in zip(port.random_rays_from_reel(rgb, mask, K, tf, pix, img, hm), ref.random_rays_from_reel(rgb, mask, K, tf, pix, img, hm)):
            eq(a, b)


def test_volume_rendering(libs, world):
    port, ref = libs
    w = world
    s = port.compact(port.march_samples(w["o"], w["d"], w["te"], w["tx"], 2e-3, 64, 1 << 16, grid=w["grid"]))
    M = s.total()
    rng = np.random.default_rng(4)
    rgb = rng.uniform(size=(M, 3)).astype(np.float32)
    sigma = rng.uniform(0, 60, (M, 1)).astype(np.float32)
    sdf = scene.analytic_sdf(s.pos) + rng.normal(0, 2e-3, (M, 1)).astype(np.float32)
    s.sdf, s.has_sdf = sdf.copy(), True
    for res in (lambda L: L.volume_render_nerf(s, rgb, sigma), lambda L: L.compute_dt(s, w["tx"], True),
                lambda L: L.compute_dt(s, w["tx"], False)):
        a, b = res(port), res(ref)
        for x, y in zip(a if isinstance(a, tuple) else (a,), b if isinstance(b, tuple) else (b,)):
            eq(x, y)
    pred, depth, bg, wts = port.volume_render_nerf(s, rgb, sigma)
    gp, gb = rng.normal(size=pred.shape).astype(np.float32), rng.normal(size=bg.shape).astype(np.float32)
    for x, y in zip(port.volume_render_nerf_backward(s, gp, gb, pred, bg, rgb, sigma), ref.volume_render_nerf_backward(s, gp, gb, pred, bg, rgb, sigma)):
        eq(x, y)
    for dyn, inv_s, mult in ((True, 512.0, 1.0), (True, 512.0, 2.0), (False, 300.0, 1.0)):
        eq(port.sdf2alpha(s, sdf, inv_s, dyn, mult), ref.sdf2alpha(s, sdf, inv_s, dyn, mult))
    alpha = np.clip(port.sdf2alpha(s, sdf, 512.0, True, 1.0), 0, 1)
    one_minus = (1 - alpha + 1e-7).astype(np.float32)
    for x, y in zip(port.cumprod(s, one_minus), ref.cumprod(s, one_minus)):
        eq(x, y)
    T, bgT = port.cumprod(s, one_minus)
    wgt = (alpha * T).astype(np.float32)
    for C in (1, 2, 3, 32):
        v = rng.normal(size=(M, C)).astype(np.float32)
        for x, y in zip(port.sum_over_each_ray(s, v), ref.sum_over_each_ray(s, v)):
            eq(x, y)
        if C <= 3:
            g1, g2 = rng.normal(size=(s.R, C)).astype(np.float32), rng.normal(size=(M, C)).astype(np.float32)
            eq(port.sum_over_each_ray_backward(s, g1, g2, v), ref.sum_over_each_ray_backward(s, g1, g2, v))
    eq(port.integrate(s, rgb, wgt), ref.integrate(s, rgb, wgt))
    for x, y in zip(port.integrate_backward(s, gp, rgb, wgt, compat=True), ref.integrate_backward(s, gp, rgb, wgt)):
        eq(x, y)
    for inv in (False, True):
        eq(port.cumsum(s, wgt, inv), ref.cumsum(s, wgt, inv))
    gT = rng.normal(size=T.shape).astype(np.float32)
    cs = port.cumsum(s, (gT * T).astype(np.float32), True)
    eq(port.cumprod_backward(s, gT, gb, one_minus, T, bgT, cs), ref.cumprod_backward(s, gT, gb, one_minus, T, bgT, cs))
    wsum, wsum_s = port.sum_over_each_ray(s, wgt)
    wn = (wgt / np.clip(wsum_s, 1e-6, None)).astype(np.float32)
    eq(port.compute_cdf(s, wn), ref.compute_cdf(s, wn))
    cdf = port.compute_cdf(s, wn)
    for jitter in (False, True):
        ia, ib = port.importance_sample(s, w["o"], w["d"], cdf, 16, jitter), ref.importance_sample(s, w["o"], w["d"], cdf, 16, jitter)
        for k in ("pos", "dirs", "z"):
            eq(getattr(ia, k), getattr(ib, k))
        ia.sdf = scene.analytic_sdf(ia.pos)
        ia.has_sdf = True
        ca, cb = port.combine(s, ia, w["o"], w["d"], w["tx"]), ref.combine(s, ia, w["o"], w["d"], w["tx"])
        eq_samples(ca, cb)
        eq(ca.sdf, cb.sdf)
