"""Seeded golden cases shared by make_golden.py (runs them on the reference build) and test_oracle_golden.py (runs
them on the C restatement and compares bit for bit)."""
import numpy as np

from tests import scene


def run_all(L):
    """L: an oracle.Oracle back-end.  Returns {name: array}."""
    out = {}
    n = 32
    occ = scene.shell_occupancy(L, n, seed=11)
    grid = (n, 1.0, [0, 0, 0], occ)
    o, d = scene.make_rays(160, seed=21)
    p0, te, p1, tx, hit = L.sphere_intersect(0.5, [0, 0, 0], o, d)
    out.update(sph_p0=p0, sph_t0=te, sph_p1=p1, sph_t1=tx, sph_hit=hit)
    out["morton"] = np.array([L.morton3D(x, y, z) for x, y, z in [(1, 0, 0), (0, 1, 0), (0, 0, 1), (5, 9, 1023)]], np.uint32)
    u, f, st = L.pcg32(8, 0)
    out.update(pcg_u=u, pcg_f=f, pcg_adv=np.array([L.pcg32(1, 1 << 32)[2]], np.uint64))
    out["grid_pts"] = L.grid_points(8, 1.0, [0.1, 0, -0.1])
    out["grid_pts_jit"] = L.grid_points(8, 1.0, [0.1, 0, -0.1], randomize=True)
    rng = np.random.default_rng(31)
    vals, oc = rng.uniform(0, 2, n ** 3).astype(np.float32), rng.uniform(size=n ** 3) > 0.5
    sdf = rng.normal(0, 0.05, n ** 3).astype(np.float32)
    idx = np.unique(rng.integers(0, n ** 3, 500)).astype(np.int32)
    v1, o1 = L.update_with_sdf(vals, oc, sdf, n, 1.0, 256.0, 1e-4)
    v2, o2 = L.update_with_sdf(vals, oc, sdf[:len(idx)], n, 1.0, 256.0, 1e-4, idx)
    v3, o3 = L.update_with_density(vals, oc, np.abs(sdf) * 30, 0.95, 0.5)
    out.update(upd_sdf_v=v1, upd_sdf_o=o1, upd_sdfr_v=v2, upd_sdfr_o=o2, upd_den_v=v3, upd_den_o=o3)
    pts = rng.uniform(-0.499, 0.499, (500, 3)).astype(np.float32)
    out["check_occ"] = L.check_occupancy(n, 1.0, [0, 0, 0], occ, pts)
    for jit in (False, True):
        s = L.march_samples(o, d, te, tx, 2e-3, 48, 1 << 14, grid=grid, jitter=jit)
        c = L.compact(s)
        k = "march%d_" % jit
        out.update({k + "se": c.start_end, k + "z": c.z, k + "dt": c.dt, k + "pos": c.pos, k + "fdt": c.fixed_dt,
                    k + "cur": s.cur.copy()})
    fg = L.compact(L.march_samples(o, d, te, tx, 1e-2, 32, 1 << 13))
    out.update(fg_se=fg.start_end, fg_z=fg.z, fg_dt=fg.dt)
    bg = L.samples_bg(o, d, tx, 16, 0.5, [0, 0, 0], True, True)
    out.update(bg_z=bg.z, bg_dt=bg.dt, bg_p3=bg.pos, bg_p4=bg.pos4)
    fh = L.compact(L.first_hit_samples(o, d, te, tx, 1 << 12, grid))
    out.update(fh_se=fh.start_end, fh_z=fh.z, fh_pos=fh.pos)
    start = np.clip(o + te * d, -0.49, 0.49).astype(np.float32)
    ap, aw = L.advance_samples(d, start, grid)
    out.update(adv_pos=ap, adv_within=aw)
    out["sh5"] = L.spherical_harmonics(d, 5)
    out["sh7"] = L.spherical_harmonics(d, 7)
    # compositing on the packed samples
    s = L.compact(L.march_samples(o, d, te, tx, 2e-3, 48, 1 << 14, grid=grid))
    M = s.total()
    rgb = rng.uniform(size=(M, 3)).astype(np.float32)
    sigma = rng.uniform(0, 60, (M, 1)).astype(np.float32)
    s.sdf = (scene.analytic_sdf(s.pos) + rng.normal(0, 2e-3, (M, 1))).astype(np.float32)
    s.has_sdf = True
    pred, depth, bgT, w = L.volume_render_nerf(s, rgb, sigma)
    out.update(nerf_pred=pred, nerf_depth=depth, nerf_bg=bgT, nerf_w=w)
    g1, g2 = L.volume_render_nerf_backward(s, np.ones_like(pred), np.ones_like(bgT), pred, bgT, rgb, sigma)
    out.update(nerf_grgb=g1, nerf_gsig=g2)
    alpha = L.sdf2alpha(s, s.sdf, 512.0, True, 1.0)
    om = (1 - np.clip(alpha, 0, 1) + 1e-7).astype(np.float32)
    T, bg2 = L.cumprod(s, om)
    wgt = (np.clip(alpha, 0, 1) * T).astype(np.float32)
    ws, wss = L.sum_over_each_ray(s, wgt)
    wn = (wgt / np.clip(wss, 1e-6, None)).astype(np.float32)
    cdf = L.compute_cdf(s, wn)
    out.update(alpha=alpha, T=T, bgT=bg2, wsum=ws, integ=L.integrate(s, rgb, wgt), cdf=cdf,
               cumsum_inv=L.cumsum(s, wgt, True), dt_exit=L.compute_dt(s, tx, True))
    gi = L.integrate_backward(s, np.ones((s.R, 3), np.float32), rgb, wgt, True) if L.kind == "port" else \
        L.integrate_backward(s, np.ones((s.R, 3), np.float32), rgb, wgt)
    out.update(integ_grgb=gi[0], integ_gw=gi[1])
    cs = L.cumsum(s, (wgt * T).astype(np.float32), True)
    out["cumprod_bwd"] = L.cumprod_backward(s, wgt, np.ones_like(bg2), om, T, bg2, cs)
    imp = L.importance_sample(s, o, d, cdf, 8, True)
    imp.sdf, imp.has_sdf = scene.analytic_sdf(imp.pos), True
    out.update(imp_z=imp.z, imp_pos=imp.pos)
    c = L.compact(L.combine(s, imp, o, d, tx))
    out.update(comb_se=c.start_end, comb_z=c.z, comb_dt=c.dt, comb_sdf=c.sdf)
    return out


def run_encoding():
    import torch
    from oracle import permuto_oracle as po
    out = {}
    for P in (3, 4):
        L_, T, F = 6, 2 ** 12, 2
        lat, sh = po.make_params(P, T, L_, F, seed=5, init_scale=1.0)
        g = torch.Generator().manual_seed(6)
        pts = torch.rand(64, P, generator=g) - 0.5
        sl = np.geomspace(1.0, 1e-3, L_)
        win = po.coarse2fine_window(0.7, L_)
        out["enc_p%d" % P] = po.encode(pts, lat, sl, sh, win, True, 1e-3).numpy()
        rem0, rank, bary = po.simplex(pts, sh[2], po.scale_factors(sl, P)[2])
        out["enc_rank_p%d" % P] = rank.numpy().astype(np.int32)
        out["enc_idx_p%d" % P] = po.vertex_indices(rem0, rank, T).numpy().astype(np.int32)
    return out
