"""How far is "bit-exact against the reference" from a real CUDA run?  (CPU tool; test infrastructure.)

The parity tests hold our kernels bit-exact against the reference's kernel headers compiled WITHOUT floating-point contraction
(oracle/_ref/libpsdf_ref.so, -ffp-contract=off; our HIP build uses the same setting).  nvcc builds the reference with
contraction ON by default (-fmad=true; /root/reference/CMakeLists.txt sets no flag against it), e.g.
`pos = ray_origin + t * ray_dir` (kernels/permuto_sdf/OccupancyGridGPU.cuh:563) becomes two fmas per component.  This tool
builds the SAME headers both ways (make -C oracle ref ref_fma) and runs the golden scenes through both: per operator, how many
rays change their sample count, how many samples land in another voxel, how far sample depths move.  g++ and nvcc do not
contract exactly the same expressions, so the numbers are an estimate of the size of the gap, not a prediction per ray.

python tools/fma_contraction_gap.py [--rays 20000] > profiles/r05_fma_contraction_gap.txt
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from tests import scene  # noqa: E402


def voxel_of(pos, n, extent=1.0):
    q = np.floor((pos / extent + 0.5) * n).astype(np.int64)
    return (q[:, 0] * n + q[:, 1]) * n + q[:, 2]


def compare_samples(a, b, n_grid=None):
    """a, b: oracle.Samples of the two builds for the same rays"""
    ca = a.start_end[:, 1] - a.start_end[:, 0]
    cb = b.start_end[:, 1] - b.start_end[:, 0]
    same = ca == cb
    out = {"rays": int(len(ca)), "rays_with_other_sample_count": int((~same).sum()),
           "samples": [int(ca.sum()), int(cb.sum())]}
    # rays with equal counts: compare sample by sample
    dz, dvox, nsmp, ident = 0.0, 0, 0, 0
    for r in np.nonzero(same)[0]:
        s0, s1 = a.start_end[r]
        t0 = b.start_end[r, 0]
        k = s1 - s0
        if k <= 0:
            continue
        za, zb = a.z[s0:s1, 0], b.z[t0:t0 + k, 0]
        dz = max(dz, float(np.abs(za - zb).max()))
        ident += int((za.view(np.uint32) == zb.view(np.uint32)).sum())
        nsmp += int(k)
        if n_grid:
            dvox += int((voxel_of(a.pos[s0:s1], n_grid) != voxel_of(b.pos[t0:t0 + k], n_grid)).sum())
    out.update(samples_compared=nsmp, depths_bit_identical=ident, max_abs_dz=dz)
    if n_grid:
        out["samples_in_another_voxel"] = dvox
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=20000)
    args = ap.parse_args()
    O.build(ref=True)
    ref, fma = O.Oracle("ref"), O.Oracle("ref_fma")
    n = 64
    occ = scene.shell_occupancy(ref, n)
    assert np.array_equal(occ, scene.shell_occupancy(fma, n))
    grid = (n, 1.0, [0, 0, 0], occ)
    o, d = scene.make_rays(args.rays, seed=3)
    res = {"scene": "tests/scene.py: %d rays from a radius-1.5 sphere at a spherical shell (r = 0.3 +- 0.05, 10 %% holes) in a %d^3 "
                    "occupancy grid" % (args.rays, n)}
    sa, sb = ref.sphere_intersect(0.5, [0, 0, 0], o, d), fma.sphere_intersect(0.5, [0, 0, 0], o, d)
    te, tx = sa[1], sa[3]
    res["a25 sphere intersection"] = {"rays": args.rays, "t_enter_bit_identical": int((sa[1].view(np.uint32) == sb[1].view(np.uint32)).sum()),
                                      "max_abs_dt": float(max(np.abs(sa[1] - sb[1]).max(), np.abs(sa[3] - sb[3]).max())),
                                      "hit_flags_differ": int((sa[4] != sb[4]).sum())}
    # the operators below get the SAME entry / exit depths (the contraction-free ones), so that each row shows one operator's own gap
    for jitter in (False, True):
        a = ref.march_samples(o, d, te, tx, 1e-3, 64, 1 << 22, grid=grid, jitter=jitter)
        b = fma.march_samples(o, d, te, tx, 1e-3, 64, 1 << 22, grid=grid, jitter=jitter)
        res["a16 compute_samples_in_occupied_regions (jitter %s)" % jitter] = compare_samples(a, b, n)
    a = ref.first_hit_samples(o, d, te, tx, 1 << 16, grid)
    b = fma.first_hit_samples(o, d, te, tx, 1 << 16, grid)
    res["a17 first-hit samples"] = compare_samples(a, b, n)
    pts = np.random.default_rng(5).uniform(-0.499, 0.499, (200000, 3)).astype(np.float32)
    ca, cb = ref.check_occupancy(*grid[:3], grid[3], pts), fma.check_occupancy(*grid[:3], grid[3], pts)
    res["a17 check_occupancy"] = {"points": len(pts), "answers_differ": int((np.asarray(ca) != np.asarray(cb)).sum())}
    for jitter in (False, True):
        a = ref.march_samples(o, d, te, tx, 1e-2, 48, 1 << 22, jitter=jitter)
        b = fma.march_samples(o, d, te, tx, 1e-2, 48, 1 << 22, jitter=jitter)
        res["a19 compute_samples_fg (jitter %s)" % jitter] = compare_samples(a, b)
        a = ref.samples_bg(o, d, tx, 32, 0.5, [0, 0, 0], jitter, True)
        b = fma.samples_bg(o, d, tx, 32, 0.5, [0, 0, 0], jitter, True)
        res["a18 compute_samples_bg (contracted space, jitter %s)" % jitter] = compare_samples(a, b)
    # a23: opacity / cdf / importance sampling / merge on the contraction-free samples
    s = ref.compact(ref.march_samples(o, d, te, tx, 2e-3, 64, 1 << 22, grid=grid))
    M = s.total()
    sdf = scene.analytic_sdf(s.pos) + np.random.default_rng(4).normal(0, 2e-3, (M, 1)).astype(np.float32)
    s.sdf, s.has_sdf = sdf.copy(), True
    aa, ab = ref.sdf2alpha(s, sdf, 512.0, True, 1.0), fma.sdf2alpha(s, sdf, 512.0, True, 1.0)
    res["a23 sdf2alpha"] = {"samples": int(M), "bit_identical": int((aa.view(np.uint32) == ab.view(np.uint32)).sum()),
                            "max_abs_diff": float(np.abs(aa - ab).max())}
    alpha = np.clip(aa, 0, 1)
    T, _ = ref.cumprod(s, (1 - alpha + 1e-7).astype(np.float32))
    wgt = (alpha * T).astype(np.float32)
    _, wsum_s = ref.sum_over_each_ray(s, wgt)
    wn = (wgt / np.clip(wsum_s, 1e-6, None)).astype(np.float32)
    cdfa, cdfb = ref.compute_cdf(s, wn), fma.compute_cdf(s, wn)
    res["a23 compute_cdf"] = {"samples": int(M), "bit_identical": int((cdfa.view(np.uint32) == cdfb.view(np.uint32)).sum()),
                              "max_abs_diff": float(np.abs(cdfa - cdfb).max())}
    for jitter in (False, True):
        ia, ib = ref.importance_sample(s, o, d, cdfa, 16, jitter), fma.importance_sample(s, o, d, cdfa, 16, jitter)
        res["a23 importance_sample (jitter %s)" % jitter] = {
            "samples": int(ia.z.shape[0]), "depths_bit_identical": int((ia.z.view(np.uint32) == ib.z.view(np.uint32)).sum()),
            "max_abs_dz": float(np.abs(ia.z - ib.z).max())}
    ia = ref.importance_sample(s, o, d, cdfa, 16, False)
    ia.sdf, ia.has_sdf = scene.analytic_sdf(ia.pos), True
    ca, cb = ref.combine(s, ia, o, d, tx), fma.combine(s, ia, o, d, tx)
    res["a23 combine_uniform_samples_with_imp"] = compare_samples(ca, cb)
    print(json.dumps(res, indent=1))
    return res


if __name__ == "__main__":
    main()
