"""The case list shared by tools/dump_upstream_encoding_vectors.py (runs the cases on the real upstream
`permutohedral_encoding` CUDA package, wherever that exists) and tests/test_upstream_vectors.py (replays them on the oracle and
on the HIP kernels).  Pure numpy, imports nothing else: the dump script must run outside this repository's environment."""
import numpy as np

CASES = [
    dict(name="p3_concat", pos_dim=3, capacity=2 ** 12, nr_levels=8, nr_feat=2, concat_points=True, concat_points_scaling=1e-3,
         n=512, seed=101),
    dict(name="p3_plain", pos_dim=3, capacity=2 ** 12, nr_levels=8, nr_feat=2, concat_points=False, concat_points_scaling=1.0,
         n=512, seed=102),
    dict(name="p4_concat", pos_dim=4, capacity=2 ** 12, nr_levels=8, nr_feat=2, concat_points=True, concat_points_scaling=1.0,
         n=512, seed=103),
    dict(name="p4_plain", pos_dim=4, capacity=2 ** 12, nr_levels=8, nr_feat=2, concat_points=False, concat_points_scaling=1.0,
         n=512, seed=104),
    # the reference's own shape (models.py:143-149): 24 levels, geomspace(1, 1e-4), table 2^18 is too big to ship -> 2^14
    dict(name="p3_reference_shape", pos_dim=3, capacity=2 ** 14, nr_levels=24, nr_feat=2, concat_points=True,
         concat_points_scaling=1e-3, n=256, seed=105, scale_hi=1.0, scale_lo=1e-4),
]
C2F_TS = [0.0, 0.05, 0.3, 0.31, 0.5, 0.77, 1.0, 1.5]


def make_inputs(case):
    """deterministic inputs of a case (numpy RandomState: the same bits on every machine)"""
    rs = np.random.RandomState(case["seed"])
    P, L = case["pos_dim"], case["nr_levels"]
    n = case["n"]
    positions = (rs.rand(n, P).astype(np.float32) - 0.5)
    scale_list = np.geomspace(case.get("scale_hi", 1.0), case.get("scale_lo", 1e-3), L)
    window = (0.25 + 0.75 * rs.rand(L)).astype(np.float32)
    cmax = case["nr_feat"] * (L + (P + case["nr_feat"] - 1) // case["nr_feat"])     # the widest layout any convention produces
    grad_out_full = rs.randn(n, cmax).astype(np.float32)
    dd_v = rs.randn(n, P).astype(np.float32)
    return dict(positions=positions, scale_list=scale_list, window=window, grad_out_full=grad_out_full, dd_v=dd_v)
