"""CPU: the size of the gap between the contraction-free reference build the parity tests are bit-exact against and the same
headers built with floating-point contraction on (how nvcc builds them by default): tools/fma_contraction_gap.py on a small
scene.  The assertions are loose sanity bounds; the numbers that matter are in profiles/r05_fma_contraction_gap.json (20 000
rays) and quoted in README.md next to "bit-exact".  Needs /root/reference (like tests/test_oracle_vs_ref.py)."""
import importlib.util
import os

import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not (os.path.isdir("/root/reference/kernels") and O.build(ref=True) and O.have_ref()),
                                reason="reference CPU build not available")


def test_contraction_moves_few_rays_and_voxels(capsys):
    spec = importlib.util.spec_from_file_location("fma_gap", os.path.join(ROOT, "tools", "fma_contraction_gap.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import sys
    argv, sys.argv = sys.argv, ["fma_contraction_gap.py", "--rays", "1500"]
    try:
        res = mod.main()
    finally:
        sys.argv = argv
    capsys.readouterr()
    for jit in (False, True):
        a16 = res["a16 compute_samples_in_occupied_regions (jitter %s)" % jit]
        assert a16["rays_with_other_sample_count"] <= 0.01 * a16["rays"], a16
        assert a16["samples_in_another_voxel"] <= 0.002 * a16["samples_compared"], a16
        assert a16["depths_bit_identical"] >= 0.5 * a16["samples_compared"], a16
    assert res["a17 check_occupancy"]["answers_differ"] == 0
    assert res["a17 first-hit samples"]["rays_with_other_sample_count"] == 0
    assert res["a23 compute_cdf"]["max_abs_diff"] <= 1e-6
    assert res["a23 combine_uniform_samples_with_imp"]["rays_with_other_sample_count"] == 0
