"""GPU: BASELINE config 4 against THE REFERENCE'S OWN STEP.  tools/reference_step_parity.py cuts the loss block of train()
(permuto_sdf_py/train_permuto_sdf.py:338-383, from `TIME_START("run_net")` to the mask loss) out of the unmodified reference
source, executes it with the reference's own `run_net` and model classes (torch.nn MLPs, the reference's autograd Functions) over
the UNFUSED drop-in operators, differentiates it with torch autograd -- and holds `ManualTrainer.step` and `Trainer.step`
against it on identical weights, rays and random streams: loss, every dense gradient, the three lattice gradients.  Four
states: `sphere` (the sphere-initialisation step, the reference's loss_sphere_init), `early` (curvature term on, coarse-to-fine
window partly open), `late` (curvature off, Lipschitz term on, every level open, sharp NeuS variance), `mask` (--with_mask: no
background, BCE on the weight sum).

What the numbers mean (measured on MI355X, profiles/r04_reference_step_parity.json):
  * `*_same_samples` (the reference's foreground samples handed to our trainers): everything agrees to a few 1e-5 in every state,
    curvature term included.  Bar: 1e-4 of the largest entry for dense gradients and for each lattice, 1e-5 for the loss -- or
    twice the reference's OWN rounding noise on those samples where that is larger, tensor by tensor
    (`reference_self_noise_same_samples`: the reference re-run SIX times with the hidden units of its SDF MLP re-numbered and
    the same samples forced, the largest deviation kept; in the `late` state its last SDF bias gradient, a sum of ~50 000
    cancelling NeuS terms, moves by ~1e-4 that way, and ours sit at the same distance).  A float64 arbiter of that one entry
    (`arbiter_last_sdf_bias`: the reference's own per-sample terms summed in float64) shows where the difference is NOT: the
    fp32 summation is exact to 1e-7; it is the terms that move with the last bits of the SDF values.
  * whole step, each side drawing its own samples: the importance samples follow each side's own SDF evaluations, and the
    finest lattice levels have cells of 1e-4 -- last-bit differences of the SDF (our fused evaluator vs torch.nn) move samples
    across cells.  The reference run against ITSELF with the hidden units of its SDF MLP re-numbered (`reference_self_noise`:
    the same function, another fp32 summation order) shows that noise: 2e-2 of the largest entry of the SDF lattice gradient in
    `late`, what our trainers show there in most runs -- and up to 4e-1 when one of the few samples that carry the late state's
    gradient is among those that moved.  Held tightly: the samplers' agreement (counts, ranges, >= 90 % of the depths
    bit-identical); the gradient bars of this comparison are per state and loose (WHOLE_STEP_BARS), the precise statement is the
    shared-sample one.
Runs in a subprocess (the reference module sets the default tensor type to CUDA at import).  Needs a reference checkout:
/root/reference or the git-ignored <repo>/_refcopy that travels to the GPU box; skipped with that reason otherwise."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reference():
    for c in (os.environ.get("PSDF_REFERENCE"), "/root/reference", os.path.join(ROOT, "_refcopy")):
        if c and os.path.isdir(os.path.join(c, "permuto_sdf_py")):
            return c
    return None


@pytest.fixture(scope="module")
def parity(tmp_path_factory):
    if _reference() is None:
        pytest.skip("no reference checkout: neither $PSDF_REFERENCE, /root/reference nor <repo>/_refcopy holds permuto_sdf_py")
    # PSDF_PARITY_OUT: keep the tool's JSON (profiles/r04_reference_step_parity.json is one of these)
    out = os.environ.get("PSDF_PARITY_OUT") or str(tmp_path_factory.mktemp("parity") / "reference_step_parity.json")
    env = dict(os.environ)
    env.pop("PSDF_FUSE_REFERENCE_MLPS", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "reference_step_parity.py"), "--out", out], capture_output=True,
                       text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.load(open(out))
    assert "cases" in d and set(d["cases"]) == {"early", "late", "mask", "sphere"}, d.keys()
    lo, hi = d["reference_block_lines"]
    assert 330 <= lo <= 345 and 380 <= hi <= 390, (lo, hi)        # the block IS train_permuto_sdf.py:338-383 (+- its comments)
    return d


def _report(case, names):
    for n in names:
        m = case[n]
        print("  %-22s loss_rel %.1e  dense %.1e  lattice max %.1e L2 %.1e  %s" % (
            n, m["loss_rel"], m["worst_dense"], m["worst_lattice"], m["worst_lattice_l2"], m.get("own_samples_vs_reference", "")))


def test_sphere_initialisation_step(parity):
    c = parity["cases"]["sphere"]
    _report(c, ("manual", "autograd"))
    for n in ("manual", "autograd"):
        m = c[n]
        assert not m["not_in_reference"] and all("missing" not in v for v in m["grads"].values())
        assert m["loss_rel"] <= 1e-5 and m["worst_dense"] <= 1e-4 and m["worst_lattice"] <= 1e-4, (n, m["loss_rel"], m["worst_dense"])


@pytest.mark.parametrize("mode", ["early", "late", "mask"])
def test_step_with_the_reference_samples(parity, mode):
    """only the step differs (the reference's foreground samples are handed to our trainers): the north_star bar, 1e-4 of the
    largest entry of every gradient -- or twice what the reference deviates from itself on the same samples -- or (dense tensors)
    within twice the diameter of the cloud of the reference's seven own evaluations, measured from its nearest member.  Only the
    SDF net's last bias in the `late` state ever needs more than the first clause; everything else sits below 1e-4"""
    c = parity["cases"][mode]
    # the reference against ITSELF on the same samples, hidden units of its SDF MLP re-numbered (the same function, another fp32
    # summation order): what rounding alone does to this step's gradients.  In the `late` state (NeuS variance exp(8): every
    # gradient is a difference of large terms) that is up to ~1e-4 for the SDF net's last bias, 1e-5 elsewhere.
    noise, rep = c["reference_self_noise_same_samples"], c["reference_repeat_noise"]
    arb = c["arbiter_last_sdf_bias"]
    print("  reference, hidden units re-numbered (six draws), same samples: dense %.1e (draws %s)  lattice max %.1e L2 %.1e   (simply "
          "run again: dense %.1e)" % (noise["worst_dense"], " ".join("%.1e" % v for v in noise["draws_worst_dense"]),
                                      noise["worst_lattice"], noise["worst_lattice_l2"], rep["worst_dense"]))
    print("  float64 arbiter of %s (sum of %d per-sample terms of the reference): reference %.1e, re-numbered reference %s, ours %.1e / %.1e"
          % (arb["tensor"], arb["terms_summed"], arb["reference_vs_f64"],
             " ".join("%.1e" % v for v in arb["reference_renumbered_same_samples_vs_f64"]),
             arb["manual_same_samples_vs_f64"], arb["autograd_same_samples_vs_f64"]))
    _report(c, ("manual_same_samples", "autograd_same_samples"))
    # the arbiter's finding (round 6): torch sums the last bias's ~50 000 terms to 1e-7 of their float64 sum -- the summation is
    # not where the sides part.  The TERMS move: re-numbering the hidden units changes the SDF values in their last bits, and
    # through the late state's NeuS variance (inv_s = e^8) that moves the cancelling per-sample terms by ~1e-4 of their sum.
    assert arb["reference_vs_f64"] <= 1e-5
    for n in ("manual_same_samples", "autograd_same_samples"):
        m = c[n]
        assert not m["not_in_reference"] and all("missing" not in v for v in m["grads"].values())
        assert m["nr_fg_samples"] == c["reference_terms"]["nr_fg_samples"]
        assert m["loss_rel"] <= 1e-5, (n, m["loss_rel"])
        # north_star bar 1e-4 of the largest entry -- or, tensor by tensor, twice the largest deviation the reference shows
        # from ITSELF over six re-numberings on these samples (measured in this very run; no constant from another day) -- or,
        # for the dense tensors, no further from the NEAREST of the reference's seven own evaluations of these gradients (the run
        # and its six re-numberings: equally valid fp32 roundings) than twice the largest distance between two of them.  (One run
        # is one draw of that cloud and so is the maximum of six distances from it: the late state's last SDF bias sits 1e-5 ..
        # 1.5e-4 from a given run while that run's six distances reach 2e-5 .. 1e-4 -- a bar of one draw failed 1 run in 8.)
        diam = noise.get("ensemble_diameter_by_tensor", {})
        near = m.get("nearest_reference_member_by_tensor", {})
        for k, v in m["grads"].items():
            if "max_rel" not in v:
                continue
            own = noise["by_tensor_max"].get(k, 0.0)
            bar = max(1e-4, (3.0 if "lattice" in k else 2.0) * own)
            inside = k in near and k in diam and near[k] <= 2.0 * diam[k]
            assert v["max_rel"] <= bar or inside, (n, k, v["max_rel"], own, near.get(k), diam.get(k))
            assert v["max_rel"] <= 1e-3, (n, k, v["max_rel"])       # whatever the cloud looks like: never beyond 1e-3
        assert m["worst_lattice_l2"] <= max(1e-4, 3 * noise["worst_lattice_l2"]), (n, m["worst_lattice_l2"])
        # the last bias against the float64 sum of the reference's terms: no further from it than the reference's re-numbered self
        # (-- or inside the cloud of the reference's own evaluations, as above: the re-numbered references scatter 2e-5 .. 2e-4
        # around that sum from run to run and box to box)
        ka = arb["tensor"]
        inside = ka in near and ka in diam and near[ka] <= 2.0 * diam[ka]
        assert arb[n + "_vs_f64"] <= max(1e-4, 2.0 * arb["reference_renumbered_same_samples_vs_f64_max"]) or inside, (n, arb, near.get(ka), diam.get(ka))


# whole-step bars (dense max, lattice max, lattice L2), per state: measured deviations in parentheses, over seven runs
# `late`: measured 7e-5 .. 2.5e-2 (dense), 2e-2 .. 4.4e-1 (lattice max), 7e-3 .. 7.5e-2 (lattice L2): only an O(1) sanity bar
# makes sense there; what is asserted tightly is the samplers' agreement and the loss, and the gradients are held by
# test_step_with_the_reference_samples (shared samples).
WHOLE_STEP_BARS = {"early": (1e-2, 5e-2, 2e-2),      # (2e-5 .. 2e-4, 1e-5 .. 2e-3, 3e-5 .. 1.4e-3)
                   "mask": (1e-2, 5e-2, 2e-2),       # (4e-6 .. 2e-5, 2e-5, 3e-5)
                   # O(1) sanity bars: a wrong sign or a missing loss term moves these by ~1; the measured noise
                   # (dense <= 2.5e-2, lattice max <= 4.4e-1, lattice L2 <= 7.5e-2) passes with room
                   "late": (0.25, 1.0, 0.5)}


@pytest.mark.parametrize("mode", ["early", "late", "mask"])
def test_whole_step_within_the_reference_own_noise(parity, mode):
    """Sampling included: every side draws its own importance samples from its own SDF evaluations.  What can be held tightly is
    the SAMPLERS' agreement: same per-ray counts and ranges (bit-exact), >= 90 % of the ~50 000 sample depths bit-identical, the
    rest within 1e-3.  The gradients that follow are as far apart as those few moved samples make them, and in the `late` state
    that is far: inv_s = e^8 = 2981, a sample that moves by 3e-5 along its ray changes the exponent of its NeuS opacity by 0.1,
    its weight by 10 %, and it is one of the handful of samples near the surface that carry the whole gradient -- measured over
    seven runs: dense gradients 7e-5 .. 2.5e-2 apart, the SDF lattice 2e-2 .. 4.4e-1 (max) / 7e-3 .. 7.5e-2 (L2), with the
    reference against ITSELF (hidden units of its SDF MLP re-numbered) showing 2e-2 .. 5e-2 / 7e-3 .. 1.1e-2 on the lattice.  So
    the `early` / `mask` gradient bars only catch a wrong step (a missing loss term moves them by O(1)), `late` has none, and the
    precise statement about the gradients is test_step_with_the_reference_samples."""
    c = parity["cases"][mode]
    noise = c["reference_self_noise"]
    print("  reference against itself (hidden units re-numbered): dense %.1e  lattice max %.1e L2 %.1e" % (
        noise["worst_dense"], noise["worst_lattice"], noise["worst_lattice_l2"]))
    _report(c, ("manual", "autograd"))
    bars = WHOLE_STEP_BARS[mode]
    for n in ("manual", "autograd"):
        m = c[n]
        assert m["nr_fg_samples"] == c["reference_terms"]["nr_fg_samples"]          # same rays, same counts (bit-exact samplers)
        st = m["own_samples_vs_reference"]
        assert st["same_count"] and st["same_ranges"], st
        assert st["identical"] >= 0.9 * st["of"] and st["max_abs_dz"] <= 1e-3, st
        assert m["loss_rel"] <= 1e-4, (n, m["loss_rel"])
        if bars is None:
            continue
        bar_dense, bar_lat, bar_l2 = bars
        assert m["worst_dense"] <= max(bar_dense, 3 * noise["worst_dense"]), (n, m["worst_dense"], noise["worst_dense"])
        assert m["worst_lattice"] <= max(bar_lat, 3 * noise["worst_lattice"]), (n, m["worst_lattice"], noise["worst_lattice"])
        assert m["worst_lattice_l2"] <= max(bar_l2, 3 * noise["worst_lattice_l2"]), (n, m["worst_lattice_l2"])


def test_manual_and_autograd_trainers_take_the_same_samples(parity):
    """both of our trainers run the same samplers on the same SDF evaluator: where they use the hand-written step their OWN
    samples are identical to each other's distance from the reference's"""
    for mode in ("early", "late"):
        a, b = (parity["cases"][mode][n]["own_samples_vs_reference"] for n in ("manual", "autograd"))
        assert a == b, (mode, a, b)
