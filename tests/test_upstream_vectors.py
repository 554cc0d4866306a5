"""PINNING THE ENCODING AGAINST UPSTREAM.  The permutohedral encoding's arithmetic lives in the un-vendored CUDA package
github.com/RaduAlexandru/permutohedral_encoding (models.py:20), absent here: parity of SURVEY.md rows a1-a3 is UNPINNED until
somebody who has that package runs

    python tools/dump_upstream_encoding_vectors.py     ->  tests/golden/upstream_encoding_vectors.npz

With the file present these tests compare the CPU oracle (and, `-m gpu`, the HIP kernels) with upstream's forward, lattice /
position gradients, double backward, parameter shapes / init statistics and Coarse2Fine; on a mismatch they search the
convention grid and NAME the combination that reproduces upstream (then: PSDF_ENC_CONVENTIONS=... or one edit of
permuto_sdf_amd/csrc/encode_conventions.h).  Without the file they XFAIL with the reason -- "unpinned" is reported, not skipped.

`test_pipeline_*` run the whole pipeline here against a stand-in package built on the oracle (tests/fake_upstream), once with
matching and once with deliberately different conventions: the dump script, the file format, the layout normalisation, the
comparison and the convention search are exercised end to end on the CPU.

Tolerance: 1e-4 relative to the largest entry (north_star), written at the assert."""
import itertools
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import permuto_oracle as po
from tests.golden.upstream_cases import CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VEC = os.path.join(ROOT, "tests", "golden", "upstream_encoding_vectors.npz")
UNPINNED = ("PARITY UNPINNED: tests/golden/upstream_encoding_vectors.npz is absent.  It can only be produced where the upstream "
            "CUDA package permutohedral_encoding is installed: python tools/dump_upstream_encoding_vectors.py")
TOL = 1e-4
GRID = {"PSDF_ENC_RANK_TIE_RAISES_LATER": (1, 0), "PSDF_ENC_SCALE_SQRT_TERM": (1, 0), "PSDF_ENC_SCALE_INV_STDDEV": (0, 1),
        "PSDF_ENC_CONCAT_DEFAULT_LAYOUT": (1, 2)}


# ------------------------------------------------------------------------------------------------- file access
def load(path):
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def to_LTF(arr, case):
    """upstream's raw parameter layout -> [L, T, F] (T = capacity is the unambiguous axis)"""
    L_, T, F = case["nr_levels"], case["capacity"], case["nr_feat"]
    a = np.asarray(arr)
    assert sorted(a.shape) == sorted((L_, T, F)), (a.shape, (L_, T, F))
    t_axis = list(a.shape).index(T)
    rest = [i for i in range(3) if i != t_axis]
    # of the two remaining axes the one of length F comes last; L == F only if nr_levels == nr_feat (not in the case list)
    l_axis, f_axis = (rest if a.shape[rest[1]] == F and a.shape[rest[0]] == L_ else rest[::-1])
    return np.ascontiguousarray(a.transpose(l_axis, t_axis, f_axis)), (l_axis, t_axis, f_axis)


def case_vectors(z, case):
    n = case["name"]
    v = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(n + "/")}
    v["lattice_LTF"], perm = to_LTF(v["lattice_values"], case)
    for k in ("grad_lattice", "dbl_grad_lattice"):
        if k in v:
            v[k + "_LTF"] = np.ascontiguousarray(v[k].transpose(*perm))
    v["shift_LP"] = v["random_shift_per_level"].reshape(case["nr_levels"], case["pos_dim"]) if "random_shift_per_level" in v \
        else np.zeros((case["nr_levels"], case["pos_dim"]), np.float32)
    return v


# ------------------------------------------------------------------------------------------------- oracle replay
class conventions:
    """with conventions(po, {...}): the oracle evaluates with these values of encode_conventions.h's #defines"""

    def __init__(self, conv):
        self.conv = conv

    def __enter__(self):
        self.old = po.CONV
        po.CONV = dict(po.CONV, **self.conv)

    def __exit__(self, *exc):
        po.CONV = self.old


def oracle_replay(case, v, conv, want=("out", "grads", "dbl")):
    with conventions(conv):
        pos = torch.from_numpy(v["positions"]).requires_grad_(True)
        lat = torch.from_numpy(v["lattice_LTF"]).requires_grad_(True)
        y = po.encode(pos, lat, v["scale_list"], torch.from_numpy(v["shift_LP"]), torch.from_numpy(v["window"]),
                      case["concat_points"], case["concat_points_scaling"])
        r = {"out": y.detach().numpy()}
        if y.shape[1] != v["out"].shape[1] or "grads" not in want:
            return r
        g_out = torch.from_numpy(v["grad_out"]).requires_grad_(True)
        (g_pos,) = torch.autograd.grad(y, pos, g_out, create_graph=True)
        (g_lat,) = torch.autograd.grad(y, lat, g_out, retain_graph=True)
        r.update(grad_positions=g_pos.detach().numpy(), grad_lattice_LTF=g_lat.detach().numpy())
        if "dbl" in want and "dbl_grad_lattice_LTF" in v:
            d_lat, d_g = torch.autograd.grad((g_pos * torch.from_numpy(v["dd_v"])).sum(), [lat, g_out])
            r.update(dbl_grad_lattice_LTF=d_lat.numpy(), dbl_grad_gout=d_g.numpy())
    return r


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.shape != b.shape:
        return float("inf")
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def errors(v, r):
    return {k: rel(r[k], v[k]) for k in ("out", "grad_positions", "grad_lattice_LTF", "dbl_grad_lattice_LTF", "dbl_grad_gout")
            if k in r and k in v}


def search(case, v):
    """every combination of the convention grid that reproduces upstream's forward AND first-order gradients for this case"""
    hits = []
    for combo in itertools.product(*GRID.values()):
        conv = dict(zip(GRID.keys(), combo))
        e = errors(v, oracle_replay(case, v, conv, want=("out", "grads")))
        if len(e) >= 3 and max(e.values()) < TOL:
            hits.append(conv)
    return hits


def check_file_against_oracle(path):
    """-> (report lines, {case: errors under the conventions in force}, {case: matching combinations})"""
    z, meta = load(path)
    lines, errs, hits = [], {}, {}
    for case in CASES:
        if case["name"] + "/out" not in z.files:
            continue
        v = case_vectors(z, case)
        e = errors(v, oracle_replay(case, v, {}))
        errs[case["name"]] = e
        ok = len(e) >= 3 and max(e.values()) < TOL
        lines.append("%-20s %s  %s" % (case["name"], "MATCH" if ok else "MISMATCH", " ".join("%s %.1e" % kv for kv in e.items())))
        if not ok:
            hits[case["name"]] = search(case, v)
            lines.append("    conventions that reproduce upstream: %s" % (hits[case["name"]] or
                         "none on the grid (tie, sqrt term, inv-std-dev, layout) -> hash multiplier / elevation differ"))
    return lines, errs, hits


# ------------------------------------------------------------------------------------------------- the real file
def test_oracle_matches_upstream_vectors():
    if not os.path.exists(VEC):
        pytest.xfail(UNPINNED)
    lines, errs, hits = check_file_against_oracle(VEC)
    print("\n".join(lines))
    bad = {n: e for n, e in errs.items() if not e or max(e.values()) >= TOL}
    assert not bad, "oracle != upstream under the conventions in force:\n" + "\n".join(lines)


def test_parameters_and_coarse2fine_match_upstream():
    """state_dict key / shape, init statistics of lattice_values and random_shift_per_level, output_dims, Coarse2Fine"""
    if not os.path.exists(VEC):
        pytest.xfail(UNPINNED)
    z, meta = load(VEC)
    for case in CASES:
        cm = meta["cases"].get(case["name"])
        if cm is None:
            continue
        assert any("lattice_values" in k for k in cm["state_dict"]), cm["state_dict"]          # models.py:408-420 relies on it
        v = case_vectors(z, case)
        assert cm["output_dims"] == v["out"].shape[1]
        assert cm["output_dims"] in (po.output_dims(case["pos_dim"], case["nr_levels"], case["nr_feat"], case["concat_points"], 1),
                                     po.output_dims(case["pos_dim"], case["nr_levels"], case["nr_feat"], case["concat_points"], 2))
        s = float(v["lattice_values"].std())
        assert 0.8 * po.CONV["PSDF_ENC_LATTICE_INIT_SCALE"] < s < 1.25 * po.CONV["PSDF_ENC_LATTICE_INIT_SCALE"], s
        if "random_shift_per_level" in v:
            s = float(v["random_shift_per_level"].std())
            assert 0.5 * po.CONV["PSDF_ENC_RANDOM_SHIFT_SCALE"] < s < 2.0 * po.CONV["PSDF_ENC_RANDOM_SHIFT_SCALE"], s
        if "scale_factor" in v:
            sf = np.asarray(v["scale_factor"]).reshape(case["nr_levels"], case["pos_dim"])
            assert rel(po.scale_factors(v["scale_list"], case["pos_dim"]).numpy(), sf) < 1e-6
    for t, w in zip(z["c2f_t"], z["c2f_window"]):
        assert np.abs(po.coarse2fine_window(float(t), 24).numpy() - w).max() < 1e-6, t


@pytest.mark.gpu
def test_hip_kernels_match_upstream_vectors(dev):
    if not os.path.exists(VEC):
        pytest.xfail(UNPINNED)
    from permuto_sdf_amd import PermutoEncoding
    z, meta = load(VEC)
    for case in CASES:
        if case["name"] + "/out" not in z.files:
            continue
        v = case_vectors(z, case)
        enc = PermutoEncoding(case["pos_dim"], case["capacity"], case["nr_levels"], case["nr_feat"], v["scale_list"],
                              concat_points=case["concat_points"], concat_points_scaling=case["concat_points_scaling"]).to(dev)
        with torch.no_grad():
            enc.lattice_values.copy_(torch.from_numpy(v["lattice_LTF"]))
            enc.random_shift_per_level.copy_(torch.from_numpy(v["shift_LP"]))
        pos = torch.from_numpy(v["positions"]).to(dev).requires_grad_(True)
        y = enc(pos, torch.from_numpy(v["window"]).to(dev))
        assert tuple(y.shape) == v["out"].shape, (case["name"], tuple(y.shape), v["out"].shape)
        g_out = torch.from_numpy(v["grad_out"]).to(dev).requires_grad_(True)
        (g_pos,) = torch.autograd.grad(y, pos, g_out, create_graph=True)
        (g_lat,) = torch.autograd.grad(y, enc.lattice_values, g_out, retain_graph=True)
        e = {"out": rel(y.detach().cpu().numpy(), v["out"]), "grad_positions": rel(g_pos.detach().cpu().numpy(), v["grad_positions"]),
             "grad_lattice": rel(g_lat.cpu().numpy(), v["grad_lattice_LTF"])}
        if "dbl_grad_lattice_LTF" in v:
            d_lat, d_g = torch.autograd.grad((g_pos * torch.from_numpy(v["dd_v"]).to(dev)).sum(), [enc.lattice_values, g_out])
            e["dbl_grad_lattice"] = rel(d_lat.cpu().numpy(), v["dbl_grad_lattice_LTF"])
            e["dbl_grad_gout"] = rel(d_g.cpu().numpy(), v["dbl_grad_gout"])
        print(case["name"], " ".join("%s %.1e" % kv for kv in e.items()))
        assert max(e.values()) < TOL, (case["name"], e)


# ------------------------------------------------------------------------------------------------- the pipeline, end to end
def _dump_with_fake_upstream(tmp_path, conv):
    out = str(tmp_path / "vectors.npz")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests", "fake_upstream"), ROOT]),
               FAKE_UPSTREAM_CONV=json.dumps(conv))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dump_upstream_encoding_vectors.py"), "--device", "cpu",
                        "--out", out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return out


def test_pipeline_with_matching_stand_in(tmp_path):
    path = _dump_with_fake_upstream(tmp_path, {})
    lines, errs, hits = check_file_against_oracle(path)
    assert len(errs) == len(CASES) and not hits, "\n".join(lines)
    for n, e in errs.items():
        assert set(e) == {"out", "grad_positions", "grad_lattice_LTF", "dbl_grad_lattice_LTF", "dbl_grad_gout"}, (n, e)
        assert max(e.values()) < 1e-6, (n, e)          # same arithmetic on both sides: far below the tolerance
    z, meta = load(path)
    assert meta["cases"]["p3_concat"]["tie_probes"] == 8 and "lattice_values" in meta["cases"]["p3_concat"]["lattice_name"]


def test_pipeline_names_the_conventions_of_a_differing_upstream(tmp_path):
    """a stand-in whose tie rule, inverse-std-dev term and concatenation layout differ from encode_conventions.h: the comparison
    must fail under the conventions in force and the search must name exactly the stand-in's combination"""
    other = {"PSDF_ENC_RANK_TIE_RAISES_LATER": 0, "PSDF_ENC_SCALE_INV_STDDEV": 1, "PSDF_ENC_CONCAT_DEFAULT_LAYOUT": 2}
    path = _dump_with_fake_upstream(tmp_path, other)
    lines, errs, hits = check_file_against_oracle(path)
    assert set(hits) == {c["name"] for c in CASES}, "\n".join(lines)
    for n, found in hits.items():
        assert found, (n, lines)
        for f in found:
            assert f["PSDF_ENC_SCALE_INV_STDDEV"] == 1 and f["PSDF_ENC_SCALE_SQRT_TERM"] == 1, (n, f)
        p3 = n.startswith("p3")
        concat = "plain" not in n
        if p3 and concat:            # the layout is observable only where the two layouts differ (P=3, F=2, concat)
            assert all(f["PSDF_ENC_CONCAT_DEFAULT_LAYOUT"] == 2 for f in found), (n, found)
        # the tie rule is observable through the position gradient at the tie probes
        assert all(f["PSDF_ENC_RANK_TIE_RAISES_LATER"] == 0 for f in found), (n, found)
