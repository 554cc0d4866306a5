"""The frozen conventions of the permutohedral encoding (upstream source absent: PARITY UNPINNED) live in ONE header,
permuto_sdf_amd/csrc/encode_conventions.h.  These tests show that the three consumers read that one file:
the compiled library (psdf_encode_convention), the host mirror (permuto_sdf_amd/conventions.py) and the CPU oracle
(oracle/permuto_oracle.py) -- so flipping a convention is a one-line change they follow together.  The GPU half (both
concatenation layouts through the HIP kernels) is at the bottom, marked gpu."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import permuto_oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "permuto_sdf_amd", "csrc", "encode_conventions.h")
ORDER = ["PSDF_ENC_HASH_MULTIPLIER", "PSDF_ENC_RANK_TIE_RAISES_LATER", "PSDF_ENC_SCALE_SQRT_TERM", "PSDF_ENC_SCALE_INV_STDDEV",
         "PSDF_ENC_CONCAT_DEFAULT_LAYOUT"]


def test_library_host_and_oracle_read_the_same_header():
    from permuto_sdf_amd import conventions as CV
    from permuto_sdf_amd._lib import LIB_PATH
    assert os.path.samefile(CV.HEADER, HEADER) and os.path.samefile(po.CONVENTIONS_HEADER, HEADER)
    assert CV.C == po.CONV
    dll = ctypes.CDLL(LIB_PATH)
    dll.psdf_encode_convention.restype = ctypes.c_int64
    compiled = [int(dll.psdf_encode_convention(i)) for i in range(len(ORDER))]
    assert compiled == [int(CV.C[k]) for k in ORDER]          # what hipcc compiled in == what Python parsed
    assert dll.psdf_encode_convention(99) == -1
    # the kernels take every convention from the header, nothing is restated in the sources
    for f in ("encode_device.h", "encode.hip", "fused.hip"):
        src = open(os.path.join(ROOT, "permuto_sdf_amd", "csrc", f)).read()
        assert "2531011" not in src, f


# The frozen defaults, written out a SECOND time on purpose: product header and oracle read one file, so a typo there would move
# checker and product together.  These literals are the restated algorithm's own constants (SURVEY.md
# App. A.2 / A.3: multiplier 2531011, a tie raises the later rank, sqrt((i+1)(i+2)) scaling, no extra inverse-stddev factor,
# pseudo-level concatenation, lattice init 1e-5, shift scale 10) and must be edited by hand, with a reason, when a convention flips.
FROZEN_DEFAULTS = {"PSDF_ENC_HASH_MULTIPLIER": 2531011, "PSDF_ENC_RANK_TIE_RAISES_LATER": 1, "PSDF_ENC_SCALE_SQRT_TERM": 1,
                   "PSDF_ENC_SCALE_INV_STDDEV": 0, "PSDF_ENC_CONCAT_DEFAULT_LAYOUT": 1, "PSDF_ENC_LATTICE_INIT_SCALE": 1e-5,
                   "PSDF_ENC_RANDOM_SHIFT_SCALE": 10.0, "PSDF_ENC_CONCAT_NONE": 0, "PSDF_ENC_CONCAT_PSEUDO_LEVELS": 1,
                   "PSDF_ENC_CONCAT_APPEND": 2}


def test_the_header_still_holds_the_frozen_defaults():
    """literal copy of every default, independent of the header's text (and therefore of what product and oracle parse)"""
    assert po.parse_conventions(HEADER) == FROZEN_DEFAULTS
    # ... and the oracle's behaviour under them, on a hand-computed case that does not go through any parser: the hash of the
    # lattice key (1, -2, 3) with 2^18 rows is ((1 * m + (-2 mod 2^32)) * m + 3) * m mod 2^32 mod 2^18, m = 2531011
    m, h = 2531011, 0
    for k in (1, -2, 3):
        h = ((h + (k & 0xFFFFFFFF)) * m) & 0xFFFFFFFF
    # vertex_index_scalar(rem0, rank, remainder 0) hashes the key rem0 itself
    assert h % (1 << 18) == po.vertex_index_scalar([1, -2, 3, 0], [0, 0, 0, 0], 0, 3, 1 << 18)
    rem0 = torch.tensor([[1, -2, 3, 0]])
    assert h % (1 << 18) == int(po.vertex_indices(rem0, torch.zeros_like(rem0), 1 << 18)[0, 0])


def _flipped(tmp_path, **kv):
    txt = open(HEADER).read()
    for k, v in kv.items():
        import re
        txt, n = re.subn(r"(#define\s+%s\s+)[-+0-9.eE]+" % k, r"\g<1>%s" % v, txt)
        assert n == 1
    p = tmp_path / "encode_conventions.h"
    p.write_text(txt)
    return str(p)


def test_one_line_flip_of_the_concat_layout_is_followed_by_host_and_oracle(tmp_path, monkeypatch):
    from permuto_sdf_amd import conventions as CV
    assert CV.channels(3, 24, 2, CV.concat_mode(True)) == po.output_dims(3, 24, 2, True)
    flipped = _flipped(tmp_path, PSDF_ENC_CONCAT_DEFAULT_LAYOUT=2 if CV.C["PSDF_ENC_CONCAT_DEFAULT_LAYOUT"] == 1 else 1)
    host, oracle = CV.parse(flipped), po.parse_conventions(flipped)
    assert host == oracle and host["PSDF_ENC_CONCAT_DEFAULT_LAYOUT"] != CV.C["PSDF_ENC_CONCAT_DEFAULT_LAYOUT"]
    monkeypatch.setattr(po, "CONV", oracle)
    mode = CV.concat_mode(True, c=host)
    want = {1: 52, 2: 51}[mode]                                # models.py:154 feeds output_dims() into Linear(..., 32)
    assert CV.channels(3, 24, 2, mode) == want == po.output_dims(3, 24, 2, True)
    assert CV.channels(4, 24, 2, mode) == 52 == po.output_dims(4, 24, 2, True)      # P=4, F=2: both layouts coincide
    pts = torch.rand(50, 3) - 0.5
    lat, sh = po.make_params(3, 2 ** 10, 4, 2, seed=1, init_scale=1.0)
    out = po.encode(pts, lat, np.geomspace(1, 1e-2, 4), sh, torch.ones(4), True, 0.5)
    assert out.shape[1] == {1: 12, 2: 11}[mode]
    assert torch.equal(out[:, 8:11], pts * 0.5)


def test_oracle_follows_each_convention(monkeypatch):
    torch.manual_seed(0)
    pts = torch.rand(400, 3) - 0.5
    sl = np.geomspace(1.0, 1e-3, 6)
    lat, sh = po.make_params(3, 2 ** 10, 6, 2, seed=3, init_scale=1.0)
    win = torch.ones(6)
    base = po.encode(pts, lat, sl, sh, win)
    assert torch.equal(base[:40], po.encode_scalar(pts[:40], lat, sl, sh, win))
    for key, val in (("PSDF_ENC_HASH_MULTIPLIER", 2654435761), ("PSDF_ENC_SCALE_SQRT_TERM", 0), ("PSDF_ENC_SCALE_INV_STDDEV", 1)):
        conv = dict(po.CONV)
        conv[key] = val
        monkeypatch.setattr(po, "CONV", conv)
        alt = po.encode(pts, lat, sl, sh, win)
        assert not torch.equal(alt, base), key
        assert torch.equal(alt[:40], po.encode_scalar(pts[:40], lat, sl, sh, win)), key       # scalar == vectorised under the flip
    # the tie rule only matters for points whose residuals tie exactly: lattice-aligned coordinates
    conv = dict(po.CONV)
    conv["PSDF_ENC_RANK_TIE_RAISES_LATER"] = 0
    monkeypatch.setattr(po, "CONV", conv)
    aligned = torch.zeros(4, 3)
    r1 = po.simplex(aligned, torch.zeros(3), po.scale_factors([1.0], 3)[0])[1]
    monkeypatch.setattr(po, "CONV", dict(conv, PSDF_ENC_RANK_TIE_RAISES_LATER=1))
    r2 = po.simplex(aligned, torch.zeros(3), po.scale_factors([1.0], 3)[0])[1]
    assert not torch.equal(r1, r2)
    # both concatenation layouts: identical hashed channels and points, the padded one has the extra zero column
    a = po.encode(pts, lat, sl, sh, win, True, 1e-3, layout=1)
    b = po.encode(pts, lat, sl, sh, win, True, 1e-3, layout=2)
    assert a.shape[1] == 16 and b.shape[1] == 15 and torch.equal(a[:, :15], b) and torch.equal(a[:, 15], torch.zeros(400))
    assert torch.equal(b[:30], po.encode_scalar(pts[:30], lat, sl, sh, win, True, 1e-3, layout=2))


def test_host_scale_factor_follows_the_header():
    from permuto_sdf_amd import conventions as CV
    from permuto_sdf_amd.encoding import scale_factor_tensor
    sl = np.geomspace(1.0, 1e-4, 24)
    assert torch.equal(scale_factor_tensor(sl, 3), po.scale_factors(sl, 3))
    c = dict(CV.C, PSDF_ENC_SCALE_SQRT_TERM=0)
    assert CV.scale_term(1, 3, c) == 1.0 and abs(CV.scale_term(1, 3) - 6 ** 0.5) < 1e-12


# ------------------------------------------------------------------------------------------------------- GPU half
@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["pseudo_levels", "append"])
@pytest.mark.parametrize("P,L_", [(3, 24), (3, 16), (4, 24)])
def test_gpu_both_concat_layouts_match_oracle(dev, layout, P, L_):
    """forward, backward (lattice + positions), double backward and the fused encode->MLP launch in BOTH layouts"""
    from permuto_sdf_amd import PermutoEncoding
    from permuto_sdf_amd import conventions as CV
    from permuto_sdf_amd.mlp import FusedMLP
    lay = {"pseudo_levels": 1, "append": 2}[layout]
    T, N, F = 2 ** 14, 5000, 2
    torch.manual_seed(10 * P + L_ + lay)
    sl = np.geomspace(1.0, 1e-4, L_)
    enc = PermutoEncoding(P, T, L_, F, sl, concat_points=True, concat_points_scaling=0.37, init_scale=1.0, concat_layout=layout)
    C = enc.output_dims()
    assert C == po.output_dims(P, L_, F, True, lay) == {1: F * (L_ + (P + F - 1) // F), 2: F * L_ + P}[lay]
    pts = torch.rand(N, P) - 0.5
    win = po.coarse2fine_window(0.7, L_)
    lat = enc.lattice_values.detach().clone().requires_grad_(True)
    shifts = enc.random_shift_per_level.detach().clone()
    p_ref = pts.clone().requires_grad_(True)
    ref = po.encode(p_ref, lat, sl, shifts, win, True, 0.37, layout=lay)
    g = torch.randn(N, C)
    ref.backward(g)
    enc = enc.to(dev)
    p = pts.to(dev).requires_grad_(True)
    out = enc(p, win.to(dev))
    assert out.shape == ref.shape
    assert (out.detach().cpu() - ref.detach()).abs().max() <= 1e-6 * max(1.0, float(ref.abs().max()))
    out.backward(g.to(dev))
    assert (enc.lattice_values.grad.cpu() - lat.grad).abs().max() <= 1e-5 * lat.grad.abs().max() + 1e-7
    assert (p.grad.cpu() - p_ref.grad).abs().max() <= 2e-5 * p_ref.grad.abs().max()
    # double backward through the position gradient
    w1 = torch.randn(C, 3)
    u = torch.randn(N, P)

    def second_order(fn, lat_, pts_, w1_, u_):
        pts_ = pts_.clone().requires_grad_(True)
        sdf = torch.tanh(fn(pts_) @ w1_).sum(1, keepdim=True)
        grad = torch.autograd.grad(sdf, pts_, torch.ones_like(sdf), create_graph=True)[0]
        return torch.autograd.grad(((grad * u_).sum(1) ** 2).mean() + sdf.mean(), [lat_])[0]

    lat2 = lat.detach().clone().requires_grad_(True)
    g_ref = second_order(lambda x: po.encode(x, lat2, sl, shifts, win, True, 0.37, layout=lay), lat2, pts, w1, u)
    wd = win.to(dev)
    g_hip = second_order(lambda x: enc(x, wd), enc.lattice_values, pts.to(dev), w1.to(dev), u.to(dev))
    assert (g_hip.cpu() - g_ref).abs().max() <= 2e-5 * g_ref.abs().max()
    # a net on top: the fused evaluator takes the encoding's channel count as its input width (51 or 52), forward and backward
    mlp = FusedMLP([C, 32, 32, 32, 1]).to(dev)
    ref_mlp = torch.nn.Sequential(torch.nn.Linear(C, 32), torch.nn.GELU(), torch.nn.Linear(32, 32), torch.nn.GELU(),
                                  torch.nn.Linear(32, 32), torch.nn.GELU(), torch.nn.Linear(32, 1))
    for dst, src in zip([m for m in ref_mlp if isinstance(m, torch.nn.Linear)], mlp.layers):
        dst.weight.data.copy_(src.weight.detach().cpu())
        dst.bias.data.copy_(src.bias.detach().cpu())
    y_ref = ref_mlp(ref.detach())
    pd = pts.to(dev)
    y = mlp(enc(pd, wd))
    assert (y.detach().cpu() - y_ref.detach()).abs().max() <= 2e-5 * max(1.0, float(y_ref.abs().max()))
    y.sum().backward()
    y_ref.sum().backward()
    assert (mlp.layers[0].weight.grad.cpu() - ref_mlp[0].weight.grad).abs().max() <= 1e-4 * ref_mlp[0].weight.grad.abs().max()
    if P == 3:
        from permuto_sdf_amd.fused import encode_mlp_forward_raw, fused_supported
        from permuto_sdf_amd.mlp import pack_params
        dims = [C, 32, 32, 32, 1]
        assert fused_supported(enc.cfg, dims)
        packed = pack_params(dims, [l.weight for l in mlp.layers], [l.bias for l in mlp.layers])
        y1, feat = encode_mlp_forward_raw(enc.cfg, pd, enc.lattice_values.detach(), enc.scale_factor,
                                          enc.random_shift_per_level.detach(), wd, dims, packed, want_feat=True)
        assert tuple(feat.shape) == (C, N)
        assert (feat.t().cpu() - ref.detach()).abs().max() <= 1e-6 * max(1.0, float(ref.abs().max()))
        assert (y1.view(-1, 1).cpu() - y_ref.detach()).abs().max() <= 2e-5 * max(1.0, float(y_ref.abs().max()))


def test_runtime_setter_reports_and_resets():
    """psdf_encode_set_conventions / conventions.set: the device-side conventions are runtime values of the library (host only
    here: no launch), the host-side ones change what the Python mirror computes; reset() restores the header's defaults"""
    from permuto_sdf_amd import conventions as CV
    from permuto_sdf_amd._lib import LIB_PATH
    from permuto_sdf_amd.encoding import scale_factor_tensor
    dll = ctypes.CDLL(LIB_PATH)
    dll.psdf_encode_convention.restype = ctypes.c_int64
    try:
        CV.set(hash_multiplier=2654435761, rank_tie_raises_later=0, scale_inv_stddev=1)
        # NOTE: a second CDLL handle of the same path shares the library's globals (same mapping)
        assert int(dll.psdf_encode_convention(0)) == 2654435761 and int(dll.psdf_encode_convention(1)) == 0
        sl = np.geomspace(1.0, 1e-2, 4)
        a = scale_factor_tensor(sl, 3)
        CV.reset()
        b = scale_factor_tensor(sl, 3)
        assert torch.allclose(a, b * (4 * (2.0 / 3.0) ** 0.5))
        assert dll.psdf_encode_set_conventions(ctypes.c_uint32(0), ctypes.c_int(1)) == -1     # multiplier 0 is refused
        assert dll.psdf_encode_set_conventions(ctypes.c_uint32(5), ctypes.c_int(2)) == -1
    finally:
        CV.reset()
    assert [int(dll.psdf_encode_convention(i)) for i in range(len(ORDER))] == [int(CV.C[k]) for k in ORDER]
    with pytest.raises(KeyError):
        CV.set(push_to_library=False, no_such_convention=1)


@pytest.mark.gpu
@pytest.mark.parametrize("P", [3, 4])
def test_gpu_runtime_flip_of_hash_and_tie_rule_follows_the_oracle(dev, monkeypatch, P):
    """A flag flip, not a rebuild: with another hash multiplier and the other tie rule set at RUN TIME the HIP kernels (forward,
    lattice / position gradients, double backward, fused launch) follow the oracle evaluated under the same conventions --
    including points whose residuals tie exactly (position = -shift[l]), where the tie rule decides the position gradient."""
    from permuto_sdf_amd import PermutoEncoding
    from permuto_sdf_amd import conventions as CV
    L_, T, F, N = 8, 2 ** 12, 2, 3000
    sl = np.geomspace(1.0, 1e-3, L_)
    try:
        for flip in ({}, {"hash_multiplier": 2654435761, "rank_tie_raises_later": 0}):
            torch.manual_seed(40 + P)                                          # the same parameters and points under both
            CV.reset()
            CV.set(**flip)
            monkeypatch.setattr(po, "CONV", dict(CV.C))
            enc = PermutoEncoding(P, T, L_, F, sl, concat_points=True, concat_points_scaling=0.5, init_scale=1.0)
            pts = torch.rand(N, P) - 0.5
            pts[:L_] = -enc.random_shift_per_level.detach()                    # tie probes, one per level
            win = po.coarse2fine_window(0.8, L_)
            lat = enc.lattice_values.detach().clone().requires_grad_(True)
            sh = enc.random_shift_per_level.detach().clone()
            p_ref = pts.clone().requires_grad_(True)
            ref = po.encode(p_ref, lat, sl, sh, win, True, 0.5)
            g = torch.randn(N, ref.shape[1])
            u = torch.randn(N, P)
            (gp_ref,) = torch.autograd.grad(ref, p_ref, g, create_graph=True)
            (gl_ref,) = torch.autograd.grad(ref, lat, g, retain_graph=True)
            (dl_ref,) = torch.autograd.grad((gp_ref * u).sum(), [lat])
            enc = enc.to(dev)
            p = pts.to(dev).requires_grad_(True)
            out = enc(p, win.to(dev))
            assert (out.detach().cpu() - ref.detach()).abs().max() <= 1e-6 * max(1.0, float(ref.abs().max())), flip
            (gp,) = torch.autograd.grad(out, p, g.to(dev), create_graph=True)
            (gl,) = torch.autograd.grad(out, enc.lattice_values, g.to(dev), retain_graph=True)
            (dl,) = torch.autograd.grad((gp * u.to(dev)).sum(), [enc.lattice_values])
            assert (gp.detach().cpu() - gp_ref.detach()).abs().max() <= 2e-5 * gp_ref.abs().max(), flip
            assert (gp.detach().cpu()[:L_] - gp_ref.detach()[:L_]).abs().max() <= 2e-5 * gp_ref.abs().max(), flip   # the probes
            assert (gl.cpu() - gl_ref).abs().max() <= 1e-5 * gl_ref.abs().max() + 1e-7, flip
            assert (dl.cpu() - dl_ref).abs().max() <= 2e-5 * dl_ref.abs().max(), flip
            if not flip:
                base = out.detach().cpu().clone()
            else:   # the flip really changed what the kernels compute (another hash -> other table rows)
                assert not torch.allclose(out.detach().cpu()[:, :2 * L_], base[:, :2 * L_])
    finally:
        CV.reset()
