"""GPU: the binning launch of the large-batch encode backward deals its resident round of workgroups over the levels by their
measured cost (csrc/encode.hip, LevelPlan / encode_balance).  The deal changes nothing but the schedule: the gradient equals
the equal-share launch's up to the order of float additions, and closed levels fall to the minimum share."""
import ctypes

import pytest
import torch

from permuto_sdf_amd import _lib as L
from permuto_sdf_amd.encoding import PermutoEncoding, encode_backward_raw

pytestmark = pytest.mark.gpu


def _shares():
    fn = L.lib().psdf_encode_backward_level_shares
    fn.restype = ctypes.c_int
    buf = (ctypes.c_int * 64)()
    n = fn(buf, 64)
    return list(buf[:n])


def test_level_shares_follow_cost_and_leave_the_gradient_alone():
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    Lv, N = 16, 2 ** 19
    import numpy as np
    enc = PermutoEncoding(3, 2 ** 18, Lv, 2, np.geomspace(1.0, 1e-4, Lv)).to(dev)
    with torch.no_grad():
        enc.lattice_values.normal_(0, 0.1)
    # ray-ordered samples (neighbouring samples share simplices at the coarse levels, as in the bench batch)
    o = torch.randn(N // 128, 1, 3, device=dev)
    d = torch.nn.functional.normalize(torch.randn(N // 128, 1, 3, device=dev), dim=2)
    t = torch.linspace(0, 0.5, 128, device=dev).view(1, 128, 1)
    pos = (0.2 * torch.tanh(o) + t * d).reshape(N, 3).contiguous()
    win = torch.ones(Lv, device=dev)
    win[12:] = 0.0                                  # four closed levels (a coarse-to-fine window)
    g = torch.randn(enc.output_dims(), N, device=dev)
    grads = []
    for it in range(10):
        gl = torch.zeros_like(enc.lattice_values)
        encode_backward_raw(enc.cfg, pos, enc.lattice_values.detach(), enc.scale_factor, enc.random_shift_per_level.detach(), win,
                            g, gl, None)
        torch.cuda.synchronize()
        grads.append(gl)
    sh = _shares()
    assert len(sh) == Lv and sum(sh) <= 6 * 256 + 256, sh
    print("shares after 10 calls:", sh)
    assert max(sh[12:]) <= 8 and min(sh[:12]) > 8, sh           # closed levels: the minimum; open ones share the rest
    assert sh[11] > sh[0], sh                                    # the finest open level costs more than the coarsest
    ref, last = grads[0], grads[-1]                              # first call: equal shares; last: the converged deal
    scale = float(ref.abs().max())
    assert float((ref - last).abs().max()) <= 2e-6 * scale
    assert float(last[12:].abs().max()) == 0.0
