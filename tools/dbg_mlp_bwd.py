import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import permuto_sdf_amd._lib as LL
if os.environ.get('DBG_LIB'):
    LL.LIB_PATH = os.environ['DBG_LIB']
from permuto_sdf_amd import FusedMLP
from tests.test_gpu_mlp import _ref_net, BWD_NETS
dev = torch.device("cuda:0")
for N in (16, 5000):
    for dims in BWD_NETS[:3]:
        torch.manual_seed(dims[0] + N)
        ref = _ref_net(dims)
        x = torch.randn(N, dims[0]); gy = torch.randn(N, dims[-1])
        ref64 = _ref_net(dims).double()
        ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
        x64 = x.double().requires_grad_(True)
        ref64(x64).backward(gy.double())
        m = FusedMLP.from_sequential(ref).to(dev)
        xd = x.to(dev).requires_grad_(True)
        m(xd).backward(gy.to(dev))
        sc = lambda t: max(1e-6, t.abs().max().item())
        errs = ["dx %.1e" % ((xd.grad.cpu().double() - x64.grad).abs().max() / sc(x64.grad))]
        lin64 = [l for l in ref64 if isinstance(l, torch.nn.Linear)]
        for i, (a, b) in enumerate(zip(m.layers, lin64)):
            e = (a.weight.grad.cpu().double() - b.weight.grad).abs()
            errs.append("W%d %.1e" % (i, e.max() / sc(b.weight.grad)))
            if e.max() / sc(b.weight.grad) > 1e-4:
                bad = (e > 1e-4 * sc(b.weight.grad)).nonzero()
                errs.append("bad rows %s cols %s" % (sorted(set(bad[:, 0].tolist()))[:8], sorted(set(bad[:, 1].tolist()))[:20]))
            errs.append("b%d %.1e" % (i, (a.bias.grad.cpu().double() - b.bias.grad).abs().max() / sc(b.bias.grad)))
        print(N, dims, " ".join(errs), flush=True)
