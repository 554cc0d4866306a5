"""Fixed costs of the fused kernels at the batch sizes of a cfg-4 training step (1 K - 50 K samples, where launch, weight-image
staging and accumulator flush -- not arithmetic -- set the time): SDF net 51-32-32-32-33, GPU time per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permuto_sdf_amd import FusedMLP  # noqa: E402
from permuto_sdf_amd.mlp import _grad_views, mlp_backward_raw, mlp_double_backward, mlp_forward_raw, pack_params  # noqa: E402

dev = torch.device("cuda:0")
dims = [int(a) for a in (sys.argv[1].split("-") if len(sys.argv) > 1 else "51-32-32-32-33".split("-"))]
torch.manual_seed(0)
m = FusedMLP(dims).to(dev)
ws, bs = [l.weight for l in m.layers], [l.bias for l in m.layers]
_, dWs, dbs = _grad_views(dims, dev=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for N in (1024, 8192, 49152, 262144):
    x = torch.randn(dims[0], N, device=dev)
    gy = torch.randn(dims[-1], N, device=dev)
    v = torch.randn(dims[0], N, device=dev)
    packed = pack_params(dims, ws, bs)
    r = {"fwd": timed(lambda: mlp_forward_raw(dims, x, packed)),
         "pack": timed(lambda: pack_params(dims, ws, bs)),
         "bwd dx+dW": timed(lambda: mlp_backward_raw(dims, x, ws, bs, gy, need_dx=True, into=(dWs, dbs))),
         "bwd dx": timed(lambda: mlp_backward_raw(dims, x, ws, bs, gy, need_dx=True, need_dw=False))}
    try:
        r["dbl bwd"] = timed(lambda: mlp_double_backward(dims, x, ws, bs, gy, v, into=(dWs, dbs)))
    except Exception as ex:   # widths without a fused double backward
        r["dbl bwd"] = float("nan")
    print("N=%7d  " % N + "  ".join("%s %.1f us" % kv for kv in r.items()), flush=True)
