"""First-contact timing probe (not a test, not the bench): encode fwd/bwd, MLP fwd at the BASELINE size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from permuto_sdf_amd import PermutoEncoding, FusedMLP

dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0), "cpus", os.cpu_count())


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


N = 2 * 1024 * 1024
for L_ in (16, 24):
    torch.manual_seed(0)
    enc = PermutoEncoding(3, 2 ** 18, L_, 2, np.geomspace(1.0, 1e-4, L_), concat_points=True, concat_points_scaling=1e-3).to(dev)
    # points uniform in the radius-0.5 ball
    p = torch.randn(N, 3, device=dev); p = p / p.norm(dim=1, keepdim=True) * 0.5 * torch.rand(N, 1, device=dev) ** (1 / 3)
    win = torch.ones(L_, device=dev)
    t = timeit(lambda: enc.forward_feature_major(p, win))
    print(f"L={L_} encode fwd {t:.3f} ms  {N / t / 1e3:.1f} Msamples/s  algGB/s {N * 4 * (3 + L_ * 2 * 4 + L_ * 2) / t / 1e6:.1f}")
    C = enc.output_dims()
    g = torch.randn(C, N, device=dev).t()
    def bwd():
        out = enc(p, win)
        torch.autograd.grad(out, enc.lattice_values, g)
    t2 = timeit(bwd, n=5, warm=2)
    print(f"L={L_} encode fwd+bwd(lattice) {t2:.3f} ms -> bwd ~{t2 - t:.3f} ms")
    pp = p.clone().requires_grad_(True)
    def bwd2():
        out = enc(pp, win)
        torch.autograd.grad(out, [enc.lattice_values, pp], g)
    t3 = timeit(bwd2, n=5, warm=2)
    print(f"L={L_} encode fwd+bwd(lattice+pos) {t3:.3f} ms")
    # per-level bwd timing through the raw ABI to see the contention profile
    mlp = FusedMLP([C, 64, 64, 64, 1]).to(dev)
    x = enc.forward_feature_major(p, win)
    tm = timeit(lambda: mlp.forward_feature_major(x))
    flops = 2 * (C * 64 + 64 * 64 * 2 + 64) * N
    print(f"L={L_} mlp fwd {tm:.3f} ms  {N / tm / 1e3:.1f} Msamples/s  {flops / tm / 1e9:.2f} TFLOP/s")
    ref = torch.nn.Sequential(*[m for i, l in enumerate(mlp.layers) for m in ([l, torch.nn.GELU()] if i < 3 else [l])])
    xt = x.t().contiguous()
    with torch.no_grad():
        tr = timeit(lambda: ref(xt))
    print(f"L={L_} torch.nn mlp fwd {tr:.3f} ms")

print("---- per-level fwd+bwd(lattice), single-level encodings, 2M points")
p = torch.randn(N, 3, device=dev); p = p / p.norm(dim=1, keepdim=True) * 0.5 * torch.rand(N, 1, device=dev) ** (1 / 3)
for sc in np.geomspace(1.0, 1e-4, 16):
    enc1 = PermutoEncoding(3, 2 ** 18, 1, 2, [sc], concat_points=False).to(dev)
    w1 = torch.ones(1, device=dev)
    g1 = torch.randn(2, N, device=dev).t()
    tf = timeit(lambda: enc1.forward_feature_major(p, w1), n=5, warm=2)
    def b1():
        torch.autograd.grad(enc1(p, w1), enc1.lattice_values, g1)
    tb = timeit(b1, n=5, warm=2)
    nz = 0
    print(f"scale {sc:.5f}: fwd {tf:.3f} ms  fwd+bwd {tb:.3f} ms")
