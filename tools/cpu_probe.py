import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import permuto_oracle as po
print("cpus", os.cpu_count(), "torch threads default", torch.get_num_threads(), flush=True)
for th in (8, 32, 64):
    torch.set_num_threads(th)
    N = 32768
    pos = torch.rand(N, 3) - 0.5
    lat, sh = po.make_params(3, 2 ** 18, 16, 2, seed=2)
    lat.requires_grad_(True)
    t = time.perf_counter()
    f = po.encode(pos, lat, np.geomspace(1, 1e-4, 16), sh, torch.ones(16), True, 1e-3)
    t1 = time.perf_counter()
    f.sum().backward()
    t2 = time.perf_counter()
    print(th, "threads: fwd %.2fs bwd %.2fs  (%d samples)" % (t1 - t, t2 - t1, N), flush=True)
