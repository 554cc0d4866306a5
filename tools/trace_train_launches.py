"""Which Python line issues which GPU kernels in a cfg-4 training step?  torch.profiler with stacks over 5 steps; kernels are
attributed to the innermost frame inside this repository.  Prints (launches/step, GPU us/step, kernel, source line)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from permuto_sdf_amd.train_step import SyntheticReel, Trainer  # noqa: E402

dev = torch.device("cuda:0")
tr = Trainer(dev)
reel = SyntheticReel(dev)
for _ in range(12):
    tr.step(reel)
torch.cuda.synchronize()
STEPS = 5
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    for _ in range(STEPS):
        tr.step(reel)
    torch.cuda.synchronize()

agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.key_averages(group_by_stack_n=12):
    dt = getattr(ev, "self_device_time_total", 0.0)
    if dt <= 0:
        continue
    where = "?"
    for fr in ev.stack or []:
        if ROOT in fr and "tools/" not in fr:
            where = fr.replace(ROOT + "/", "")
            break
    k = (ev.key[:40], where[:110])
    agg[k][0] += ev.count
    agg[k][1] += dt
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print("device time of torch ops %.1f us/step (ctypes launches are not torch ops and do not appear)" % (tot / STEPS))
for (name, where), (n, t) in rows[:90]:
    print("%5.1f/step %8.1f us/step  %-40s %s" % (n / STEPS, t / STEPS, name, where))
