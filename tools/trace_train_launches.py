"""Which torch ops launch kernels in a cfg-4 training step, and on what shapes?  torch.profiler over 5 steps (Python stacks
are not recorded by this build, the shapes identify the call sites).  python tools/trace_train_launches.py"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from permuto_sdf_amd.train_step import SyntheticReel, Trainer  # noqa: E402

dev = torch.device("cuda:0")
if "--manual" in sys.argv:
    from permuto_sdf_amd.train_manual import ManualTrainer
    tr = ManualTrainer(dev)
else:
    tr = Trainer(dev)
reel = SyntheticReel(dev)
for _ in range(12):
    tr.step(reel)
torch.cuda.synchronize()
STEPS = 5
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(STEPS):
        tr.step(reel)
    torch.cuda.synchronize()

agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.key_averages(group_by_input_shape=True):
    dt = getattr(ev, "self_device_time_total", 0.0)
    if dt <= 0 or not ev.key.startswith("aten::"):
        continue
    k = (ev.key[:28], str(ev.input_shapes)[:120])
    agg[k][0] += ev.count
    agg[k][1] += dt
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
print("torch ops that launch kernels, by input shapes (launches/step, GPU us/step)")
for (name, shapes), (n, t) in rows[:110]:
    print("%5.1f/step %8.1f us/step  %-28s %s" % (n / STEPS, t / STEPS, name, shapes))
