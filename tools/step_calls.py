"""Which C-ABI entry points one cfg-4 training step calls, in order, with their integer arguments (sample counts) and the GPU
time of each call (HIP events around it, the step serialised by the events' bookkeeping only): python tools/step_calls.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from permuto_sdf_amd import _lib as L  # noqa: E402
from permuto_sdf_amd.train_manual import ManualTrainer  # noqa: E402
from train_bench import SyntheticReel  # noqa: E402

dev = torch.device("cuda:0")
tr = ManualTrainer(dev)
reel = SyntheticReel(dev)
tr.iter = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for _ in range(12):
    tr.step(reel)
torch.cuda.synchronize()
log = []
orig = L.call


def call(name, *args):
    ints = [int(a.value) for a in args if isinstance(a, (ctypes.c_int, ctypes.c_long, ctypes.c_longlong)) and abs(int(a.value)) > 255]
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    orig(name, *args)
    e.record()
    log.append((name, ints, s, e))


L.call = call
import permuto_sdf_amd  # noqa: E402
for m in list(sys.modules.values()):      # modules that imported the function by name
    if m is not None and getattr(m, "__name__", "").startswith("permuto_sdf_amd") and getattr(m, "call", None) is orig:
        m.call = call
tr.step(reel)
torch.cuda.synchronize()
tot = 0.0
for name, ints, s, e in log:
    us = s.elapsed_time(e) * 1e3
    tot += us
    print("%-44s %8.1f us  %s" % (name, us, ints))
print("calls %d, sum %.0f us" % (len(log), tot))
