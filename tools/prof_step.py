"""Host-side profile of the hand-written training step (cProfile over 100 steps): python tools/prof_step.py [start_iter] [sort]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from permuto_sdf_amd.train_manual import ManualTrainer  # noqa: E402
from train_bench import SyntheticReel  # noqa: E402

dev = torch.device("cuda:0")
tr = ManualTrainer(dev)
reel = SyntheticReel(dev)
tr.iter = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for _ in range(30):
    tr.step(reel)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(100):
        tr.step(reel)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue %.3f ms/step, wall %.3f ms/step" % ((t1 - t0) * 10, (t2 - t0) * 10))
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    tr.step(reel)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats(sys.argv[2] if len(sys.argv) > 2 else "tottime").print_stats(70)
