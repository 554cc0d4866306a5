"""Where does the HOST time of a cfg-4 training step go?  cProfile over 40 steps (after warm-up), top functions by own time
and by cumulative time.  python tools/profile_train_host.py > gpurun_out/r02/train_host_profile.txt"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permuto_sdf_amd.train_step import SyntheticReel, Trainer  # noqa: E402

dev = torch.device("cuda:0")
if "--manual" in sys.argv:
    from permuto_sdf_amd.train_manual import ManualTrainer
    tr = ManualTrainer(dev)
else:
    tr = Trainer(dev)
reel = SyntheticReel(dev)
for _ in range(15):
    tr.step(reel)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(40):
    tr.step(reel)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.strip_dirs().sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
