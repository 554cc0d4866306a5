"""cProfile of the reference's own training loop through the drop-in WITH the backward on the calling thread
(torch.autograd.set_multithreading_enabled(False): the engine's worker thread is invisible to cProfile):
PSDF_FUSE_REFERENCE_MLPS=1 python tools/prof_reference.py [sort] -- the arguments of tools/run_reference_on_gpu.py follow"""
import cProfile
import os
import pstats
import runpy
import sys

import torch

torch.autograd.set_multithreading_enabled(False)
sort = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "tottime"
rest = [a for a in sys.argv[1:] if a != sort or a.startswith("-")]
here = os.path.dirname(os.path.abspath(__file__))
sys.argv = [os.path.join(here, "run_reference_on_gpu.py")] + rest
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
finally:
    pr.disable()
    pstats.Stats(pr).sort_stats(sort).print_stats(90)
