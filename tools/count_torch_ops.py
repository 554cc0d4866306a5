import os, sys, collections, traceback, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from permuto_sdf_amd.train_manual import ManualTrainer
from train_bench import SyntheticReel
dev = torch.device("cuda:0")
tr = ManualTrainer(dev); reel = SyntheticReel(dev); tr.iter = 20000
for _ in range(12): tr.step(reel)
torch.cuda.synchronize()
cnt = collections.Counter()
def wrap(obj, name, label):
    orig = getattr(obj, name)
    def f(*a, **k):
        st = traceback.extract_stack(limit=4)[:-1]
        site = " <- ".join("%s:%d" % (os.path.basename(s.filename), s.lineno) for s in reversed(st) if "permuto_sdf_amd" in s.filename or "tools" in s.filename)
        cnt[(label, site)] += 1
        return orig(*a, **k)
    setattr(obj, name, f)
for n in ("zeros", "zeros_like", "ones", "ones_like", "full", "cat", "randn_like", "rand", "randn", "randint", "empty_like", "where", "add", "exp", "tensor"):
    wrap(torch, n, n)
for n in ("zero_", "fill_", "uniform_", "clone", "contiguous", "index_select", "index_add_", "copy_", "to", "__add__", "__mul__", "__sub__", "__radd__", "__rmul__", "t", "long", "sum"):
    wrap(torch.Tensor, n, "T." + n)
tr.step(reel)
torch.cuda.synchronize()
for (label, site), c in sorted(cnt.items(), key=lambda kv: (kv[0][0], -kv[1])):
    if label in ("T.t", "T.contiguous", "T.to") : continue
    print("%-14s %3d  %s" % (label, c, site))
