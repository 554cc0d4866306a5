import sys, torch
sys.path.insert(0, "/root/repo")
from permuto_sdf_amd.train_step import SyntheticReel, Trainer
from permuto_sdf_amd.bridge import OccupancyGrid, RaySampler, VolumeRendering
dev = torch.device("cuda:0")
reel = SyntheticReel(dev, nr_images=4, height=60, width=80)
res = {}
for flag in (False, True, False):
    for cls in (OccupancyGrid, RaySampler, VolumeRendering):
        cls._rng = type(cls._rng)()
    tr = Trainer(dev, seed=1, touched_rows=flag)
    tr.nr_rays = 128
    grads = {}
    orig = tr.opt.step
    def fake(grad_scale=1.0, tr=tr, grads=grads, flag=flag):
        for name, m in (("sdf", tr.sdf), ("rgb", tr.rgb), ("bg", tr.bg)):
            p = m.encoding.lattice_values
            grads[name] = (m.encoding.touched_rows.grad.clone() if flag else p.grad.clone())
            if flag:
                grads[name + "_touched"] = m.encoding.touched_rows.touched.clone()
    tr.opt.step = fake
    loss = tr.step(reel)
    key = ("touched" if flag else "dense") + ("2" if (not flag and "dense" in res) else "")
    res[key] = (float(loss), grads)
    print(key, "loss", float(loss), "fg", tr.last)
for name in ("sdf", "rgb", "bg"):
    a, b, c = res["dense"][1][name], res["dense2"][1][name], res["touched"][1][name]
    print(name, "dense-dense max", float((a-b).abs().max()), "touched-dense max", float((c-a).abs().max()), "scale", float(a.abs().max()),
          "nonzero dense", int((a!=0).sum()), "nonzero touched", int((c!=0).sum()))
    t = res["touched"][1][name+"_touched"]
    rows = (c != 0).any(-1).view(24, -1, 128).any(-1)
    print("   written-not-touched blocks:", int((rows & ~t.bool()).sum()), "touched frac", float(t.float().mean()))
