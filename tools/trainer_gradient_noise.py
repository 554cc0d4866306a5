"""Run-to-run noise of the SDF lattice gradient of a cfg-4 step (float atomics / queue order) against the difference between the
autograd trainer and the hand-written step: same parameters, same rays, samplers' generators rewound."""
import copy, sys, torch
sys.path.insert(0, '/root/repo')
from permuto_sdf_amd.bridge import OccupancyGrid, RaySampler, VolumeRendering
from permuto_sdf_amd.train_manual import ManualTrainer
from permuto_sdf_amd.train_step import HyperParams, SyntheticReel, Trainer
dev = torch.device("cuda:0")
reel = SyntheticReel(dev, nr_images=4, height=60, width=80)
owners = (OccupancyGrid, RaySampler, VolumeRendering)
for adv in (0, 3, 7, 11):
    for c in owners:
        for _ in range(adv): c._rng.advance()
    saved = [copy.deepcopy(c._rng) for c in owners]
    res = []
    for cls in (Trainer, Trainer, ManualTrainer, ManualTrainer):
        for c, r in zip(owners, saved): c._rng = copy.deepcopy(r)
        hp = HyperParams(); hp.nr_rays, hp.target_nr_of_samples = 256, 256 * 96
        hp.nr_iter_sphere_fit, hp.lr_warmup_iters = 0, 4
        tr = cls(dev, hp, reference_schedule=True, nr_images=4)
        tr.capture_grads = {}
        tr.step(reel)
        res.append(tr.capture_grads["lattices"])
    def rel(x, y): return float((x - y).norm() / x.norm())
    print("advance", adv, "lattice0: a-a %.2e  m-m %.2e  a-m %.2e | lattice1 a-m %.2e lattice2 a-m %.2e" % (
        rel(res[0][0], res[1][0]), rel(res[2][0], res[3][0]), rel(res[0][0], res[2][0]), rel(res[0][1], res[2][1]), rel(res[0][2], res[2][2])), flush=True)
    a, m, m2 = res[0][0], res[2][0], res[3][0]
    print("   per level a-m:", " ".join("%.0e" % float((a[l] - m[l]).norm() / a[l].norm().clamp_min(1e-30)) for l in range(a.shape[0])))
    print("   per level m-m:", " ".join("%.0e" % float((m2[l] - m[l]).norm() / m[l].norm().clamp_min(1e-30)) for l in range(a.shape[0])))
    print("   level norms  :", " ".join("%.0e" % float(a[l].norm()) for l in range(a.shape[0])))
