"""the reference's Python imports UNMODIFIED against the drop-in packages + compat/ stand-ins.
Needs the reference checkout (this container only; skipped elsewhere).  Runs in a subprocess so that the stand-ins
(PYTHONPATH, sitecustomize) never leak into the other tests."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

SCRIPT = r'''
import importlib
mods = ["permuto_sdf_py.utils.common_utils", "permuto_sdf_py.volume_rendering.volume_rendering_funcs",
        "permuto_sdf_py.volume_rendering.volume_rendering_modules", "permuto_sdf_py.models.modules",
        "permuto_sdf_py.models.models", "permuto_sdf_py.utils.nerf_utils", "permuto_sdf_py.utils.sdf_utils",
        "permuto_sdf_py.utils.permuto_sdf_utils", "permuto_sdf_py.schedulers.multisteplr",
        "permuto_sdf_py.schedulers.warmup", "permuto_sdf_py.callbacks.callback_utils"]
for m in mods:
    importlib.import_module(m)
import permuto_sdf, permutohedral_encoding, torch
from permuto_sdf_py.models.models import SDF, LipshitzMLP
assert permuto_sdf.OccupancyGrid.__module__.startswith("permuto_sdf_amd")
assert permutohedral_encoding.PermutoEncoding.__module__.startswith("permuto_sdf_amd")
# the reference's SDF model builds on the drop-in encoding (CPU construction; evaluation needs the GPU)
sdf = SDF(3, None, 32, 10000)
assert sdf.encoding.output_dims() == sdf.mlp_sdf[0].in_features
names = dict(sdf.named_parameters())
assert any("lattice_values" in k for k in names)            # models.py:408-420 looks parameters up by this name
# LR schedulers of the reference (pass `verbose` positionally: needs the compat shim on PyTorch >= 2.7)
from permuto_sdf_py.schedulers.multisteplr import MultiStepLR
from permuto_sdf_py.schedulers.warmup import GradualWarmupScheduler
p = torch.nn.Parameter(torch.zeros(3))
opt = torch.optim.AdamW([p], lr=1e-3)
sched = GradualWarmupScheduler(opt, multiplier=1, total_epoch=3, after_scheduler=MultiStepLR(opt, milestones=[5, 10], gamma=0.3))
for _ in range(12):
    opt.step(); sched.step()
assert 0 < opt.param_groups[0]["lr"] < 1e-3
# train_step.lr_schedule == these scheduler objects driven the way train_permuto_sdf.py:304,419-422 drives them
from permuto_sdf_amd.train_step import HyperParams, lr_schedule
hp = HyperParams()
hp.nr_iter_sphere_fit, hp.lr_warmup_iters, hp.lr_milestones = 5, 7, (4, 9, 11)
q = torch.nn.Parameter(torch.zeros(3))
opt2 = torch.optim.AdamW([q], lr=hp.lr)
decay = MultiStepLR(opt2, milestones=list(hp.lr_milestones), gamma=0.3, verbose=False)
for it in range(40):
    assert abs(opt2.param_groups[0]["lr"] - lr_schedule(it, hp)) < 1e-15, (it, opt2.param_groups[0]["lr"], lr_schedule(it, hp))
    opt2.step()
    if it == hp.nr_iter_sphere_fit:
        warm = GradualWarmupScheduler(opt2, multiplier=1, total_epoch=hp.lr_warmup_iters, after_scheduler=decay)
    if it >= hp.nr_iter_sphere_fit:
        warm.step()
# the training script itself: everything up to the first CUDA call at module level must import
try:
    importlib.import_module("permuto_sdf_py.train_permuto_sdf")
    print("TRAIN_IMPORT full")
except (RuntimeError, AssertionError, TypeError) as e:
    assert "cuda" in str(e).lower() or "CUDA" in str(e) or "not available" in str(e), e
    print("TRAIN_IMPORT up-to-cuda")
print("COMPAT_OK")
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "permuto_sdf_py")), reason="reference checkout not present")
def test_reference_python_imports_unmodified():
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "compat"), REF])
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, cwd="/tmp", timeout=600)
    assert r.returncode == 0 and "COMPAT_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


FUSE_SCRIPT = r'''
import torch
torch.nn.Module.cuda = lambda self, *a, **k: self        # VolumeRenderingNeus.__init__ calls .cuda() (volume_rendering_modules.py:121)
from permuto_sdf_py.models.models import SDF, RGB, NerfHash
from permuto_sdf_amd import reference_fusion as RF
from permuto_sdf_amd.mlp import LipshitzMLP
# (round 5: besides the MLP sub-modules, SDF.get_sdf_and_gradient runs its inner pass input-gradient-only and the
#  VolumeRenderingNeus child of RGB evaluates its NeuS weights as one operator -- same objects, same Parameters)
for make, names in ((lambda: SDF(3, None, 32, 10000), ["get_sdf_and_gradient()", "mlp_sdf"]),
                    (lambda: RGB(3, None, 32, 1), ["volume_renderer_neus.compute_weights()", "mlp"]),
                    (lambda: NerfHash(4, None, 1), ["mlp_feat_and_density", "mlp_rgb"])):
    m = make()
    assert m.encoding._fuse_owner is not None and m.encoding._fuse_owner() is m     # the owner was found on the stack
    keys, ids = list(m.state_dict().keys()), {k: id(p) for k, p in m.named_parameters()}
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    assert RF.fuse_model(m) == names, names
    assert list(m.state_dict().keys()) == keys                    # checkpoint keys unchanged
    assert {k: id(p) for k, p in m.named_parameters()} == ids     # the very same Parameter objects (optimisers keep working)
    assert all(torch.equal(v, sd[k]) for k, v in m.state_dict().items())
    for n in names:
        if n.endswith("()"):
            continue
        sub = getattr(m, n)
        assert isinstance(sub, (RF.FusedSequential, LipshitzMLP)), type(sub)
        if isinstance(sub, RF.FusedSequential):
            assert sub.fused, (n, sub.dims)                       # every Sequential of the reference has a fused kernel
    make().load_state_dict(m.state_dict())                        # a fused model's checkpoint loads into an unfused one
    assert RF.fuse_model(m) == []                                 # idempotent
    assert sorted(RF.unfuse_model(m)) == sorted(names)            # and reversible: the reference's own classes / methods again
    assert "get_sdf_and_gradient" not in m.__dict__ and not any(hasattr(type(c), "_reference_class") for c in m.children())
print("FUSE_OK")
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "permuto_sdf_py")), reason="reference checkout not present")
def test_fusion_hook_keeps_parameters_and_checkpoint_keys():
    """PSDF_FUSE_REFERENCE_MLPS=1 on the reference's real model classes (CPU: bookkeeping only; numerics are a GPU test,
    tests/test_gpu_reference_call_patterns.py::test_fused_evaluators_behind_reference_style_models)"""
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "compat"), REF])
    env["PSDF_FUSE_REFERENCE_MLPS"] = "1"
    r = subprocess.run([sys.executable, "-c", FUSE_SCRIPT], capture_output=True, text=True, env=env, cwd="/tmp", timeout=600)
    assert r.returncode == 0 and "FUSE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


APEX_SCRIPT = r'''
import os, sys, torch
from permuto_sdf_py.utils.permuto_sdf_utils import module_exists       # the reference's own probe (train_permuto_sdf.py:60)
if os.environ.get("PSDF_COMPAT_NO_APEX") == "1":
    assert not module_exists("apex")
    print("APEX_OFF")
    sys.exit(0)
assert module_exists("apex")
import apex
from permuto_sdf_amd.optim import FusedAdamW
m = torch.nn.Linear(4, 3)
# the reference's call, train_permuto_sdf.py:293-301: named groups with their own lr / weight decay
opt = apex.optimizers.FusedAdam([{"params": m.parameters(), "weight_decay": 0.1, "lr": 2e-3, "name": "model_colorcal"}],
                                amsgrad=False, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.0, lr=1e-3)
assert isinstance(opt, FusedAdamW) and isinstance(opt, torch.optim.Optimizer)
g = opt.param_groups[0]
assert (g["name"], g["lr"], g["weight_decay"], g["betas"], g["eps"]) == ("model_colorcal", 2e-3, 0.1, (0.9, 0.99), 1e-15)
from permuto_sdf_py.schedulers.multisteplr import MultiStepLR           # the reference's schedulers drive it like any optimiser
from permuto_sdf_py.schedulers.warmup import GradualWarmupScheduler
s = GradualWarmupScheduler(opt, multiplier=1, total_epoch=4, after_scheduler=MultiStepLR(opt, milestones=[2], gamma=0.3, verbose=False))
m.weight.grad = torch.zeros_like(m.weight)
opt.zero_grad()
assert m.weight.grad is None                                           # apex default: set_grad_none=True
try:
    apex.optimizers.FusedAdam(m.parameters(), amsgrad=True)
    raise SystemExit("amsgrad accepted")
except RuntimeError:
    pass
print("APEX_OK")
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "permuto_sdf_py")), reason="reference checkout not present")
@pytest.mark.parametrize("off", [False, True])
def test_compat_apex_is_the_references_fused_optimizer_hook(off):
    """train_permuto_sdf.py:60-64,300-303 takes apex.optimizers.FusedAdam when `apex` imports: compat/apex supplies it on the fused
    AdamW kernels; PSDF_COMPAT_NO_APEX=1 hides it (the reference then builds torch.optim.AdamW)."""
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "compat"), REF])
    if off:
        env["PSDF_COMPAT_NO_APEX"] = "1"
    else:
        env.pop("PSDF_COMPAT_NO_APEX", None)
    r = subprocess.run([sys.executable, "-c", APEX_SCRIPT], capture_output=True, text=True, env=env, cwd="/tmp", timeout=600)
    assert r.returncode == 0 and ("APEX_OFF" if off else "APEX_OK") in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
