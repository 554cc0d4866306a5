"""Loaded automatically when compat/ is on PYTHONPATH: restores `torch._six` (removed in PyTorch 2; the reference's
schedulers import `inf` from it, permuto_sdf_py/schedulers/multisteplr.py:5) and supplies an inert
`torch.utils.tensorboard.SummaryWriter` when the tensorboard package is absent -- without importing torch at start-up."""
import importlib.abc
import importlib.machinery
import sys
import types


def _accept_verbose_argument():
    """The reference's MultiStepLR passes `verbose` positionally to the scheduler base class
    (permuto_sdf_py/schedulers/multisteplr.py:48); PyTorch >= 2.7 removed that parameter.  Called when `torch._six` is
    requested, i.e. exactly by the modules that need it (torch is imported by then)."""
    import torch.optim.lr_scheduler as lrs
    base = lrs.LRScheduler
    if getattr(base.__init__, "_psdf_compat", False):
        return
    orig = base.__init__

    def __init__(self, optimizer, last_epoch=-1, verbose=None, *args, **kwargs):
        orig(self, optimizer, last_epoch)
    __init__._psdf_compat = True
    base.__init__ = __init__


class _TorchSix(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname == "torch._six":
            return importlib.machinery.ModuleSpec(fullname, self)
        if fullname == "torch.utils.tensorboard":   # needs the `tensorboard` package; the reference only logs through it
            try:
                import tensorboard  # noqa: F401
                return None
            except ImportError:
                return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        if spec.name == "torch.utils.tensorboard":
            m = types.ModuleType(spec.name)

            class SummaryWriter:
                def __init__(self, *a, **k):
                    pass

                def __getattr__(self, name):
                    return lambda *a, **k: None
            m.SummaryWriter = SummaryWriter
            return m
        _accept_verbose_argument()
        m = types.ModuleType(spec.name)
        m.inf = float("inf")
        m.nan = float("nan")
        m.string_classes = (str, bytes)
        m.int_classes = (int,)
        import collections.abc as cabc
        m.container_abcs = cabc
        return m

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _TorchSix())
