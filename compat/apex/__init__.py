"""`apex` as the reference sees it: train_permuto_sdf.py:60-64,300-303 uses `apex.optimizers.FusedAdam` instead of
`torch.optim.AdamW` whenever the package imports -- the reference's own plug-in point for a fused optimiser.  NVIDIA's apex does
not exist for this platform; this stand-in supplies that ONE class on the fused AdamW kernels of this repository
(csrc/optim.hip).  A real apex elsewhere on sys.path takes precedence (`_defer.become_real`); PSDF_COMPAT_NO_APEX=1 makes the
import fail, so that the reference falls back to torch.optim.AdamW (A/B runs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _defer import become_real  # noqa: E402

_REAL = become_real(__name__, globals())

if not _REAL:
    if os.environ.get("PSDF_COMPAT_NO_APEX") == "1":
        raise ImportError("compat/apex disabled by PSDF_COMPAT_NO_APEX=1")
    from . import optimizers  # noqa: E402,F401
