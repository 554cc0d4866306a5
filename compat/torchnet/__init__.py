"""inert stand-in for `torchnet` (imported at module level by the reference, never used on the training path)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _defer import become_real  # noqa: E402
_REAL = become_real(__name__, globals())

if not _REAL:
    from _inert import Inert  # noqa: E402

    def __getattr__(name):
        if name.startswith('__'):
            raise AttributeError(name)
        return Inert()
