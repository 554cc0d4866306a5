"""Stand-in for EasyPBR (viewer / profiler), imported with `from easypbr import *` by the reference
(permuto_sdf_py/utils/common_utils.py:24, train_permuto_sdf.py:17-18, sdf_utils.py:12).  Nothing is drawn or timed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _inert import Inert, inert_class  # noqa: E402

_NAMES = ["Scene", "Viewer", "Gui", "Mesh", "Frame", "Camera", "ColorMngr", "Mat", "VisOptions", "Recorder", "MeshGL",
          "Affine3f", "Affine3d", "Quaternionf", "Texture2D", "SpotLight"]
for _n in _NAMES:
    globals()[_n] = inert_class(_n)


class Profiler:
    """TIME_START / TIME_END of the reference (common_utils.py:33-40) call these"""

    @staticmethod
    def is_profiling_gpu():
        return False

    @staticmethod
    def start(name):
        pass

    @staticmethod
    def end(name):
        pass

    @staticmethod
    def print_all_stats():
        pass

    @staticmethod
    def set_profile_gpu(v):
        pass


def tensor2mat(t):
    return Inert()


def mat2tensor(m, flip_red_blue=False):
    raise RuntimeError("easypbr stand-in: there are no images to convert (compat/README.md)")


__all__ = _NAMES + ["Profiler", "tensor2mat", "mat2tensor"]
