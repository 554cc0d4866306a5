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


class Mesh(inert_class("Mesh")):
    """The geometry side of easypbr's Mesh is real: V [n,3], F [m,3], NV [n,3], C [n,3] as numpy arrays and `save_to_file`
    (.ply binary little-endian, .obj) -- what the reference's mesh export hands over (permuto_sdf_py/utils/sdf_utils.py:283-290,
    experiments/evaluation/create_my_meshes.py:162).  Everything viewer-related stays inert."""

    def __init__(self, *a, **k):
        import numpy as np
        self.V, self.F, self.NV, self.C = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32)
        self.name = ""

    def is_empty(self):
        return len(self.V) == 0

    def save_to_file(self, path):
        import numpy as np
        V = np.asarray(self.V, np.float32).reshape(-1, 3)
        F = np.asarray(self.F, np.int32).reshape(-1, 3)
        NV = np.asarray(self.NV, np.float32).reshape(-1, 3)
        has_n = len(NV) == len(V) and len(V) > 0
        os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
        if path.lower().endswith(".obj"):
            with open(path, "w") as f:
                for v in V:
                    f.write("v %.8g %.8g %.8g\n" % tuple(v))
                if has_n:
                    for n in NV:
                        f.write("vn %.8g %.8g %.8g\n" % tuple(n))
                for t in F + 1:
                    f.write(("f %d//%d %d//%d %d//%d\n" % (t[0], t[0], t[1], t[1], t[2], t[2])) if has_n else ("f %d %d %d\n" % tuple(t)))
            return
        props = "property float x\nproperty float y\nproperty float z\n" + ("property float nx\nproperty float ny\nproperty float nz\n" if has_n else "")
        header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\n%selement face %d\nproperty list uchar int vertex_indices\nend_header\n"
                  % (len(V), props, len(F)))
        vert = np.concatenate([V, NV], 1).astype("<f4") if has_n else V.astype("<f4")
        face = np.empty(len(F), dtype=[("n", "u1"), ("i", "<i4", (3,))])
        face["n"], face["i"] = 3, F
        with open(path, "wb") as f:
            f.write(header.encode("ascii"))
            f.write(vert.tobytes())
            f.write(face.tobytes())


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
