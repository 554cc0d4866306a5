"""Drop-in for the reference's pybind11 module `permuto_sdf` (src/PyBridge.cxx): the reference's Python code
(`from permuto_sdf import OccupancyGrid, RaySamplesPacked, VolumeRendering, ...`) imports this package unchanged.
All compute lives in permuto_sdf_amd (HIP kernels behind the C ABI of include/psdf.h)."""
from permuto_sdf_amd.bridge import (OccupancyGrid, PermutoSDF, RaySampler, RaySamplesPacked, Sphere, TrainParams,
                                    VolumeRendering)


class NGPGui:
    """Viewer-only ImGui panel (src/NGPGui.cxx); out of scope. `--no_viewer` runs never construct it."""

    @staticmethod
    def create(view):
        raise NotImplementedError("NGPGui needs the EasyPBR viewer; run with --no_viewer")


__all__ = ["OccupancyGrid", "PermutoSDF", "RaySampler", "RaySamplesPacked", "Sphere", "TrainParams", "VolumeRendering",
           "NGPGui"]
