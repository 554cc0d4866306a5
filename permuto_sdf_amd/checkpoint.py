"""On-disk compatibility with the reference's checkpoints.

The reference saves `model.state_dict()` per network (permuto_sdf_py/models/models.py:296-307 and the analogous methods
of RGB / NerfHash) and the two grid tensors, and loads them by file name (permuto_sdf_py/utils/permuto_sdf_utils.py:222-237):
    sdf_model.pt, rgb_model.pt, nerf_hash_model_bg.pt, grid_values.pt, grid_occupancy.pt
This module writes / reads those files for the networks of `train_step.py`, renaming the keys to the reference's:
a fused `layers.{i}` stack is `{2i}` inside the reference's torch.nn.Sequential (GELU modules sit at the odd indices), the
NeuS variance lives at `volume_renderer_neus.deviation_network.variance`.  Tensor layouts are already the reference's
(torch.nn.Linear [out, in]; lattice_values [levels, capacity, features])."""
import os
import re

import torch

FILES = {"sdf": "sdf_model.pt", "rgb": "rgb_model.pt", "bg": "nerf_hash_model_bg.pt"}
_SEQUENTIAL = {"sdf": ("mlp_sdf",), "rgb": (), "bg": ("mlp_feat_and_density", "mlp_rgb")}


def to_reference_keys(kind, state_dict):
    out = {}
    for k, v in state_dict.items():
        for name in _SEQUENTIAL[kind]:
            m = re.match(r"^%s\.layers\.(\d+)\.(weight|bias)$" % name, k)
            if m:
                k = "%s.%d.%s" % (name, 2 * int(m.group(1)), m.group(2))
        if kind == "rgb" and k == "variance":
            k = "volume_renderer_neus.deviation_network.variance"
        out[k] = v
    return out


def from_reference_keys(kind, state_dict):
    out = {}
    for k, v in state_dict.items():
        for name in _SEQUENTIAL[kind]:
            m = re.match(r"^%s\.(\d+)\.(weight|bias)$" % name, k)
            if m:
                k = "%s.layers.%d.%s" % (name, int(m.group(1)) // 2, m.group(2))
        if kind == "rgb" and k == "volume_renderer_neus.deviation_network.variance":
            k = "variance"
        out[k] = v
    return out


def save(folder, sdf=None, rgb=None, bg=None, grid=None, colorcal=None):
    """writes the reference's file set into `folder` (the reference's <ckpt>/<experiment>/<iter>/models directory)"""
    os.makedirs(folder, exist_ok=True)
    for kind, model in (("sdf", sdf), ("rgb", rgb), ("bg", bg)):
        if model is not None:
            torch.save(to_reference_keys(kind, model.state_dict()), os.path.join(folder, FILES[kind]))
    if colorcal is not None:      # models.py:753-760: colorcal_model.pt, keys weight_delta / bias
        torch.save(colorcal.state_dict(), os.path.join(folder, "colorcal_model.pt"))
    if grid is not None:
        torch.save(grid.get_grid_values(), os.path.join(folder, "grid_values.pt"))
        torch.save(grid.get_grid_occupancy(), os.path.join(folder, "grid_occupancy.pt"))


def load(folder, sdf=None, rgb=None, bg=None, grid=None, colorcal=None, map_location=None, strict=True):
    for kind, model in (("sdf", sdf), ("rgb", rgb), ("bg", bg)):
        if model is not None:
            sd = torch.load(os.path.join(folder, FILES[kind]), map_location=map_location)
            model.load_state_dict(from_reference_keys(kind, sd), strict=strict)
    if colorcal is not None:
        colorcal.load_state_dict(torch.load(os.path.join(folder, "colorcal_model.pt"), map_location=map_location), strict=strict)
    if grid is not None:
        grid.set_grid_values(torch.load(os.path.join(folder, "grid_values.pt"), map_location=map_location))
        grid.set_grid_occupancy(torch.load(os.path.join(folder, "grid_occupancy.pt"), map_location=map_location))
