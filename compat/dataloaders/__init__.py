"""Stand-in for the DataLoaders package (`from dataloaders import *`, train_permuto_sdf.py:19; common_utils.py:412-415).
Dataset loaders raise when constructed (no datasets offline); `TensorReel` is a plain container with the fields
`PermutoSDF.random_rays_from_reel` reads (src/PermutoSDF.cu:70-102)."""


def _loader(name):
    def __init__(self, *a, **k):
        raise RuntimeError("dataloaders stand-in: %s needs the DataLoaders package and its dataset (compat/README.md)" % name)
    return type(name, (), {"__init__": __init__})


for _n in ["DataLoaderEasyPBR", "DataLoaderMultiFace", "DataLoaderPhenorobCP1", "DataLoaderDTU", "DataLoaderNerf",
           "DataLoaderColmap", "DataLoaderBlenderFB", "DataLoaderShapeNetImg", "DataLoaderSRN", "DataLoaderLLFF"]:
    globals()[_n] = _loader(_n)


class TensorReel:
    def __init__(self, rgb_reel=None, mask_reel=None, K_reel=None, tf_world_cam_reel=None, has_mask=False):
        self.rgb_reel, self.mask_reel, self.K_reel, self.tf_world_cam_reel = rgb_reel, mask_reel, K_reel, tf_world_cam_reel
        self.has_mask = has_mask


class MiscDataFuncs:
    @staticmethod
    def frames2tensors(frames):
        raise RuntimeError("dataloaders stand-in: frames2tensors needs DataLoaders frames (compat/README.md)")


__all__ = [n for n in list(globals()) if n.startswith("DataLoader")] + ["TensorReel", "MiscDataFuncs"]
