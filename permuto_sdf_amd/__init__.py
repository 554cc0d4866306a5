"""MI355X-native (gfx950) implementation of the PermutoSDF rendering/training hot path.

Host side: Python/PyTorch-ROCm (device memory, streams, autograd, torch.distributed) calling hand-written
HIP kernels through the C-ABI library declared in include/psdf.h.  No CPU fallback exists.
"""
from . import _lib
from .encoding import PermutoEncoding, Coarse2Fine
from .mlp import FusedMLP, LipshitzMLP

__all__ = ["PermutoEncoding", "Coarse2Fine", "FusedMLP", "LipshitzMLP"]
