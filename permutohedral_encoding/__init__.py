"""Drop-in for the external package `permutohedral_encoding` that the reference imports as `permuto_enc`
(permuto_sdf_py/models/models.py:20): `PermutoEncoding`, `Coarse2Fine`."""
from permuto_sdf_amd.encoding import Coarse2Fine, PermutoEncoding

__all__ = ["PermutoEncoding", "Coarse2Fine"]
