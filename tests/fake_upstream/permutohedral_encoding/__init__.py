"""TEST INFRASTRUCTURE: a stand-in for the upstream `permutohedral_encoding` package with the same constructor / call API,
built on the CPU oracle (oracle/permuto_oracle.py).  It exists so that the pinning pipeline -- tools/
dump_upstream_encoding_vectors.py -> tests/golden/upstream_encoding_vectors.npz -> tests/test_upstream_vectors.py -- can be run
end to end here, where the real package is absent: the dump script is pointed at this directory, optionally with conventions
that DIFFER from encode_conventions.h (FAKE_UPSTREAM_CONV='{"PSDF_ENC_RANK_TIE_RAISES_LATER": 0, ...}'), and the comparator
must detect the difference and name the matching combination.  Parameters are stored in a deliberately different layout
([T, L, F]) to exercise the layout normalisation.  Never imported by the product."""
import json
import os

import numpy as np
import torch

from oracle import permuto_oracle as po

_CONV = dict(po.CONV)
_CONV.update(json.loads(os.environ.get("FAKE_UPSTREAM_CONV", "{}")))


class _Swap:
    def __enter__(self):
        self.old = po.CONV
        po.CONV = _CONV

    def __exit__(self, *exc):
        po.CONV = self.old


class PermutoEncoding(torch.nn.Module):
    def __init__(self, pos_dim, capacity, nr_levels, nr_feat_per_level, scale_list, appply_random_shift_per_level=True,
                 concat_points=False, concat_points_scaling=1.0):
        super().__init__()
        self.cfg = (pos_dim, capacity, nr_levels, nr_feat_per_level)
        self.scale_list = np.asarray(scale_list, np.float64)
        self.concat_points, self.concat_points_scaling = concat_points, concat_points_scaling
        self.lattice_values = torch.nn.Parameter(torch.randn(capacity, nr_levels, nr_feat_per_level) * _CONV["PSDF_ENC_LATTICE_INIT_SCALE"])
        sh = torch.randn(nr_levels, pos_dim) * _CONV["PSDF_ENC_RANDOM_SHIFT_SCALE"] if appply_random_shift_per_level else torch.zeros(nr_levels, pos_dim)
        self.random_shift_per_level = torch.nn.Parameter(sh, requires_grad=False)

    def output_dims(self):
        with _Swap():
            return po.output_dims(self.cfg[0], self.cfg[2], self.cfg[3], self.concat_points)

    def forward(self, positions, window):
        with _Swap():
            return po.encode(positions, self.lattice_values.permute(1, 0, 2), self.scale_list, self.random_shift_per_level,
                             window.reshape(-1), self.concat_points, self.concat_points_scaling)


class Coarse2Fine(torch.nn.Module):
    def __init__(self, nr_levels):
        super().__init__()
        self.nr_levels = nr_levels
        self.last_t = 0.0

    def forward(self, t):
        self.last_t = t
        return po.coarse2fine_window(t, self.nr_levels)

    def get_last_t(self):
        return self.last_t
