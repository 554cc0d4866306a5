"""Host-side view of csrc/encode_conventions.h: the frozen conventions of the permutohedral encoding are defined in that
ONE header (the kernels #include it); this module parses its `#define NAME value` lines so that the scale-factor formula,
the parameter initialisation and the default concatenation layout of `PermutoEncoding` follow the same file."""
import math
import os
import re

HEADER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "encode_conventions.h")

CONCAT_NONE, CONCAT_PSEUDO_LEVELS, CONCAT_APPEND = 0, 1, 2


def parse(path=HEADER):
    out = {}
    for m in re.finditer(r"^#define\s+(PSDF_ENC_[A-Z0-9_]+)\s+([-+0-9.eE]+)\s*$", open(path).read(), re.M):
        v = m.group(2)
        out[m.group(1)] = float(v) if any(ch in v for ch in ".eE") else int(v)
    return out


C = parse()
assert (C["PSDF_ENC_CONCAT_NONE"], C["PSDF_ENC_CONCAT_PSEUDO_LEVELS"], C["PSDF_ENC_CONCAT_APPEND"]) == (0, 1, 2)


def scale_term(i, pos_dim, c=C):
    """scale_factor[l][i] = 1 / (scale_term(i) * scale_list[l])"""
    t = math.sqrt((i + 1) * (i + 2)) if c["PSDF_ENC_SCALE_SQRT_TERM"] else 1.0
    if c["PSDF_ENC_SCALE_INV_STDDEV"]:
        t /= (pos_dim + 1) * math.sqrt(2.0 / 3.0)
    return t


def concat_mode(concat_points, layout=None, c=C):
    """`concat_points` (bool, the reference's keyword) + optional explicit layout -> PSDF_ENC_CONCAT_* value"""
    if not concat_points:
        return CONCAT_NONE
    layout = c["PSDF_ENC_CONCAT_DEFAULT_LAYOUT"] if layout is None else layout
    if layout in ("pseudo_levels", CONCAT_PSEUDO_LEVELS):
        return CONCAT_PSEUDO_LEVELS
    if layout in ("append", CONCAT_APPEND):
        return CONCAT_APPEND
    raise ValueError("concat layout must be 'pseudo_levels' (1) or 'append' (2), got %r" % (layout,))


def channels(pos_dim, nr_levels, nr_feat, mode):
    if mode == CONCAT_PSEUDO_LEVELS:
        return nr_feat * (nr_levels + int(math.ceil(pos_dim / nr_feat)))
    if mode == CONCAT_APPEND:
        return nr_feat * nr_levels + pos_dim
    return nr_feat * nr_levels
