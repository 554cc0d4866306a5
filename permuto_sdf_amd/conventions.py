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


# ---- runtime override (a flag flip, not a rebuild) --------------------------------------------------------------------
# The header holds the DEFAULTS.  Someone who has the upstream CUDA package and finds a disagreement
# (tools/dump_upstream_encoding_vectors.py -> tests/test_upstream_vectors.py names the combination that matches) flips the
# convention for the process, either in code -- conventions.set(rank_tie_raises_later=0) -- or from the environment:
#   PSDF_ENC_CONVENTIONS="RANK_TIE_RAISES_LATER=0,SCALE_INV_STDDEV=1,HASH_MULTIPLIER=2654435761"
# Device-side conventions (hash multiplier, tie rule) are pushed into the library (psdf_encode_set_conventions, kernel
# arguments from then on); host-side ones (scale_factor formula, default concatenation layout, initialisation scales) take
# effect for every PermutoEncoding constructed afterwards.
_KEYS = {"hash_multiplier": "PSDF_ENC_HASH_MULTIPLIER", "rank_tie_raises_later": "PSDF_ENC_RANK_TIE_RAISES_LATER",
         "scale_sqrt_term": "PSDF_ENC_SCALE_SQRT_TERM", "scale_inv_stddev": "PSDF_ENC_SCALE_INV_STDDEV",
         "concat_default_layout": "PSDF_ENC_CONCAT_DEFAULT_LAYOUT", "lattice_init_scale": "PSDF_ENC_LATTICE_INIT_SCALE",
         "random_shift_scale": "PSDF_ENC_RANDOM_SHIFT_SCALE"}
DEFAULTS = dict(C)


def set(push_to_library=True, **kw):   # noqa: A001  (deliberately the obvious name: conventions.set(...))
    """conventions.set(hash_multiplier=..., rank_tie_raises_later=0|1, scale_sqrt_term=0|1, scale_inv_stddev=0|1,
    concat_default_layout=1|2, lattice_init_scale=..., random_shift_scale=...) -> dict of the values now in force"""
    for k, v in kw.items():
        name = _KEYS.get(k.lower(), k if k in C else "PSDF_ENC_" + k.upper())
        if name not in C:
            raise KeyError("unknown encoding convention %r (known: %s)" % (k, ", ".join(sorted(_KEYS))))
        C[name] = type(DEFAULTS[name])(v)
    if push_to_library:
        import ctypes
        from . import _lib as L
        fn = L.lib().psdf_encode_set_conventions
        fn.restype = ctypes.c_int
        L.check(fn(ctypes.c_uint32(int(C["PSDF_ENC_HASH_MULTIPLIER"]) & 0xFFFFFFFF),
                   ctypes.c_int(int(C["PSDF_ENC_RANK_TIE_RAISES_LATER"]))), "psdf_encode_set_conventions")
    return dict(C)


def reset(push_to_library=True):
    C.update(DEFAULTS)
    return set(push_to_library=push_to_library)


def _from_env():
    import os
    spec = os.environ.get("PSDF_ENC_CONVENTIONS", "").strip()
    if not spec:
        return
    kw = {}
    for item in spec.split(","):
        k, _, v = item.partition("=")
        kw[k.strip()] = float(v) if any(ch in v for ch in ".eE") else int(v, 0)
    set(**kw)


_from_env()
