"""minimal stand-in for `hjson` (permuto_sdf_py imports it for optional config dumps): plain JSON only"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _defer import become_real  # noqa: E402
_REAL = become_real(__name__, globals())

if not _REAL:
    import json

    def load(fp, **kw):
        return json.load(fp)

    def loads(s, **kw):
        return json.loads(s)

    def dump(obj, fp, **kw):
        return json.dump(obj, fp)

    def dumps(obj, **kw):
        return json.dumps(obj)
