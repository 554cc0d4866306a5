"""minimal stand-in for `hjson` (permuto_sdf_py imports it for optional config dumps): plain JSON only"""
import json


def load(fp, **kw):
    return json.load(fp)


def loads(s, **kw):
    return json.loads(s)


def dump(obj, fp, **kw):
    return json.dump(obj, fp)


def dumps(obj, **kw):
    return json.dumps(obj)
