"""Host-side dispatch predicates (permuto_sdf_amd/mlp.py) must list exactly the template instantiations the C ABI
dispatches to (csrc/mlp_bwd.hip CASE tables): a mismatch would surface only at run time as status -2 or as a needless
torch fallback."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cases(src, func):
    body = src[src.index("int %s(" % func):]
    body = body[:body.index("return PSDF_ERR_UNSUPPORTED;\n}")]
    out = set()
    # CASE_DX_ONLY: shapes whose parameter gradients come from mlp_wide.hip (psdf_mlp_backward routes them there first); the
    # single-wave kernel serves their data gradient only
    for m in re.finditer(r"^\s*CASE(?:_DX_ONLY)?\((\d+), (\d+), (\d+), (\d+), (\d+), (true|false)\)", body, re.M):
        out.add(tuple(int(x) for x in m.groups()[:5]) + (m.group(6) == "true",))
    return out


def _python_set(func_name):
    import ast
    src = open(os.path.join(ROOT, "permuto_sdf_amd", "mlp.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == func_name][0]
    sets = [n for n in ast.walk(fn) if isinstance(n, ast.Set)]
    return set(ast.literal_eval(ast.unparse(sets[0])))


def test_backward_and_double_backward_tables_match():
    src = open(os.path.join(ROOT, "permuto_sdf_amd", "csrc", "mlp_bwd.hip")).read()
    assert _cases(src, "psdf_mlp_backward") == _python_set("backward_supported")
    # (the table lives in the function both psdf_mlp_double_backward and psdf_mlp_double_backward_plus call)
    assert _cases(src, "mlp_double_backward_impl") == _python_set("double_backward_supported")
    # the fused form (double backward + plain backward of an output gradient): exactly the two SDF-net instantiations, and the host
    # predicate admits exactly their tile signatures
    plus = set(re.findall(r"launch_dbl_bwd<(\d), (\d), (\d), (\d), (\d), false, true>", src))
    assert plus == {("4", "2", "2", "2", "3"), ("3", "2", "2", "2", "3")}


def test_predicates_on_the_nets_of_the_reference():
    from permuto_sdf_amd.mlp import backward_supported, double_backward_supported
    assert backward_supported([52, 32, 32, 32, 33]) and double_backward_supported([52, 32, 32, 32, 33])   # SDF net
    assert backward_supported([36, 64, 64, 64, 1]) and double_backward_supported([36, 64, 64, 64, 1])     # BASELINE net
    assert backward_supported([52, 64, 64, 64, 65]) and backward_supported([80, 64, 64, 3])               # background nets
    from permuto_sdf_amd.mlp import double_backward_plus_supported
    assert double_backward_plus_supported([52, 32, 32, 32, 33]) and double_backward_plus_supported([36, 32, 32, 32, 33])
    assert not double_backward_plus_supported([36, 64, 64, 64, 1]) and not double_backward_plus_supported([52, 32, 32, 32, 1])
    assert backward_supported([112, 128, 128, 64, 3]) and backward_supported([111, 128, 128, 64, 3])     # colour net: mlp_wide.hip
    assert not backward_supported([200, 256, 256, 64, 3]) and not double_backward_supported([112, 128, 128, 64, 3])
    assert not double_backward_supported([80, 64, 64, 3])
