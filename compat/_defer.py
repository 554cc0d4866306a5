"""the stand-ins of compat/ are ordinary top-level packages, so with compat/ ahead of site-packages they would
shadow a REAL installation of torchvision / wandb / scikit-image / hjson / torchnet.  Every stand-in therefore first calls
`become_real(__name__, globals())`: if the same distribution is importable from any OTHER entry of sys.path, the stand-in
turns itself into it (runs the real package's __init__ in its own namespace, with the real __path__) and returns True."""
import importlib.machinery
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def become_real(name, namespace):
    paths = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != _HERE]
    try:
        spec = importlib.machinery.PathFinder.find_spec(name, paths)
    except (ImportError, ValueError):
        spec = None
    if spec is None or spec.origin is None or os.path.abspath(os.path.dirname(spec.origin)).startswith(_HERE):
        return False
    namespace["__file__"] = spec.origin
    namespace["__spec__"] = spec
    if spec.submodule_search_locations is not None:
        namespace["__path__"] = list(spec.submodule_search_locations)
    with open(spec.origin, "rb") as f:
        code = compile(f.read(), spec.origin, "exec")
    exec(code, namespace)
    return True
