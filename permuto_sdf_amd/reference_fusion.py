"""Fused evaluators BEHIND THE UNMODIFIED REFERENCE MODELS (north_star: "the fused small-MLP SDF+color evaluators
... so permuto_sdf_py/models and train_permuto_sdf.py run unmodified").

The reference builds its networks as `torch.nn.Sequential(Linear, GELU, ..., Linear)` (permuto_sdf_py/models/models.py:153-161
`SDF.mlp_sdf`, :451-470 `NerfHash.mlp_feat_and_density` / `mlp_rgb`) and its own `LipshitzMLP` class (:54-129, `RGB.mlp`), so through
the drop-in packages alone they run on rocBLAS + elementwise kernels.  `fuse_model(model)` swaps those sub-modules, in place, for

  * `FusedSequential` -- a torch.nn.Sequential holding THE SAME Linear / GELU module objects (same Parameters, same
    `state_dict()` keys `mlp_sdf.0.weight`, `mlp_sdf.2.bias`, ...: checkpoints are untouched) whose forward is one launch of the fused
    MFMA evaluator (csrc/mlp.hip), with its backward / double backward kernels (csrc/mlp_bwd*.hip);
  * `permuto_sdf_amd.mlp.LipshitzMLP` re-using the reference module's Parameters (`layers.i.weight|bias`,
    `lipshitz_bound_per_layer.i`: the same names).

Widths without a fused kernel keep torch's evaluation (forward) or fall back to it (backward: allow_torch_fallback is set for these
modules, the reference must keep working whatever it builds).  Opt-in: with `PSDF_FUSE_REFERENCE_MLPS=1` every `PermutoEncoding` that is
constructed inside the constructor of a reference model (`SDF`, `RGB`, `NerfHash`: models.py:149,333,442) remembers that model and
fuses it at its own first forward call -- no import hook, nothing of the reference is patched; `fuse_model` can also be called by
hand on any module.
"""
import os
import sys

import torch

from . import mlp as M


def forward_supported(dims):
    """True when csrc/mlp.hip has a forward instantiation for these widths (tiles of 32; the CASE table of mlp_forward_impl)"""
    n_layers = len(dims) - 1
    if n_layers not in (3, 4) or dims[0] > 128:
        return False
    t = [(d + 31) // 32 for d in dims]
    sig = (t[1], t[2], t[3] if n_layers == 4 else 0, t[n_layers], dims[-1] <= 4)
    return sig in {(2, 2, 2, 1, True), (1, 1, 1, 1, True), (1, 1, 1, 2, False), (2, 2, 2, 3, False), (2, 2, 2, 2, False),
                   (2, 2, 0, 1, True), (4, 4, 2, 1, True)}


def _is_linear_gelu_stack(seq):
    mods = list(seq)
    if len(mods) < 3 or len(mods) % 2 == 0:
        return False
    for i, m in enumerate(mods):
        if i % 2 == 0:
            if type(m) is not torch.nn.Linear or m.bias is None:
                return False
        elif type(m) is not torch.nn.GELU or getattr(m, "approximate", "none") != "none":
            return False
    return all(mods[i].out_features == mods[i + 2].in_features for i in range(0, len(mods) - 2, 2))


class FusedSequential(torch.nn.Sequential):
    """The reference's Linear/GELU Sequential with a fused forward.  Holds the original sub-modules under their original indices."""

    def __init__(self, seq):
        super().__init__(*list(seq))
        lin = [m for m in self if isinstance(m, torch.nn.Linear)]
        self.dims = [lin[0].in_features] + [l.out_features for l in lin]
        self.n_layers = len(lin)
        self.grad_buffer = None
        self._input_grad_only = False
        self.allow_torch_fallback = True       # the reference must keep working for any widths it builds
        self.fused = forward_supported(self.dims)

    def forward(self, x):
        if not (self.fused and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32):
            return super().forward(x)
        lin = [m for m in self if isinstance(m, torch.nn.Linear)]
        return M._FusedMLPFunc.apply(self, x, *[l.weight for l in lin], *[l.bias for l in lin])

    def input_gradient_only(self):
        return M._InputGradOnly((self,))


def fuse_lipshitz(ref):
    """reference LipshitzMLP (models.py:54-129) -> permuto_sdf_amd.mlp.LipshitzMLP sharing its Parameters"""
    dims = [ref.layers[0].in_features] + [l.out_features for l in ref.layers]
    ours = M.LipshitzMLP(dims[0], dims[1:], ref.last_layer_linear)
    for i, l in enumerate(ref.layers):
        ours.layers[i].weight, ours.layers[i].bias = l.weight, l.bias
    ours.weights_per_layer = torch.nn.ParameterList([l.weight for l in ours.layers])
    ours.biases_per_layer = torch.nn.ParameterList([l.bias for l in ours.layers])
    ours.lipshitz_bound_per_layer = torch.nn.ParameterList(list(ref.lipshitz_bound_per_layer))
    ours.allow_torch_fallback = True
    ours.__dict__["_reference_module"] = ref      # not registered as a sub-module: unfuse_model() hands it back
    return ours


# ---- round 5: two more pieces of the reference's step behind its own objects --------------------------------------------------
def _fused_compute_weights(self, ray_samples_packed, sdf, gradients, cos_anneal_ratio, forced_variance=None):
    """VolumeRenderingNeus.compute_weights (volume_rendering_modules.py:129-172) with the ~30 elementwise launches between
    `inv_s` and `alpha` (and their ~60 autograd nodes) as ONE differentiable operator (csrc/neus.hip); the deviation network, the
    transmittance cumprod and the per-ray sum are the module's own sub-modules, untouched"""
    from .neus import neus_alpha
    inv_s = self.deviation_network(forced_variance)
    inv_s = inv_s.clip(1e-6, 1e6)
    self.last_inv_s = inv_s
    alpha, one_minus = neus_alpha(sdf, ray_samples_packed.samples_dirs, gradients, ray_samples_packed.samples_dt, inv_s,
                                  cos_anneal_ratio)
    transmittance, bg_transmittance = self.cumprod_alpha2transmittance_module(ray_samples_packed, one_minus)
    weights = (alpha * transmittance).view(-1, 1)
    weights_sum, _ = self.sum_ray_module(ray_samples_packed, weights)
    return weights, weights_sum, bg_transmittance, inv_s


def _fused_get_sdf_and_gradient(self, points, iter_nr, method="autograd"):
    """SDF.get_sdf_and_gradient (models.py:199-251), autograd branch: the same differentiated evaluation, but the inner
    torch.autograd.grad no longer computes -- and throws away -- the lattice and the MLP parameter gradients of the first-order
    pass (the encoding and the fused MLP are told that only the input gradient is wanted); twice differentiable as before"""
    if method != "autograd" or not hasattr(self.mlp_sdf, "input_gradient_only") or not hasattr(self.encoding, "positions_gradient_only"):
        return type(self).get_sdf_and_gradient(self, points, iter_nr, method)
    with torch.set_grad_enabled(True):
        points.requires_grad_(True)
        sdf, geom_feat = self.forward(points, iter_nr)
        d_output = torch.ones_like(sdf, requires_grad=False, device=sdf.device)
        with self.encoding.positions_gradient_only(), self.mlp_sdf.input_gradient_only():
            gradients = torch.autograd.grad(outputs=sdf, inputs=points, grad_outputs=d_output, create_graph=True,
                                            retain_graph=True, only_inputs=True)[0]
    return sdf, gradients, geom_feat


def fuse_model(model, verbose=False):
    """swap every direct child that is a Linear/GELU Sequential or a reference-style LipshitzMLP; returns the names swapped.
    PSDF_FUSE_REFERENCE_STEP=0 keeps the two round-5 additions off (NeuS weights of a `VolumeRenderingNeus` child as one operator,
    input-gradient-only inner pass of `SDF.get_sdf_and_gradient`)."""
    done = []
    step_level = os.environ.get("PSDF_FUSE_REFERENCE_STEP", "1") != "0"
    if step_level and type(model).__name__ == "SDF" and hasattr(model, "get_sdf_and_gradient") and "get_sdf_and_gradient" not in model.__dict__:
        import types
        model.__dict__["get_sdf_and_gradient"] = types.MethodType(_fused_get_sdf_and_gradient, model)
        done.append("get_sdf_and_gradient()")
    for name, child in list(model.named_children()):
        if step_level and type(child).__name__ == "VolumeRenderingNeus" and hasattr(child, "cumprod_alpha2transmittance_module"):
            # same object, same Parameters and state_dict keys: only the class it looks its method up in changes
            child.__class__ = type("FusedVolumeRenderingNeus", (type(child),), {"compute_weights": _fused_compute_weights,
                                                                               "_reference_class": type(child)})
            done.append(name + ".compute_weights()")
            continue
        if isinstance(child, (FusedSequential, M.LipshitzMLP, M.FusedMLP)):
            continue
        if isinstance(child, torch.nn.Sequential) and _is_linear_gelu_stack(child):
            setattr(model, name, FusedSequential(child))
            done.append(name)
        elif type(child).__name__ == "LipshitzMLP" and hasattr(child, "lipshitz_bound_per_layer") and hasattr(child, "layers"):
            dims = [child.layers[0].in_features] + [l.out_features for l in child.layers]
            if forward_supported(dims) and child.last_layer_linear:
                setattr(model, name, fuse_lipshitz(child))
                done.append(name)
    if verbose and done:
        print("[permuto_sdf_amd] fused evaluators behind %s: %s" % (type(model).__name__, ", ".join(done)), file=sys.stderr)
    return done


def unfuse_model(model):
    """undo fuse_model (same Parameters again): for A/B comparisons of the fused and the torch evaluation of one set of weights"""
    done = []
    if "get_sdf_and_gradient" in model.__dict__:
        del model.__dict__["get_sdf_and_gradient"]
        done.append("get_sdf_and_gradient()")
    for name, child in list(model.named_children()):
        if hasattr(type(child), "_reference_class"):
            child.__class__ = type(child)._reference_class
            done.append(name + ".compute_weights()")
        elif isinstance(child, FusedSequential):
            setattr(model, name, torch.nn.Sequential(*list(child)))
            done.append(name)
        elif isinstance(child, M.LipshitzMLP) and "_reference_module" in child.__dict__:
            setattr(model, name, child.__dict__["_reference_module"])
            done.append(name)
    return done


# ---------------------------------------------------------------------------------------------- opt-in, automatic
_CLASSES = ("SDF", "RGB", "NerfHash")


def enabled():
    return os.environ.get("PSDF_FUSE_REFERENCE_MLPS") == "1"


def owner_under_construction():
    """Called from PermutoEncoding.__init__: the reference constructs its encoding inside `SDF.__init__` / `RGB.__init__` /
    `NerfHash.__init__` (models.py:149,333,442), BEFORE the MLPs exist.  Find that model object on the Python stack and hand
    back a weak reference; the encoding fuses its owner at its first forward call (the models call the encoding first and the
    MLP after it).  Independent of import order -- nothing of the reference is patched."""
    import weakref
    f = sys._getframe(2)
    depth = 0
    while f is not None and depth < 8:
        owner = f.f_locals.get("self")
        if isinstance(owner, torch.nn.Module) and type(owner).__name__ in _CLASSES and f.f_code.co_name == "__init__":
            return weakref.ref(owner)
        f = f.f_back
        depth += 1
    return None
