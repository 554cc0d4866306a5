#!/usr/bin/env python
"""Pin the encoding: dump golden vectors from the REAL upstream `permutohedral_encoding` package.

WHY.  The arithmetic of the permutohedral hash encoding lives in github.com/RaduAlexandru/permutohedral_encoding (CUDA, no
version pin; imported at permuto_sdf_py/models/models.py:20).  Its source is absent from this repository's build container, so
oracle/permuto_oracle.py and the HIP kernels freeze a few conventions by recollection (permuto_sdf_amd/csrc/
encode_conventions.h): PARITY UNPINNED.  Anyone with an NVIDIA GPU and the upstream package closes that in one command:

    python tools/dump_upstream_encoding_vectors.py            # writes tests/golden/upstream_encoding_vectors.npz (~3 MB)
    python -m pytest tests/test_upstream_vectors.py -q        # CPU: oracle vs upstream; -m gpu: HIP kernels vs upstream

With the file present the test compares forward, lattice / position gradients and the double backward with the upstream
numbers; on a mismatch it searches the convention grid and NAMES the combination that reproduces upstream, which is then one
environment variable (PSDF_ENC_CONVENTIONS=..., permuto_sdf_amd/conventions.py) or one edit of encode_conventions.h away.

WHAT IS DUMPED, per case (P in {3, 4} x concat_points in {False, True}; small tables so the file stays small):
  meta (JSON)            constructor arguments, upstream version / git hash if discoverable, state_dict keys -> shapes,
                         output_dims(), attribute names the script found the parameters under
  positions [N,P], window [L], scale_list [L]             inputs (numpy RandomState(seed): identical on every machine)
  lattice_values, random_shift_per_level                   upstream's OWN parameter tensors, raw layout (their init distribution
                                                           is part of what is being pinned), + scale_factor if exposed
  out [N,C]                                                forward
  grad_out [N,C] (input), grad_lattice, grad_positions     backward of sum(out * grad_out)
  dd_v [N,P] (input), dbl_grad_lattice, dbl_grad_gout      double backward: gradients of sum(d<out,grad_out>/dpositions * dd_v) with
                                                           respect to lattice_values and grad_out (models.py:245-251 needs it)
  c2f_t, c2f_window                                        Coarse2Fine(L)(t) for a few t

The script imports nothing from this repository except tests/golden/upstream_cases.py (pure numpy: the shared case list).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.append(ROOT)          # LAST: the upstream package must win over this repository's drop-in of the same name
from tests.golden.upstream_cases import CASES, C2F_TS, make_inputs   # noqa: E402


def find_tensor(module, needle):
    """first parameter / buffer / tensor attribute whose name contains `needle` -> (name, tensor) or (None, None)"""
    import torch
    for name, t in list(module.named_parameters()) + list(module.named_buffers()):
        if needle in name:
            return name, t
    for name, v in vars(module).items():
        if needle in name and isinstance(v, torch.Tensor):
            return name, v
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "upstream_encoding_vectors.npz"))
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args()
    import torch
    import permutohedral_encoding as permuto_enc
    if os.path.abspath(os.path.dirname(permuto_enc.__file__)) == os.path.join(ROOT, "permutohedral_encoding"):
        raise SystemExit("this is the drop-in package of THIS repository, not upstream: run with the upstream "
                         "permutohedral_encoding first on sys.path (and without this repository's root on PYTHONPATH)")
    dev = torch.device(args.device)
    out = {}
    meta = {"package_file": permuto_enc.__file__, "version": getattr(permuto_enc, "__version__", None),
            "torch": torch.__version__, "cases": {}}
    try:
        import subprocess
        meta["git"] = subprocess.run(["git", "-C", os.path.dirname(permuto_enc.__file__), "rev-parse", "HEAD"],
                                     capture_output=True, text=True).stdout.strip() or None
    except Exception:
        meta["git"] = None
    for case in CASES:
        name = case["name"]
        P, T, L_, F = case["pos_dim"], case["capacity"], case["nr_levels"], case["nr_feat"]
        inp = make_inputs(case)
        torch.manual_seed(case["seed"])
        enc = permuto_enc.PermutoEncoding(P, T, L_, F, inp["scale_list"], appply_random_shift_per_level=True,
                                          concat_points=case["concat_points"],
                                          concat_points_scaling=case["concat_points_scaling"]).to(dev)
        lat_name, lat = find_tensor(enc, "lattice_values")
        sh_name, sh = find_tensor(enc, "shift")
        sf_name, sf = find_tensor(enc, "scale_factor")
        cm = {"ctor": {k: case[k] for k in ("pos_dim", "capacity", "nr_levels", "nr_feat", "concat_points",
                                            "concat_points_scaling")},
              "state_dict": {k: list(v.shape) for k, v in enc.state_dict().items()},
              "output_dims": int(enc.output_dims()), "lattice_name": lat_name, "shift_name": sh_name, "scale_factor_name": sf_name}
        if sh is not None and sh.numel() == L_ * P:
            # tie probes: position = -shift[l] makes every elevated coordinate of level l exactly 0 -> all residuals tie there;
            # the forward cannot see the tie rule (the tied vertex has weight 0) but the position gradient can
            inp["positions"][:L_] = -sh.detach().cpu().numpy().reshape(L_, P).astype(np.float32)
            cm["tie_probes"] = L_
        pos = torch.from_numpy(inp["positions"]).to(dev).requires_grad_(True)
        win = torch.from_numpy(inp["window"]).to(dev)
        y = enc(pos, win)
        C = y.shape[1]
        g_out = torch.from_numpy(inp["grad_out_full"][:, :C].copy()).to(dev).requires_grad_(True)
        dd_v = torch.from_numpy(inp["dd_v"]).to(dev)
        (g_pos,) = torch.autograd.grad(y, pos, g_out, create_graph=True)
        (g_lat,) = torch.autograd.grad(y, lat, g_out, retain_graph=True)
        dbl = torch.autograd.grad((g_pos * dd_v).sum(), [lat, g_out], allow_unused=True)
        out.update({name + "/positions": inp["positions"], name + "/window": inp["window"], name + "/scale_list": inp["scale_list"],
                    name + "/lattice_values": lat.detach().cpu().numpy(), name + "/out": y.detach().cpu().numpy(),
                    name + "/grad_out": g_out.detach().cpu().numpy(), name + "/grad_lattice": g_lat.detach().cpu().numpy(),
                    name + "/grad_positions": g_pos.detach().cpu().numpy(), name + "/dd_v": inp["dd_v"]})
        if sh is not None:
            out[name + "/random_shift_per_level"] = sh.detach().cpu().numpy()
        if sf is not None:
            out[name + "/scale_factor"] = sf.detach().cpu().numpy()
        if dbl[0] is not None:
            out[name + "/dbl_grad_lattice"] = dbl[0].detach().cpu().numpy()
        if dbl[1] is not None:
            out[name + "/dbl_grad_gout"] = dbl[1].detach().cpu().numpy()
        meta["cases"][name] = cm
        print("%-22s out %s  lattice %s (%s)  shift %s  scale_factor %s" % (
            name, tuple(y.shape), tuple(lat.shape), lat_name, sh_name, sf_name))
    c2f = permuto_enc.Coarse2Fine(24)
    out["c2f_t"] = np.asarray(C2F_TS, np.float32)
    out["c2f_window"] = np.stack([c2f(float(t)).detach().cpu().numpy().reshape(-1) for t in C2F_TS]).astype(np.float32)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(args.out, **out)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
