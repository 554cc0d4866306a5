"""CPU ORACLE for one whole step of the hot path.  TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke()).

Strings the three oracles together in the order the reference's Python strings the operators together
(permuto_sdf_py/train_permuto_sdf.py:111-169 run_net: SDF evaluation -> NeuS weights -> integrate; loss
permuto_sdf_py/utils/permuto_sdf_utils.py:43-47):

  oracle/permuto_oracle.py  encode (vectorised torch restatement, PARITY UNPINNED, see its header)
  torch.nn.Sequential       Linear/GELU x3 + Linear, the unmodified modules of models.py:153-161 in fp32
  oracle/neus_oracle.py     section-point opacity, transmittance, weights, integration (incl. the reference's backward quirk), L1

and differentiates the L1 radiance loss with torch autograd.  `reference_step` takes the PARAMETERS of a
permuto_sdf_amd.hotpath.SdfHotPath as CPU tensors (it never imports the product) and returns what the HIP path is compared with.
"""
import torch

from . import neus_oracle as no
from . import permuto_oracle as po


def reference_step(pos, dirs, normals, dt, rgb, gt, nr_rays, per_ray, lattice, scale_per_level, shifts, window, weights, biases,
                   inv_s, cos_anneal_ratio, points_scaling=1e-3, reference_compat=True, threads=None):
    """All arguments CPU fp32 tensors ([N,3] pos / dirs / normals / rgb, [N,1] dt, [R,3] gt, lattice [L,T,F], shifts [L,3],
    window [L], weights / biases of the 4 linear layers).  Returns a dict: sdf [N,1], pred [R,3], loss (float), g_lattice [L,T,F],
    g_weights [4], g_biases [4]."""
    if threads:
        torch.set_num_threads(threads)
    lat = lattice.detach().clone().requires_grad_(True)
    feat = po.encode(pos, lat, scale_per_level, shifts, window, True, points_scaling)
    n = len(weights)
    mods = []
    for i, (w, b) in enumerate(zip(weights, biases)):
        lin = torch.nn.Linear(w.shape[1], w.shape[0])
        lin.weight.data.copy_(w)
        lin.bias.data.copy_(b)
        mods.append(lin)
        if i < n - 1:
            mods.append(torch.nn.GELU())
    mlp = torch.nn.Sequential(*mods)
    sdf = mlp(feat)
    alpha, om = no.neus_alpha(sdf, dirs, normals, dt, inv_s, cos_anneal_ratio)
    pred, w_, T_ = no.composite_equal(alpha, om, rgb, nr_rays, per_ray, reference_compat=reference_compat)
    loss = no.rgb_loss(gt, pred, torch.ones(nr_rays, 1))
    loss.backward()
    lin = [m for m in mlp if isinstance(m, torch.nn.Linear)]
    return dict(feat=feat.detach(), sdf=sdf.detach(), pred=pred.detach(), loss=float(loss), g_lattice=lat.grad,
                g_weights=[m.weight.grad for m in lin], g_biases=[m.bias.grad for m in lin], alpha=alpha.detach(),
                weights=w_.detach())


def reference_forward(pos, dirs, normals, dt, rgb, gt, nr_rays, per_ray, lattice, scale_per_level, shifts, window, weights, biases,
                      inv_s, cos_anneal_ratio, points_scaling=1e-3, dtype=torch.float64, chunk_rays=1024, device="cpu"):
    """Forward only (sdf [N,1], radiance [R,3], loss) of the same chain in `dtype`, ray chunk by ray chunk, so that the FULL bench
    batch (16 384 rays x 128) fits a CPU in about a minute.  dtype = float64 is the ARBITER of the full-size parity test: both
    the HIP path and an fp32 evaluation of the reference arithmetic are measured against it.  (The simplex of a point is found
    from the same fp32 positions; the encoding is continuous across simplex boundaries, so a different rounding of the
    elevated coordinates moves features by O(eps), never by a jump.)  device: where torch evaluates the restatement -- the same
    elementwise torch expressions on "cuda" finish the full batch in seconds (float64 on the GPU is still the oracle's arithmetic,
    not the product's kernels)."""
    lat = lattice.detach().to(device=device, dtype=dtype)
    sh = shifts.detach().to(device=device, dtype=dtype)
    win = torch.as_tensor(window).to(device=device, dtype=dtype)
    pos, dirs, normals, dt, rgb, gt = (t.to(device) for t in (pos, dirs, normals, dt, rgb, gt))
    mods = []
    for i, (w, b) in enumerate(zip(weights, biases)):
        lin = torch.nn.Linear(w.shape[1], w.shape[0]).to(dtype)
        lin.weight.data.copy_(w.to(dtype))
        lin.bias.data.copy_(b.to(dtype))
        mods.append(lin)
        if i < len(weights) - 1:
            mods.append(torch.nn.GELU())
    mlp = torch.nn.Sequential(*mods).to(device)
    inv_s = torch.as_tensor(inv_s).to(device=device, dtype=dtype)
    sdfs, preds = [], []
    with torch.no_grad():
        for r0 in range(0, nr_rays, chunk_rays):
            r1 = min(nr_rays, r0 + chunk_rays)
            s = slice(r0 * per_ray, r1 * per_ray)
            feat = po.encode(pos[s].to(dtype), lat, scale_per_level, sh, win, True, points_scaling)
            sdf = mlp(feat)
            alpha, om = no.neus_alpha(sdf, dirs[s].to(dtype), normals[s].to(dtype), dt[s].to(dtype), inv_s, cos_anneal_ratio)
            pred, _, _ = no.composite_equal(alpha, om, rgb[s].to(dtype), r1 - r0, per_ray, reference_compat=False)
            sdfs.append(sdf)
            preds.append(pred)
    sdf, pred = torch.cat(sdfs), torch.cat(preds)
    loss = no.rgb_loss(gt.to(dtype), pred, torch.ones(nr_rays, 1, dtype=dtype, device=device))
    return dict(sdf=sdf.cpu(), pred=pred.cpu(), loss=float(loss))
