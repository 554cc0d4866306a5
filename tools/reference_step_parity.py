#!/usr/bin/env python3
"""BASELINE config 4, one optimisation step, THREE ways on identical weights / rays / random streams:

  reference : the reference's own step -- the lines `TIME_START("run_net")` .. `loss+=loss_mask*...` of train()
              (permuto_sdf_py/train_permuto_sdf.py:338-383), cut out of the UNMODIFIED source file by their markers, executed
              with the reference's own `run_net`, model classes (`SDF`, `RGB`, `NerfHash`, `Colorcal`: torch.nn.Sequential MLPs,
              the reference's autograd Functions of volume_rendering_funcs.py) over this repository's drop-in operators,
              UNFUSED (PSDF_FUSE_REFERENCE_MLPS off), differentiated by `loss.backward()` -- torch autograd;
  manual    : permuto_sdf_amd.train_manual.ManualTrainer.step (hand-written forward + backward over the raw kernels);
  autograd  : permuto_sdf_amd.train_step.Trainer.step (torch autograd over the fused operators).

Compared: the loss, every dense parameter gradient, the three lattice gradients -- as the optimiser is about to see them.
Also the sphere-initialisation step (train_permuto_sdf.py:322-330: the reference's own `loss_sphere_init`).

Every comparison is made twice: with each trainer's OWN foreground samples (the whole step, sampling included) and with the
reference's foreground samples handed to the trainer (`*_same_samples`: only the step itself differs).  And the reference is run
against ITSELF with the hidden units of its SDF MLP re-numbered (the same function, another fp32 summation order):
`reference_self_noise` is the size of the reference's own rounding noise at that state -- the importance samples follow the SDF
values, the finest lattice levels have cells of 1e-4, so last-bit differences of the SDF move samples across cells.

What is shared: the networks' state (written in the reference's checkpoint layout and loaded by the reference's classes),
the occupancy grid, the rays (`PermutoSDF.random_rays_from_reel` drawn once), torch's generators (seeded alike before each
run: the curvature term's `randn_like` and `rand_points_inside` draw the same numbers) and the three PCG32 jitter generators
of the samplers.  What is NOT shared: everything downstream -- each runner makes its own samples, importance samples, networks
evaluations and gradients.

The reference checkout is found at $PSDF_REFERENCE, /root/reference or <repo>/_refcopy.  Prints one JSON object.

    python tools/reference_step_parity.py [--out FILE] [--modes early,late,mask,sphere]
"""
import argparse
import contextlib
import importlib.util
import json
import os
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find_reference():
    for c in (os.environ.get("PSDF_REFERENCE"), "/root/reference", os.path.join(ROOT, "_refcopy")):
        if c and os.path.isdir(os.path.join(c, "permuto_sdf_py")):
            return c
    return None


def setup_paths(ref):
    for p in (ref, os.path.join(ROOT, "compat"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    spec = importlib.util.spec_from_file_location("psdf_compat_sitecustomize", os.path.join(ROOT, "compat", "sitecustomize.py"))
    spec.loader.exec_module(importlib.util.module_from_spec(spec))


def extract_loss_block(path):
    """the reference's step between `TIME_START("run_net")` and the no-grad bookkeeping block that follows the losses
    (`with torch.set_grad_enabled(False):` + `#update occupancy`): source text, first and last line number (1-based)"""
    src = open(path).read().splitlines()
    i0 = next(i for i, l in enumerate(src) if l.strip() == 'TIME_START("run_net")')
    i1 = next(i for i in range(i0, len(src) - 1)
              if src[i].strip() == "with torch.set_grad_enabled(False):" and "#update occupancy" in src[i + 1])
    block = textwrap.dedent("\n".join(src[i0:i1]))
    assert "run_net(args, hyperparams" in block and "rgb_loss(gt_selected, pred_rgb" in block and "loss_mask" in block
    return block, i0 + 1, i1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--modes", default="early,late,mask,sphere")
    ap.add_argument("--sphere-iters", type=int, default=200, help="sphere-initialisation iterations before the snapshot")
    ap.add_argument("--warm-iters", type=int, default=17, help="main-phase iterations before the snapshot (>= 9: one grid refresh)")
    a = ap.parse_args()
    ref = find_reference()
    if ref is None:
        print(json.dumps({"skipped": "no reference checkout (PSDF_REFERENCE, /root/reference, <repo>/_refcopy)"}))
        return
    os.environ.pop("PSDF_FUSE_REFERENCE_MLPS", None)          # the reference's evaluators stay torch.nn
    setup_paths(ref)
    import torch
    assert torch.cuda.is_available(), "needs the GPU"
    dev = torch.device("cuda", 0)
    from permuto_sdf_amd import checkpoint, parallel
    from permuto_sdf_amd.bridge import OccupancyGrid, PermutoSDF, RaySampler, VolumeRendering
    from permuto_sdf_amd.train_manual import ManualTrainer
    from permuto_sdf_amd.train_step import HyperParams, SyntheticReel, Trainer

    # ---- our trainers (built under torch's normal defaults), a short schedule up to a snapshot
    def make(cls, with_mask):
        hp = HyperParams()
        hp.nr_iter_sphere_fit = a.sphere_iters
        return cls(dev, hp=hp, seed=0, reference_schedule=True, nr_images=12, with_mask=with_mask)

    reel = SyntheticReel(dev, nr_images=12, height=200, width=300)
    reel.mask_reel = (torch.rand(reel.mask_reel.shape, generator=torch.Generator().manual_seed(5)) > 0.4).float().to(dev)
    pcgs = (OccupancyGrid._rng, RaySampler._rng, VolumeRendering._rng)

    import permuto_sdf_py.train_permuto_sdf as T          # module level: torch.manual_seed(0), default tensor type -> cuda

    @contextlib.contextmanager
    def default_tensor(cuda):
        torch.set_default_tensor_type(torch.cuda.FloatTensor if cuda else torch.FloatTensor)
        try:
            yield
        finally:
            torch.set_default_tensor_type(torch.cuda.FloatTensor)

    block, l0, l1 = extract_loss_block(os.path.join(ref, "permuto_sdf_py", "train_permuto_sdf.py"))
    code = compile(block, "<reference train_permuto_sdf.py:%d-%d>" % (l0, l1), "exec")
    out = {"reference": ref, "reference_block_lines": [l0, l1], "device": torch.cuda.get_device_name(0), "cases": {}}

    def rel(x, y):
        x, y = x.detach().double().flatten(), y.detach().double().flatten()
        d = (x - y).abs()
        return {"max_rel": float(d.max() / y.abs().max().clamp_min(1e-30)), "l2_rel": float(d.norm() / y.norm().clamp_min(1e-30)),
                "ref_absmax": float(y.abs().max())}

    def our_named_grads(tr):
        names = {}
        for prefix, m in (("sdf", tr.sdf), ("rgb", tr.rgb), ("bg", tr.bg), ("colorcal", tr.colorcal)):
            if m is not None:
                for k, p in m.named_parameters():
                    names[id(p)] = prefix + "." + k
        cap = tr.capture_grads
        buffered = {id(m.encoding.lattice_values) for m in (tr.sdf, tr.rgb, tr.bg)}
        dense = [p for p in tr.params if id(p) not in buffered]
        g = {names[id(p)]: gr for p, gr in zip(dense, cap["dense"])}
        for m, gr in zip((tr.sdf, tr.rgb, tr.bg), cap["lattices"]):
            g[names[id(m.encoding.lattice_values)]] = gr
        return g, float(cap["loss"])

    def ref_named_grads(models):
        g = {}
        for kind, m in models.items():
            if m is None:
                continue
            d = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters() if p.requires_grad}
            if kind != "colorcal":
                d = checkpoint.from_reference_keys(kind, d)
            for k, v in d.items():
                g[kind + "." + k] = v
        return g

    for with_mask in (False, True):
        modes = [m for m in a.modes.split(",") if (m == "mask") == with_mask]
        if not modes:
            continue
        with default_tensor(False):
            trm, tra = make(ManualTrainer, with_mask), make(Trainer, with_mask)
            reel.has_mask = with_mask
            for p in pcgs:
                p.__init__()
            for _ in range(a.sphere_iters + a.warm_iters):
                trm.step(reel)
            if hasattr(trm, "_drop_prefetch"):     # the last step has issued the NEXT step's sampling ahead of time: undo it (the
                trm._drop_prefetch()               # jitter generators return to their state before it) before they are snapshotted
            n0 = a.sphere_iters
            # ---- snapshot: networks (reference checkpoint layout), grid, ray count, generators
            state = {k: {n: v.detach().clone() for n, v in m.state_dict().items()}
                     for k, m in (("sdf", trm.sdf), ("rgb", trm.rgb), ("bg", trm.bg), ("colorcal", trm.colorcal))}
            grid_v, grid_o = trm.grid.get_grid_values().clone(), trm.grid.get_grid_occupancy().clone()
            pcg_state = [(p.state, p.inc) for p in pcgs]
            nr_rays = trm.nr_rays
            parallel.seed_generators(12345, dev)
            o, d, gt, mask, img_idx = PermutoSDF.random_rays_from_reel(reel, nr_rays)
            _, _, _, _, hit = trm.sphere.ray_intersection(o, d)
            rays = (o, d, gt, mask, img_idx, hit)

        def restore(tr):
            for k, m in (("sdf", tr.sdf), ("rgb", tr.rgb), ("bg", tr.bg), ("colorcal", tr.colorcal)):
                m.load_state_dict(state[k])
            tr.grid.set_grid_values(grid_v.clone())
            tr.grid.set_grid_occupancy(grid_o.clone())
            for p, (s, i) in zip(pcgs, pcg_state):
                p.state, p.inc = s, i
            tr.nr_rays = nr_rays
            tr._late_seen = False
            tr._draw_rays = lambda reel_: (o, d, gt, hit, img_idx, mask)
            for t in tr.touched:
                t.grad.zero_()
                t.touched.zero_()
            for gb in tr.grad_buffers:
                gb.zero()

        def run_ours(tr, git, shared_fg=None):
            """one step of `tr` from the snapshot -> (named gradients, loss, tr.last); shared_fg: a foreground sample container to
            use INSTEAD of the trainer's own (the reference's), so that only the step itself differs, not the sample positions"""
            restore(tr)
            tr.iter = git
            tr.capture_grads = {}
            orig = tr._samples
            seen = {}

            def patched(o_, d_, it_, jitter=True, **kw):
                fg_own, bg_own = orig(o_, d_, it_, jitter, **kw)
                seen["fg"] = fg_own
                return (shared_fg if shared_fg is not None else fg_own), bg_own
            tr._samples = patched
            try:
                tr.step(reel)
            finally:
                del tr._samples
            torch.cuda.synchronize()
            g, l = our_named_grads(tr)
            tr.capture_grads = None
            return g, l, dict(tr.last), seen.get("fg")

        def run_reference(mode, it, git, seed, permute_seed=None, forced_fg=None, arbiter=None):
            """the reference's own step from the snapshot.  permute_seed: re-number the hidden units of the reference's SDF MLP
            (rows of a Linear and the matching columns of the next one: the SAME function, a different fp32 summation order) --
            the distance between two such runs is the reference's own rounding noise at this state.  forced_fg: the foreground
            samples of an earlier run, returned by `importance_sampling_sdf_model` in place of its own result (the perturbed SDF
            would otherwise move the importance samples: with it the noise of the STEP is measured apart from the samplers')."""
            for p_, (s_, i_) in zip(pcgs, pcg_state):
                p_.state, p_.inc = s_, i_
            aabb = T.create_bb_for_dataset("dtu")
            hp = T.hyperparams
            model_sdf = T.SDF(in_channels=3, boundary_primitive=aabb, geom_feat_size_out=hp.sdf_geom_feat_size,
                              nr_iters_for_c2f=hp.sdf_nr_iters_for_c2f).to("cuda")
            model_rgb = T.RGB(in_channels=3, boundary_primitive=aabb, geom_feat_size_in=hp.sdf_geom_feat_size,
                              nr_iters_for_c2f=hp.rgb_nr_iters_for_c2f).to("cuda")
            model_bg = T.NerfHash(4, boundary_primitive=aabb, nr_iters_for_c2f=hp.background_nr_iters_for_c2f).to("cuda")
            model_colorcal = T.Colorcal(reel.rgb_reel.shape[0], 0)
            assert type(model_sdf.mlp_sdf) is torch.nn.Sequential, "the reference's evaluators must stay unfused here"
            model_sdf.load_state_dict(checkpoint.to_reference_keys("sdf", state["sdf"]))
            model_rgb.load_state_dict(checkpoint.to_reference_keys("rgb", state["rgb"]))
            model_bg.load_state_dict(checkpoint.to_reference_keys("bg", state["bg"]))
            model_colorcal.load_state_dict(state["colorcal"])
            perms = None
            if permute_seed is not None:
                lins = [m for m in model_sdf.mlp_sdf if isinstance(m, torch.nn.Linear)]
                gp = torch.Generator(device="cpu").manual_seed(permute_seed)
                perms = [torch.randperm(l.out_features, generator=gp, device="cpu").to(dev) for l in lins[:-1]]
                with torch.no_grad():
                    for i, pm in enumerate(perms):
                        lins[i].weight.copy_(lins[i].weight[pm].clone())
                        lins[i].bias.copy_(lins[i].bias[pm].clone())
                        lins[i + 1].weight.copy_(lins[i + 1].weight[:, pm].clone())
            occupancy_grid = T.OccupancyGrid(256, 1.0, [0, 0, 0])
            occupancy_grid.set_grid_values(grid_v.clone())
            occupancy_grid.set_grid_occupancy(grid_o.clone())
            for m in (model_sdf, model_rgb, model_bg):
                m.train(True)
            parallel.seed_generators(seed, dev)
            models = {"sdf": model_sdf, "rgb": model_rgb, "bg": model_bg, "colorcal": model_colorcal}
            hook = None
            if arbiter is not None:
                # FLOAT64 ARBITER of the one gradient that is a plain sum of per-sample terms and nothing else: the bias of the SDF
                # net's LAST layer, db = sum over every evaluation and every sample of the upstream gradient of the net's output
                # (the analytic normal n = d sdf / d x does not depend on that bias, so the eikonal / curvature path adds nothing to
                # it).  The reference's own per-sample terms (fp32, as its autograd produces them) are summed in float64: what an
                # exact accumulation of the reference's step gives.  In the `late` state these are ~49 000 cancelling NeuS terms
                # and the fp32 sum is where the reference, its re-numbered self and our trainers part in the 5th digit.
                last = [m_ for m_ in model_sdf.mlp_sdf if isinstance(m_, torch.nn.Linear)][-1]
                arbiter["sum"] = torch.zeros(last.out_features, dtype=torch.float64, device=dev)
                arbiter["terms"] = 0

                def fwd_hook(mod, inp, out_):
                    if out_.requires_grad:
                        def on_grad(g_):
                            if not arbiter.get("active"):      # (the analytic normal's inner autograd.grad passes through here too)
                                return
                            arbiter["sum"] += g_.detach().double().reshape(-1, g_.shape[-1]).sum(0)
                            arbiter["terms"] += int(g_.numel() // g_.shape[-1])
                        out_.register_hook(on_grad)
                hook = last.register_forward_hook(fwd_hook)
            terms, fg = {}, None
            if mode == "sphere":
                loss, loss_sdf, loss_eik = T.loss_sphere_init("dtu", 30000, aabb, model_sdf, git)    # :323
                terms = {"loss_sdf": float(loss_sdf), "loss_eikonal": float(loss_eik)}
            else:
                args = argparse.Namespace(with_mask=with_mask, dataset="dtu")
                ns = dict(vars(T))
                loss0, loss_rgb, loss_eikonal, loss_curvature, loss_lipshitz = T.init_losses()
                ns.update(args=args, hyperparams=hp, ray_origins=o, ray_dirs=d, img_indices=img_idx, gt_selected=gt, gt_mask=mask,
                          does_ray_intersect_box=hit, model_sdf=model_sdf, model_rgb=model_rgb, model_bg=model_bg,
                          model_colorcal=model_colorcal, occupancy_grid=occupancy_grid, iter_nr_for_anneal=it,
                          cos_anneal_ratio=T.map_range_val(it, 0.0, hp.forced_variance_finish_iter, 0.0, 1.0),
                          forced_variance=T.map_range_val(it, 0.0, hp.forced_variance_finish_iter, 0.3, hp.forced_variance_finish),
                          loss=loss0, loss_rgb=loss_rgb, loss_eikonal=loss_eikonal, loss_curvature=loss_curvature,
                          loss_lipshitz=loss_lipshitz)
                old_eik = hp.eikonal_weight
                orig_imp = T.importance_sampling_sdf_model
                if forced_fg is not None:        # (run_net looks the name up in the reference module's own globals)
                    T.importance_sampling_sdf_model = lambda *a_, **k_: forced_fg
                try:
                    exec(code, ns)
                finally:
                    T.importance_sampling_sdf_model = orig_imp
                assert hp.eikonal_weight == old_eik
                loss = ns["loss"]
                for k in ("loss_rgb", "loss_eikonal", "loss_curvature", "loss_offsurface_high_sdf", "loss_mask"):
                    if k in ns and torch.is_tensor(ns[k]):
                        terms[k] = float(ns[k].mean())
                fg = ns["fg_ray_samples_packed"]
                terms["nr_fg_samples"] = int(fg.samples_pos.shape[0])
            if arbiter is not None:
                arbiter["active"] = True
            loss.backward()
            torch.cuda.synchronize()
            if hook is not None:
                hook.remove()
            if perms is not None:     # gradients back in the original numbering of the hidden units
                lins = [m for m in model_sdf.mlp_sdf if isinstance(m, torch.nn.Linear)]
                for i, pm in enumerate(perms):
                    inv = torch.empty_like(pm)
                    inv[pm] = torch.arange(pm.numel(), device=dev)
                    lins[i].weight.grad = lins[i].weight.grad[inv].clone()
                    lins[i].bias.grad = lins[i].bias.grad[inv].clone()
                    lins[i + 1].weight.grad = lins[i + 1].weight.grad[:, inv].clone()
            return ref_named_grads(models), float(loss), terms, fg

        def compare(g, gref):
            rows = {}
            for k in sorted(gref):
                if k not in g:
                    rows[k] = {"missing": True}
                    continue
                if float(gref[k].abs().max()) == 0.0 and float(g[k].abs().max()) == 0.0:
                    continue          # no gradient on either side (forced variance, colour nets in the sphere phase ...)
                rows[k] = rel(g[k], gref[k])
                if k.endswith("encoding.lattice_values"):      # per level: fine levels turn position noise into gradient noise
                    rows[k]["per_level_max_rel"] = [
                        float((g[k][l].double() - gref[k][l].double()).abs().max() / gref[k][l].double().abs().max().clamp_min(1e-30))
                        if float(gref[k][l].abs().max()) > 0 else 0.0 for l in range(gref[k].shape[0])]
            return {"grads": rows, "not_in_reference": sorted(set(g) - set(gref)),
                    "worst_dense": max((v["max_rel"] for k, v in rows.items() if "lattice" not in k and "max_rel" in v), default=0.0),
                    "worst_lattice": max((v["max_rel"] for k, v in rows.items() if "lattice" in k and "max_rel" in v), default=0.0),
                    "worst_lattice_l2": max((v["l2_rel"] for k, v in rows.items() if "lattice" in k and "l2_rel" in v), default=0.0)}

        for mode in modes:
            it = {"early": 2001, "late": 52001, "mask": 2001, "sphere": None}[mode]
            git = (n0 + it) if it is not None else 101        # neither is a multiple of 8: no grid refresh inside the step
            assert git % 8 != 0
            seed = parallel.step_seed(trm._seed, 0, git)
            with default_tensor(True):
                arb = {}
                gref, loss_ref, terms, fg_ref = run_reference(mode, it, git, seed, arbiter=arb)
                gref2, loss_ref2, _, _ = run_reference(mode, it, git, seed, permute_seed=7)
                gref4 = None
                draws = []       # hidden units re-numbered AND the first run's samples: rounding noise of the step alone, SIX draws
                if fg_ref is not None:
                    for ps in (11, 12, 13, 14, 15, 16):
                        g4, l4, _, _ = run_reference(mode, it, git, seed, permute_seed=ps, forced_fg=fg_ref)
                        draws.append((g4, l4))
                    gref4, loss_ref4 = draws[0]
                # the ENSEMBLE of the reference's own evaluations on these samples (the run itself + the six re-numberings), dense
                # tensors only (the MLP parameters: small): a cloud of equally valid fp32 roundings of the same gradients.  Its
                # diameter per tensor (largest distance between two members) and, below, OUR distance to the nearest member are
                # what `tests/test_gpu_reference_step.py` holds the same-samples gradients to where 1e-4 of the single run is not
                # met: ONE run is one draw of that cloud, and so is the maximum of six distances from it (the late state's last
                # SDF bias: 2e-5 .. 1e-4 from run to run, float atomics upstream) -- a bar made of one draw fails ~1 run in 8.
                ensemble = None
                if draws:
                    dense_keys = [k for k in gref if "lattice" not in k]
                    ensemble = [{k: gd[k].detach().double().clone() for k in dense_keys if k in gd} for gd in [gref] + [d_[0] for d_ in draws]]
                repeats = []                                                         # the same run again, three times
                for _ in range(3):
                    g3, l3, _, _ = run_reference(mode, it, git, seed)
                    repeats.append(dict(compare(g3, gref), loss_rel=abs(l3 - loss_ref) / abs(loss_ref)))
                    del g3
            case = {"iter_nr_for_anneal": it, "global_iter": git, "nr_rays": int(o.shape[0]), "reference_loss": loss_ref,
                    "reference_terms": terms,
                    # the reference against ITSELF with the hidden units of its SDF MLP re-numbered: its own rounding noise
                    "reference_self_noise": dict(compare(gref2, gref), loss_rel=abs(loss_ref2 - loss_ref) / abs(loss_ref)),
                    # ... and REPEATED unchanged (same samples by construction): what float atomics alone do to its gradients
                    "reference_repeat_noise": {k: max(r[k] for r in repeats)
                                               for k in ("worst_dense", "worst_lattice", "worst_lattice_l2", "loss_rel")}}
            if gref4 is not None:
                # one re-numbering is ONE draw of the noise: six are made and the LARGEST deviation per tensor is what the
                # tests use as the reference's own rounding noise on these samples (no constant measured on another day)
                cmps = [dict(compare(g4, gref), loss_rel=abs(l4 - loss_ref) / abs(loss_ref)) for g4, l4 in draws]
                worst = dict(cmps[0])
                for k in ("worst_dense", "worst_lattice", "worst_lattice_l2", "loss_rel"):
                    worst[k] = max(c_[k] for c_ in cmps)
                worst["by_tensor_max"] = {k: max(c_["grads"][k]["max_rel"] for c_ in cmps if k in c_["grads"] and "max_rel" in c_["grads"][k])
                                          for k in cmps[0]["grads"] if "max_rel" in cmps[0]["grads"][k]}
                worst["draws_worst_dense"] = [c_["worst_dense"] for c_ in cmps]
                diam = {}
                for k in ensemble[0]:
                    sc_ = float(ensemble[0][k].abs().max())
                    if sc_ == 0.0:
                        continue
                    diam[k] = max(float((ensemble[i_][k] - ensemble[j_][k]).abs().max()) / sc_
                                  for i_ in range(len(ensemble)) for j_ in range(i_ + 1, len(ensemble)) if k in ensemble[i_] and k in ensemble[j_])
                worst["ensemble_members"] = len(ensemble)
                worst["ensemble_diameter_by_tensor"] = diam
                case["reference_self_noise_same_samples"] = worst
            # the float64 arbiter of the SDF net's last bias (see run_reference): the reference's own fp32 gradient against it ...
            ARB = "sdf.mlp_sdf.layers.%d.bias" % (len([k for k in gref if k.startswith("sdf.mlp_sdf.layers.") and k.endswith(".bias")]) - 1)
            assert ARB in gref and gref[ARB].numel() == arb["sum"].numel(), (ARB, sorted(gref))
            f64 = arb["sum"]
            scale64 = float(f64.abs().max())

            def to_f64(gd):
                return float((gd[ARB].detach().double().reshape(-1) - f64).abs().max() / max(scale64, 1e-300))
            case["arbiter_last_sdf_bias"] = {"tensor": ARB, "terms_summed": arb["terms"], "f64_absmax": scale64,
                                             "sum_of_abs_terms_note": "float64 sum of the reference's own fp32 per-sample upstream "
                                                                      "gradients of the SDF net's output (all evaluations of the step)",
                                             "reference_vs_f64": to_f64(gref), "reference_renumbered_vs_f64": to_f64(gref2)}
            if gref4 is not None:
                per_draw = [to_f64(g4) for g4, _ in draws]
                case["arbiter_last_sdf_bias"]["reference_renumbered_same_samples_vs_f64"] = per_draw
                case["arbiter_last_sdf_bias"]["reference_renumbered_same_samples_vs_f64_max"] = max(per_draw)
                del gref4, draws
            case["reference_repeat_noise"]["dense_by_tensor"] = {
                k: max(r["grads"][k]["max_rel"] for r in repeats if k in r["grads"] and "max_rel" in r["grads"][k])
                for k in repeats[0]["grads"] if "lattice" not in k and "max_rel" in repeats[0]["grads"][k]}
            del gref2
            with default_tensor(False):
                for name, tr in (("manual", trm), ("autograd", tra)):
                    for variant in (("", None),) + ((("_same_samples", fg_ref),) if fg_ref is not None else ()):
                        g, l, last, fg_own = run_ours(tr, git, shared_fg=variant[1])
                        case[name + variant[0]] = dict(compare(g, gref), loss=l, loss_rel=abs(l - loss_ref) / abs(loss_ref),
                                                       nr_fg_samples=last.get("nr_fg_samples"))
                        if variant[1] is not None:     # ... and ours on the SAME samples against the same float64 sum
                            case["arbiter_last_sdf_bias"][name + "_same_samples_vs_f64"] = to_f64(g)
                            if ensemble is not None:   # distance to the NEAREST member of the reference's ensemble, per dense tensor
                                near = {}
                                for k in ensemble[0]:
                                    sc_ = float(ensemble[0][k].abs().max())
                                    if sc_ == 0.0 or k not in g:
                                        continue
                                    gk = g[k].detach().double()
                                    near[k] = min(float((gk - m_[k]).abs().max()) / sc_ for m_ in ensemble if k in m_)
                                case[name + variant[0]]["nearest_reference_member_by_tensor"] = near
                        if variant[1] is None and fg_ref is not None and fg_own is not None:
                            # how far this trainer's OWN foreground samples are from the reference's (same rays, same jitter
                            # streams; the importance samples follow each side's own SDF evaluations)
                            a_, b_ = fg_own.samples_z.reshape(-1), fg_ref.samples_z.reshape(-1)
                            st = {"same_count": bool(a_.numel() == b_.numel()),
                                  "same_ranges": bool(torch.equal(fg_own.ray_start_end_idx, fg_ref.ray_start_end_idx))}
                            if st["same_count"]:
                                dz = (a_ - b_).abs()
                                st.update(identical=int((dz == 0).sum()), of=int(dz.numel()), max_abs_dz=float(dz.max()),
                                          above_1e_6=int((dz > 1e-6).sum()), above_1e_4=int((dz > 1e-4).sum()))
                            case[name + variant[0]]["own_samples_vs_reference"] = st
                        del g
            out["cases"][mode] = case
            del gref
    s = json.dumps(out, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(s)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
