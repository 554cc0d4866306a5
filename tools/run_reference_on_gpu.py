#!/usr/bin/env python3
"""Run the REFERENCE's own Python (permuto_sdf_py, unmodified) on MI355X through this repository's drop-in packages.

SURVEY.md 8f-1 / north_star: "permuto_sdf_py/models and train_permuto_sdf.py run unmodified on PyTorch-ROCm".  This
script imports the reference's training module as it is, shortens its schedule by SETTING ATTRIBUTES of its own
`hyperparams` object (no file is edited), injects a logging callback through its own callback list, and calls its own
`run()` with the command line of BASELINE config 4 (`--dataset dtu --scene dtu_scan24 --no_viewer`).  The dataset path
of `comp_1` does not exist offline, so `compat/dataloaders.DataLoaderDTU` synthesises a scene of the same shape.
After training it drives, with the models the reference built:
  * run_net               (train_permuto_sdf.py:111-169)  -> radiance compared with permuto_sdf_amd.train_step.Trainer._render
  * importance_sampling_sdf_model (sdf_utils.py:383-423)
  * sphere_trace          (sdf_utils.py:120-218), with and without occupancy grid -> compared with SphereTracer.trace
  * run_net_sphere_traced / run_net_in_chunks on a subsampled frame (train_permuto_sdf.py:172-242)
and writes a JSON log (losses, iterations/s, comparisons).

The reference checkout is found at $PSDF_REFERENCE, /root/reference, or <repo>/_refcopy (a git-ignored copy of
permuto_sdf_py/ + config/ that travels to the GPU box with gpurun; never committed).

    python tools/run_reference_on_gpu.py --sphere-iters 300 --train-iters 300 --out gpurun_out/r02/reference_run.json
"""
import argparse
import importlib.util
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find_reference():
    for c in (os.environ.get("PSDF_REFERENCE"), "/root/reference", os.path.join(ROOT, "_refcopy")):
        if c and os.path.isdir(os.path.join(c, "permuto_sdf_py")):
            return c
    raise SystemExit("reference checkout not found (set PSDF_REFERENCE or create <repo>/_refcopy)")


def setup_paths(ref):
    for p in (ref, os.path.join(ROOT, "compat"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    # compat/sitecustomize.py is loaded automatically only when compat/ is on PYTHONPATH at interpreter start
    spec = importlib.util.spec_from_file_location("psdf_compat_sitecustomize", os.path.join(ROOT, "compat", "sitecustomize.py"))
    spec.loader.exec_module(importlib.util.module_from_spec(spec))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sphere-iters", type=int, default=300, help="iterations of the reference's sphere-initialisation phase")
    ap.add_argument("--train-iters", type=int, default=300, help="iterations of the reference's main training phase")
    ap.add_argument("--log-every", type=int, default=25)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "reference_run.json"))
    ap.add_argument("--eval-rays", type=int, default=4096)
    a = ap.parse_args()
    ref = find_reference()
    setup_paths(ref)
    import torch
    assert torch.cuda.is_available(), "needs the GPU"
    import numpy as np
    import permuto_sdf
    import permutohedral_encoding
    assert permuto_sdf.OccupancyGrid.__module__.startswith("permuto_sdf_amd")
    assert permutohedral_encoding.PermutoEncoding.__module__.startswith("permuto_sdf_amd")

    import permuto_sdf_py.train_permuto_sdf as T        # module level: default tensor type -> cuda, TrainParams.create
    from permuto_sdf_py.callbacks.callback import Callback
    log = {"reference": ref, "torch": torch.__version__, "device": torch.cuda.get_device_name(0), "events": []}

    # ---- the reference's own schedule knobs (attributes of ITS hyperparams object)
    hp = T.hyperparams
    hp.nr_iter_sphere_fit = a.sphere_iters
    hp.iter_finish_training = a.sphere_iters + a.train_iters
    log["hyperparams"] = {k: getattr(hp, k) for k in ("nr_iter_sphere_fit", "iter_finish_training", "nr_rays",
                                                      "target_nr_of_samples", "max_nr_samples_per_ray")}

    # ---- logging callback, added to the reference's own callback group
    losses, stamps = [], []

    class LossLog(Callback):
        def after_forward_pass(self, phase=None, loss=None, loss_rgb=None, loss_eikonal=None, **kw):
            it = phase.iter_nr
            if it % a.log_every == 0 or it in (a.sphere_iters, a.sphere_iters + 1):
                torch.cuda.synchronize()
                stamps.append((it, time.time()))
                rec = {"iter": it, "loss": float(loss), "loss_rgb": float(loss_rgb), "loss_eikonal": float(loss_eikonal),
                       "phase": "sphere_init" if it < a.sphere_iters else "train"}
                losses.append(rec)
                print("[reference train] iter %5d %-11s loss %.5f rgb %.5f eik %.5f" %
                      (it, rec["phase"], rec["loss"], rec["loss_rgb"], rec["loss_eikonal"]), flush=True)

    orig_create = T.create_callbacks

    def create_callbacks(*args, **kw):
        cb = orig_create(*args, **kw)
        cb.callbacks.insert(0, LossLog())
        return cb
    T.create_callbacks = create_callbacks

    # ---- keep handles on what train() builds
    made = {}

    def recording(name, ctor):
        def make(*args, **kw):
            made[name] = ctor(*args, **kw)
            return made[name]
        return make
    T.SDF, T.RGB, T.NerfHash = recording("sdf", T.SDF), recording("rgb", T.RGB), recording("bg", T.NerfHash)
    T.Colorcal, T.OccupancyGrid = recording("colorcal", T.Colorcal), recording("grid", T.OccupancyGrid)
    orig_f2t = T.MiscDataFuncs.frames2tensors

    def frames2tensors(frames):
        made["frames"] = frames
        made["reel"] = orig_f2t(frames)
        return made["reel"]
    T.MiscDataFuncs.frames2tensors = staticmethod(frames2tensors)

    sys.argv = ["train_permuto_sdf.py", "--dataset", "dtu", "--scene", "dtu_scan24", "--comp_name", "comp_1", "--no_viewer"]
    t0 = time.time()
    prof_file = os.environ.get("PSDF_PROFILE_REFERENCE")   # cProfile of the whole run() -> top functions into this file
    if prof_file:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
    T.run()
    torch.cuda.synchronize()
    if prof_file:
        pr.disable()
        with open(prof_file, "w") as fh:
            st = pstats.Stats(pr, stream=fh)
            st.sort_stats("tottime").print_stats(60)
            st.sort_stats("cumulative").print_stats(90)
    log["train_wall_s"] = time.time() - t0
    log["losses"] = losses

    def rate(lo, hi):
        s = [(i, t) for i, t in stamps if lo <= i <= hi]
        return (s[-1][0] - s[0][0]) / (s[-1][1] - s[0][1]) if len(s) >= 2 and s[-1][1] > s[0][1] else None
    log["iters_per_s"] = {"sphere_init": rate(0, a.sphere_iters - 1), "train": rate(a.sphere_iters + 1, 10 ** 9)}
    tr = [r for r in losses if r["phase"] == "train"]
    log["train_loss_first_last"] = [tr[0]["loss"], tr[-1]["loss"]] if tr else None
    log["train_rgb_loss_first_last"] = [tr[0]["loss_rgb"], tr[-1]["loss_rgb"]] if tr else None
    sp = [r for r in losses if r["phase"] == "sphere_init"]
    log["sphere_loss_first_last"] = [sp[0]["loss"], sp[-1]["loss"]] if sp else None
    print("[reference train] %.1f s; it/s %s" % (log["train_wall_s"], log["iters_per_s"]), flush=True)

    model_sdf, model_rgb, model_bg, grid, reel = made["sdf"], made["rgb"], made["bg"], made["grid"], made["reel"]
    for m in (model_sdf, model_rgb, model_bg):
        m.eval()
    it_eval = a.train_iters
    cos_anneal = T.map_range_val(it_eval, 0.0, hp.forced_variance_finish_iter, 0.0, 1.0)
    forced_var = T.map_range_val(it_eval, 0.0, hp.forced_variance_finish_iter, 0.3, hp.forced_variance_finish)

    class Args:
        with_mask = False
        dataset = "dtu"

    # ---- run_net (eval mode: no jitter) on fixed rays, against this repository's own trainer with the same weights
    torch.manual_seed(123)
    with torch.no_grad():
        o, d, gt, _, img_idx = T.PermutoSDF.random_rays_from_reel(reel, a.eval_rays)
    pred_rgb, pred_bg, pred_normals, sdf_grad, wsum, fg = T.run_net(Args, hp, o, d, img_idx, model_sdf, model_rgb, model_bg,
                                                                    None, grid, it_eval, cos_anneal, forced_var)
    log["run_net"] = {"rays": int(o.shape[0]), "fg_samples": int(fg.samples_pos.shape[0]),
                      "pred_rgb_mean": float(pred_rgb.mean()), "l1_vs_gt": float((pred_rgb - gt).abs().mean())}
    from permuto_sdf_amd.train_step import Trainer
    ck = tempfile.mkdtemp(prefix="psdf_ck_")
    mp = model_sdf.save(ck, "cmp", 0)
    model_rgb.save(ck, "cmp", 0)
    model_bg.save(ck, "cmp", 0, additional_name="_bg")
    torch.save(grid.get_grid_values(), os.path.join(mp, "grid_values.pt"))
    torch.save(grid.get_grid_occupancy(), os.path.join(mp, "grid_occupancy.pt"))
    trainer = Trainer(o.device, seed=0)
    trainer.load_checkpoint(mp)
    for m in (trainer.sdf, trainer.rgb, trainer.bg):
        m.eval()
    # checkpoint files both ways: the reference's files were just loaded (strict) into this repository's nets;
    # now this repository's files go back into the reference's own model classes, strict, including RGB (LipshitzMLP's
    # duplicated keys, volume_renderer_neus.deviation_network.variance) which cannot be constructed without a GPU
    back = tempfile.mkdtemp(prefix="psdf_ck_back_")
    trainer.save_checkpoint(back)
    report = {}
    for name, model, fname in (("sdf", model_sdf, "sdf_model.pt"), ("rgb", model_rgb, "rgb_model.pt"),
                               ("bg", model_bg, "nerf_hash_model_bg.pt")):
        sd = torch.load(os.path.join(back, fname), map_location=o.device)
        ref_sd = model.state_dict()
        assert set(sd) == set(ref_sd), (name, sorted(set(sd) ^ set(ref_sd)))
        assert all(sd[k].shape == ref_sd[k].shape for k in ref_sd), name
        same = all(torch.equal(sd[k].to(ref_sd[k].device), ref_sd[k]) for k in ref_sd)
        model.load_state_dict(sd, strict=True)
        report[name] = {"keys": len(sd), "identical_after_round_trip": bool(same)}
    log["checkpoint_round_trip_reference_classes"] = report
    print("[checkpoints]", report, flush=True)
    pred2, _, fg2, _ = trainer._render(o, d, it_eval, cos_anneal, forced_var, jitter=False)
    num = (pred2.detach() - pred_rgb.detach()).abs()
    log["run_net_vs_trainer_render"] = {
        "max_abs": float(num.max()), "max_rel_to_max": float(num.max() / pred_rgb.abs().max()),
        "mean_abs": float(num.mean()), "quantile_0.9999_abs": float(torch.quantile(num.view(-1), 0.9999)),
        "values_above_1e-4": int((num > 1e-4).sum()), "values": int(num.numel()), "fg_samples_reference_python": int(fg.samples_pos.shape[0]),
        "fg_samples_trainer": int(fg2.samples_pos.shape[0]),
        "same_sample_count": int(fg.samples_pos.shape[0]) == int(fg2.samples_pos.shape[0])}
    print("[run_net vs Trainer._render]", log["run_net_vs_trainer_render"], flush=True)
    # ---- PSDF_FUSE_REFERENCE_MLPS=1: which sub-modules of the reference's models run on the fused evaluators, and the same
    # run_net with the SAME weights on torch's own Linear / GELU evaluation (the fusion undone), for the 1e-4 bar
    from permuto_sdf_amd import reference_fusion as RF
    log["fused_modules"] = {n: {k: type(c).__module__ + "." + type(c).__name__ for k, c in m.named_children()
                                if k.startswith("mlp")} for n, m in (("sdf", model_sdf), ("rgb", model_rgb), ("bg", model_bg))}
    if RF.enabled():
        undone = [RF.unfuse_model(m) for m in (model_sdf, model_rgb, model_bg)]
        pred_u = T.run_net(Args, hp, o, d, img_idx, model_sdf, model_rgb, model_bg, None, grid, it_eval, cos_anneal, forced_var)[0]
        for m in (model_sdf, model_rgb, model_bg):
            RF.fuse_model(m)
        du = (pred_u.detach() - pred_rgb.detach()).abs()
        log["run_net_fused_vs_unfused"] = {"unfused": undone, "max_abs": float(du.max()), "max_rel_to_max": float(du.max() / pred_u.abs().max()),
                                           "values_above_1e-4": int((du > 1e-4).sum()), "values": int(du.numel())}
        print("[run_net fused vs unfused]", log["run_net_fused_vs_unfused"], flush=True)

    # ---- importance sampling on its own (sdf_utils.py:383-423)
    with torch.no_grad():
        _, te, _, tx, _ = model_sdf.boundary_primitive.ray_intersection(o, d)
        fg0 = grid.compute_samples_in_occupied_regions(o, d, te, tx, hp.min_dist_between_samples, hp.max_nr_samples_per_ray,
                                                       False).compact_to_valid_samples()
        fg1 = T.importance_sampling_sdf_model(model_sdf, fg0, o, d, tx, it_eval)
        se = fg1.ray_start_end_idx
        z = fg1.samples_z.view(-1)
        # sortedness of z inside every ray (size-independent property of the merge)
        idx = permuto_sdf.RaySamplesPacked.compute_per_sample_ray_idx(se, z.shape[0]).long()
        same_ray = idx[1:] == idx[:-1]
        log["importance_sampling"] = {"uniform_samples": int(fg0.samples_pos.shape[0]), "combined_samples": int(z.shape[0]),
                                      "z_sorted_within_rays": bool(((z[1:] >= z[:-1]) | ~same_ray).all())}
    print("[importance sampling]", log["importance_sampling"], flush=True)

    # ---- sphere_trace of the reference (boolean-mask compaction loop) vs the fixed-shape tracer
    from permuto_sdf_amd.sphere_trace import SphereTracer
    frame = made["frames"][0].subsample(2.0, subsample_imgs=False)
    o_f, d_f = model_rgb.create_rays(frame, rand_indices=None)
    model_sdf(o_f[:8] * 0.2, it_eval)      # sets last_iter_nr, which sphere_trace reads (sdf_utils.py:163)
    t0 = time.time()
    pts, sdf_e, grad_e, feat_e, traced = T.sphere_trace(15, o_f, d_f, model_sdf, True, 0.9, 2e-4, occupancy_grid=grid)
    torch.cuda.synchronize()
    t_ref = time.time() - t0
    lin = [m for m in model_sdf.mlp_sdf if isinstance(m, torch.nn.Linear)]
    from permuto_sdf_amd.mlp import FusedMLP
    fm = FusedMLP([lin[0].in_features] + [l.out_features for l in lin]).to(o_f.device)
    with torch.no_grad():
        for dst, src in zip(fm.layers, lin):
            dst.weight.copy_(src.weight)
            dst.bias.copy_(src.bias)
    window = model_sdf.c2f(T.map_range_val(it_eval, 0.0, model_sdf.nr_iters_for_c2f, 0.3, 1.0)).view(-1).contiguous()
    tracer = SphereTracer(model_sdf.encoding, fm, grid, model_sdf.boundary_primitive, window=window)
    pts2, sdf2, grad2, conv2 = tracer.trace(o_f, d_f, 15, 0.9, 2e-4, True)
    # the reference returns only rays that hit an occupied region (compacted, ray order kept)
    hit = ~(pts2 == o_f).all(1)
    dp = (pts2[hit] - pts).abs().max() if int(hit.sum()) == pts.shape[0] else torch.tensor(float("nan"))
    log["sphere_trace"] = {"rays": int(o_f.shape[0]), "rays_traced_reference": int(pts.shape[0]),
                           "rays_traced_fixed_shape": int(hit.sum()), "end_point_max_abs_diff": float(dp),
                           "reference_python_s": t_ref, "converged_frac": float((sdf_e.abs() < 2e-4).float().mean())}
    print("[sphere_trace]", log["sphere_trace"], flush=True)
    # without occupancy grid: RaySamplesPacked.initialize_with_one_sample_per_ray (src/RaySamplesPacked.cu:97-122)
    pts_n, sdf_n, grad_n, _, traced_n = T.sphere_trace(15, o_f, d_f, model_sdf, True, 0.9, 2e-4, occupancy_grid=None)
    log["sphere_trace_no_grid"] = {"rays": int(pts_n.shape[0]), "start_end_dtype": str(traced_n.ray_start_end_idx.dtype),
                                   "converged_frac": float((sdf_n.abs() < 2e-4).float().mean())}
    print("[sphere_trace, no grid]", log["sphere_trace_no_grid"], flush=True)

    # ---- the reference's two full-image renderers on the subsampled frame
    t0 = time.time()
    rgb_img, _, nrm_img, w_img = T.run_net_sphere_traced(frame, Args, hp, model_sdf, model_rgb, model_bg, grid, it_eval,
                                                         cos_anneal, forced_var, 15, 0.9, 2e-4)
    torch.cuda.synchronize()
    log["run_net_sphere_traced"] = {"shape": list(rgb_img.shape), "s": time.time() - t0, "coverage": float(w_img.mean())}
    t0 = time.time()
    rgb_img2, bg_img2, nrm2, w2 = T.run_net_in_chunks(frame, 4096, Args, hp, model_sdf, model_rgb, model_bg, grid, it_eval,
                                                      cos_anneal, forced_var)
    torch.cuda.synchronize()
    gt_img = torch.as_tensor(np.ascontiguousarray(made["frames"][0].rgb_32f)).permute(2, 0, 1)[None].to(rgb_img2.device)
    gt_small = torch.nn.functional.interpolate(gt_img, size=rgb_img2.shape[-2:], mode="area")
    mse = float(((rgb_img2 - gt_small) ** 2).mean())
    log["run_net_in_chunks"] = {"shape": list(rgb_img2.shape), "s": time.time() - t0, "mse_vs_gt": mse,
                                "psnr_vs_gt": float(-10 * np.log10(max(mse, 1e-12)))}
    print("[run_net_in_chunks]", log["run_net_in_chunks"], flush=True)

    # ---- the reference's mesh export (sdf_utils.py:252-292) -> marching_cubes stand-in -> Mesh.save_to_file (.ply)
    from permuto_sdf_py.utils.sdf_utils import extract_mesh_from_sdf_model
    t0 = time.time()
    mesh = extract_mesh_from_sdf_model(model_sdf, nr_points_per_dim=192, min_val=-0.5, max_val=0.5)
    ply = os.path.splitext(a.out)[0] + "_mesh.ply"
    mesh.save_to_file(ply)
    V = np.asarray(mesh.V)
    rad = np.linalg.norm(V, axis=1)
    log["mesh_export"] = {"vertices": int(len(V)), "faces": int(len(mesh.F)), "s": time.time() - t0, "file": os.path.basename(ply),
                          "bytes": os.path.getsize(ply), "radius_median": float(np.median(rad)),
                          "note": "the synthetic scene is a radius-0.3 sphere"}
    print("[mesh export]", log["mesh_export"], flush=True)

    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(log, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
