"""BASELINE config 4: training iterations per second of the full SDF + colour + background step
(permuto_sdf_amd/train_step.py) on a synthetic image reel; one process per GPU (launch with torch.distributed.run
for N > 1, same contract as bench.py).  Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permuto_sdf_amd import parallel  # noqa: E402
from permuto_sdf_amd.train_step import SyntheticReel, Trainer  # noqa: E402


def measure(dev, manual=True, steps=60, warmup=20, repeats=3, start_iter=0):
    """it/s of the cfg-4 step on `dev` for THIS process group (parallel.init() already called): every rank runs it, the
    returned dict is complete on rank 0.  The timed block is repeated and the median repetition reported (the step is host
    bound: the rate moves by +-10 % with the state of the box's CPU); max over ranks per repetition."""
    rank, world = parallel.rank(), parallel.world_size()
    if manual:
        from permuto_sdf_amd.train_manual import ManualTrainer
        tr = ManualTrainer(dev)
    else:
        tr = Trainer(dev)
    reel = SyntheticReel(dev)
    tr.iter = start_iter
    for _ in range(warmup):
        tr.step(reel)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    reps = []
    samples = 0
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        samples = 0
        for _ in range(steps):
            tr.step(reel)
            samples += tr.last["nr_fg_samples"]
        barrier()
        e = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            torch.distributed.all_reduce(e, op=torch.distributed.ReduceOp.MAX)
        reps.append(float(e.item()))
    el = sorted(reps)[len(reps) // 2]
    tr.sync_parameters()     # (data parallel: the last step's parameter all-gather may still be in flight)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(p).all()) for p in tr.params), "a parameter is not finite after the run"
    dp = None
    if parallel.collectives_active() and getattr(tr, "last_dp", None):
        # what a step exchanges per GPU, and the analytic budget of an 8-GPU node (no such node has run this code: a model, not a
        # measurement).  xGMI: 7 point-to-point links per GPU, ~76.8 GB/s per link and direction (153.6 GB/s bidirectional);
        # reduce-scatter and all-gather of B bytes each move (w-1)/w * B per GPU and direction over all links at once.
        ld = tr.last_dp
        rs_b = sum(ld.get("sharded_bytes", [])) + sum(ld.get("bucket_bytes", []))
        ag_b = sum(ld.get("gather_bytes", []))
        w8 = 8
        peak = 7 * 76.8e9
        eff = 0.6 * peak                       # what RCCL typically sustains of the link peak at these message sizes (assumption)
        t_rs = (w8 - 1) / w8 * rs_b / eff * 1e3
        t_ag = (w8 - 1) / w8 * ag_b / eff * 1e3
        per_lat_rs, per_lat_ag = t_rs / 3.0, t_ag / 3.0
        # hidden by the schedule: the background and colour lattices' reduce-scatter (their gradients are final ~1.0 / ~0.5 ms before
        # the end of the backward), the colour lattice's all-gather (read ~0.3 ms into the next step) and ~0.15 ms of the others
        # behind the next step's ray draw + march
        exposed = max(0.0, per_lat_rs) + max(0.0, 2 * per_lat_ag - 0.15)
        dp = {"comm_bytes_per_step_per_gpu": {"reduce_scatter_gradients": rs_b, "all_gather_parameters": ag_b},
              "schedule": "reduce-scatter of a lattice's gradient starts when its last backward kernel is enqueued (background, colour, "
                          "then SDF at the end of the backward); sharded AdamW on the owned 1/world; all-gather of the PARAMETERS in "
                          "the order the next step reads them (background, SDF, colour), each waited for where it is first read",
              "deferred_gather": ld.get("deferred_gather"), "optimizer": ld.get("optimizer"),
              "analytic_8gpu_budget_ms": {"model": "7 xGMI links x 76.8 GB/s per direction, 60 % sustained (assumption, not measured)",
                                          "reduce_scatter_total": round(t_rs, 3), "all_gather_total": round(t_ag, 3),
                                          "exposed_after_overlap": round(exposed, 3),
                                          "budget_for_6x_of_8": round(el / steps * 1e3 * (8 / 6 - 1), 3)}}
    if world > 1:   # replicas must have stayed identical: same parameters on every rank after the run
        chk = torch.stack([p.detach().double().sum() for p in tr.params]).to(dev)
        lo, hi = chk.clone(), chk.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        assert torch.equal(lo, hi), "replicas diverged"
    return {"metric": "train iters/sec (cfg 4: SDF + colour + background step, synthetic reel)",
            "value": steps / el, "unit": "it/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": el / steps * 1e3,
            "fg_samples_per_step_per_gpu": samples / steps, "rays_last_step": tr.last["nr_rays"],
            "scaling": "weak", "dtype": "f32", "data": "synthetic", "start_iter": start_iter,
            "backward": "hand-written (train_manual.py)" if manual else "torch autograd over the fused operators",
            "repeats_it_per_s": [round(steps / r, 1) for r in reps], "dp": dp}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--start-iter", type=int, default=0, help="iteration counter the run starts at: 0 = the coarse-to-fine window of "
                    "the SDF lattice is still mostly closed (fine levels carry no gradient); 20000 = every level open")
    ap.add_argument("--manual", action="store_true", help="train_manual.ManualTrainer: hand-written backward over the raw kernels")
    args = ap.parse_args()
    rank, world, local = parallel.init()
    if os.environ.get("PSDF_BENCH_SINGLE_DEVICE") == "1":
        local = 0   # development aid (with PSDF_DIST_BACKEND=gloo): every rank on cuda:0, exercises the N>1 path
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    res = measure(dev, args.manual, args.steps, args.warmup, args.repeats, args.start_iter)
    if rank == 0:
        print(json.dumps(res))
    parallel.shutdown()


if __name__ == "__main__":
    main()
