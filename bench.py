#!/usr/bin/env python
"""Benchmark of the PermutoSDF data-parallel hot path on MI355X (contract: see the round instructions).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1] + the compositing of configs[2]'s chunk size): per GPU 16 384 rays x 128 samples
= 2 097 152 synthetic ray samples inside the radius-0.5 bounding sphere, already resident in HBM.  One step =
  forward : 16-level permutohedral encode (T=2^18, F=2, +points) -> fused 64x3 SDF MLP -> NeuS section-point opacity
            (volume_rendering_modules.py:129-172) -> transmittance cumprod -> weights -> radiance integration -> L1 loss
  backward: a TRUE gradient of that loss: L1 -> integrate_backward -> cumprod backward (incl. per-ray inverse cumsum) ->
            opacity backward (dL/dsdf) -> fused MLP backward (dX, dW, db) -> encode backward (lattice gradient)
  N > 1   : RCCL sum all-reduce of MLP + lattice gradients (bucketed, overlapped with the encode backward)
  update  : fused AdamW on lattice + MLP parameters
value = ray samples through that whole step per second, summed over ranks (weak scaling: per-GPU work is fixed).
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL / tensor sharing fail with hipIpcGetMemHandle otherwise); the
# launcher's environment normally carries it -- set here, before the HIP runtime starts, in case it does not
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

NR_RAYS = 16384
SAMPLES_PER_RAY = 128
NR_LEVELS = 16
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_MFMA_PEAK_TF = 157.3


def make_batch(dev, seed, nr_rays=NR_RAYS, per_ray=SAMPLES_PER_RAY):
    """Equal-count packed ray samples inside the bounding sphere + synthetic per-sample radiance."""
    from permuto_sdf import RaySamplesPacked, Sphere
    g = torch.Generator(device="cpu").manual_seed(seed)
    o = torch.randn(nr_rays, 3, generator=g)
    o = o / o.norm(dim=1, keepdim=True) * 1.5
    target = (torch.rand(nr_rays, 3, generator=g) - 0.5) * 0.7
    d = target - o
    d = d / d.norm(dim=1, keepdim=True)
    o, d = o.to(dev), d.to(dev)
    _, te, _, tx, hit = Sphere(0.5, [0, 0, 0]).ray_intersection(o, d)
    n = per_ray
    frac = (torch.arange(n, device=dev, dtype=torch.float32) + 0.5) / n
    z = te + (tx - te) * frac[None, :]                                   # [R, n]
    rs = RaySamplesPacked(nr_rays, nr_rays * n, device=dev)
    rs.rays_have_equal_nr_of_samples, rs.fixed_nr_of_samples_per_ray = True, n
    rs.samples_z = z.reshape(-1, 1).contiguous()
    rs.samples_dt = ((tx - te) / n).expand(-1, n).reshape(-1, 1).contiguous()
    rs.samples_pos = (o[:, None, :] + z[:, :, None] * d[:, None, :]).reshape(-1, 3).contiguous()
    rs.samples_dirs = d[:, None, :].expand(-1, n, -1).reshape(-1, 3).contiguous()
    rs.ray_fixed_dt = ((tx - te) / n).contiguous()
    idx = torch.arange(nr_rays, device=dev, dtype=torch.int32) * n
    rs.ray_start_end_idx = torch.stack([idx, idx + n], 1).contiguous()
    rs.cur_nr_samples.fill_(nr_rays * n)
    rs._exact = True
    rgb = torch.rand(nr_rays * n, 3, generator=g).to(dev)
    # direction of the SDF gradient at the samples (an input of the first-order path): the analytic normal of a centred
    # sphere; and the ground-truth radiance of every ray for the L1 loss
    normals = torch.nn.functional.normalize(rs.samples_pos, dim=1).contiguous()
    gt = torch.rand(nr_rays, 3, generator=g).to(dev)
    return rs, rgb, (o, d, te, tx, normals, gt)


def cpu_baseline(cores):
    """The CPU oracle timed on a bounded sample of the same step: torch-vectorised encode restatement + the unmodified
    torch.nn MLP + the torch restatement of the NeuS opacity / compositing / L1 loss (oracle/neus_oracle.py), forward and
    autograd backward, fp32, `cores` threads."""
    from oracle import neus_oracle as no
    from oracle import permuto_oracle as po
    torch.set_num_threads(cores)
    rays, per_ray = 256, SAMPLES_PER_RAY
    N = rays * per_ray
    g = torch.Generator().manual_seed(1)
    pos = torch.randn(N, 3, generator=g)
    pos = pos / pos.norm(dim=1, keepdim=True) * 0.5 * torch.rand(N, 1, generator=g) ** (1 / 3)
    lat, shifts = po.make_params(3, 2 ** 18, NR_LEVELS, 2, seed=2, init_scale=1e-2)
    lat.requires_grad_(True)
    sl = np.geomspace(1.0, 1e-4, NR_LEVELS)
    C = po.output_dims(3, NR_LEVELS, 2, True)
    mlp = torch.nn.Sequential(torch.nn.Linear(C, 64), torch.nn.GELU(), torch.nn.Linear(64, 64), torch.nn.GELU(),
                              torch.nn.Linear(64, 64), torch.nn.GELU(), torch.nn.Linear(64, 1))
    dirs = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=1)
    normals = torch.nn.functional.normalize(pos, dim=1)
    dt = torch.full((N, 1), 1.0 / per_ray / 2)
    rgb = torch.rand(N, 3, generator=g)
    gt = torch.rand(rays, 3, generator=g)
    inv_s = torch.exp(torch.tensor(5.0))
    win = torch.ones(NR_LEVELS)

    def one():
        feat = po.encode(pos, lat, sl, shifts, win, True, 1e-3)
        sdf = mlp(feat)
        alpha, om = no.neus_alpha(sdf, dirs, normals, dt, inv_s, 1.0)
        pred, _, _ = no.composite_equal(alpha, om, rgb, rays, per_ray)
        loss = no.rgb_loss(gt, pred, torch.ones(rays, 1))
        loss.backward()
        lat.grad = None
        mlp.zero_grad()

    one()  # warm-up (page-in, thread pools)
    t0 = time.perf_counter()
    reps = 0
    while reps < 3 or (time.perf_counter() - t0) < 10.0:
        one()
        reps += 1
        if time.perf_counter() - t0 > 30.0:
            break
    dt_s = (time.perf_counter() - t0) / reps
    return {"value": N / dt_s, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "%d rays x %d samples (%d samples) of the same step: torch-CPU encode restatement, torch.nn 64x3 MLP, "
                      "torch restatement of NeuS opacity + compositing + L1 loss, forward + autograd backward; %d repetitions, "
                      "%.2f s each; %d threads of the host's %d cores (the vectorised restatement gets slower beyond ~8 threads)"
                      % (rays, per_ray, N, reps, dt_s, cores, os.cpu_count() or 0)}


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--cpu-baseline-child":
        print(json.dumps(cpu_baseline(int(sys.argv[2]))), flush=True)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 200 steps of ~3 ms keep the GPU busy for ~0.6 s, long enough for an outside sampler (rocm-smi) to see the load
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra 24-level row (profiling runs: one kernel variant per name)")
    args = ap.parse_args()

    from permuto_sdf_amd import parallel
    from permuto_sdf_amd.hotpath import SdfHotPath
    rank, world, local = parallel.init()
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if os.environ.get("PSDF_BENCH_SINGLE_DEVICE") == "1":
        local = 0  # development aid: run every rank on cuda:0 (with PSDF_DIST_BACKEND=gloo) to exercise the N>1 path
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    hp = SdfHotPath(nr_levels=NR_LEVELS, hidden=64, out_channels=1, device=dev, seed=0)   # replicated parameters
    rs, rgb, aux = make_batch(dev, parallel.rank_seed(7, rank))                             # per-rank rays
    normals, gt = aux[4], aux[5]
    N = rs.samples_pos.shape[0]
    from permuto_sdf_amd.neus import l1_loss_raw

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    import ctypes
    from permuto_sdf_amd import _lib as _L
    _lp = _L.lib().psdf_last_path
    _lp.restype = ctypes.c_int
    K = args.steps

    def measure(h, warmup):
        """W untimed warm-up steps, then EXACTLY K timed steps between two barriers; per-kernel-family HIP events on the launch
        stream (torch's current stream == the stream the kernels are launched on).  -> (elapsed seconds of this rank, events,
        (mlp backward path, mlp forward path) the timed steps dispatched to)"""
        for _ in range(warmup):
            h.step(rs, rgb, normals, gt)
        ev_ = {k: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
               for k in ("enc_bwd", "mlp_bwd", "fwd", "mlp_fwd", "comm_wait")}
        barrier()
        t0_ = time.perf_counter()
        for i in range(K):
            h.events = {"enc_bwd": ev_["enc_bwd"][i], "mlp_bwd": ev_["mlp_bwd"][i], "comm_wait": ev_["comm_wait"][i],
                        "mlp_fwd": ev_["mlp_fwd"][i]}
            ev_["fwd"][i][0].record()
            pred, saved = h.forward(rs, rgb, normals)
            ev_["fwd"][i][1].record()
            loss, g_pred = l1_loss_raw(pred, gt)
            h.backward(rs, rgb, saved, g_pred)
        barrier()
        el = time.perf_counter() - t0_
        h.events = None
        # which kernels the timed steps dispatched to (debug query of the library; read NOW, later launches change it)
        return el, ev_, (int(_lp(ctypes.c_int(1))), int(_lp(ctypes.c_int(2))))

    elapsed, ev, (path_bwd, path_fwd) = measure(hp, args.warmup)
    extra = {}
    cdev = dev if (world == 1 or torch.distributed.get_backend() == "nccl") else "cpu"
    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    per_rank = None
    if world > 1:
        # self-diagnosing N > 1 line: every rank's own wall time per step and the time its step waited for communication
        # (HIP events around GradientBuckets.finish()), gathered on rank 0; the headline uses the MAX over ranks
        comm_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["comm_wait"]]))
        mine = torch.tensor([elapsed / K * 1e3, comm_ms], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        per_rank = {"ms_per_step": [float(x[0]) for x in allr], "comm_wait_ms": [float(x[1]) for x in allr]}
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())
    # self-certification of an N > 1 run (gathered on every rank, printed by rank 0): which physical device each rank ran on
    # (uuid / PCI bus id: N distinct ones, or the run says otherwise), the collective library's version, and whether the
    # replicas still hold bit-identical parameters after the timed steps (a checksum per rank, MIN == MAX over ranks)
    cert = None
    if world > 1:
        try:
            pr = torch.cuda.get_device_properties(dev)
            ident = "%s|%s|%s" % (getattr(pr, "uuid", "?"), getattr(pr, "pci_bus_id", "?"), getattr(pr, "name", "?"))
            try:
                ident += "|bus=%s" % torch.cuda.get_device_properties(dev).pci_bus_id
            except Exception:
                pass
            idents = [None] * world
            torch.distributed.all_gather_object(idents, ident)
            ck = torch.zeros(2, dtype=torch.float64, device=cdev)
            with torch.no_grad():
                ps = [hp.enc.lattice_values] + [l.weight for l in hp.mlp.layers] + [l.bias for l in hp.mlp.layers]
                ck[0] = sum(float(p.detach().double().sum()) for p in ps)
                ck[1] = sum(float((p.detach().double() * p.detach().double()).sum()) for p in ps)
            lo, hi = ck.clone(), ck.clone()
            torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
            torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
            single = os.environ.get("PSDF_BENCH_SINGLE_DEVICE") == "1"
            cert = {"devices": idents, "distinct_devices": len(set(idents)),
                    "single_device_development_run": single,
                    "replicas_bit_identical": bool((lo == hi).all()),
                    "parameter_checksum": [float(lo[0]), float(lo[1])],
                    "backend": torch.distributed.get_backend(),
                    "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version())
                                     if torch.distributed.get_backend() == "nccl" else None),
                    "torch": torch.__version__, "hip": getattr(torch.version, "hip", None)}
            if not single and torch.distributed.get_backend() == "nccl" and len(set(idents)) != world:
                cert["error"] = "ranks share a device: %d distinct devices for %d ranks" % (len(set(idents)), world)
        except Exception as e:
            cert = {"error": repr(e)}

    P, F, L_, Tcap = 3, 2, NR_LEVELS, 2 ** 18
    C_in = F * (L_ + 2)

    def rooflines(ms, path_bwd, path_fwd):
        """-> (roofline of the dominant kernel family, the others, arithmetic description) from the mean event brackets `ms`"""
        # Roofline of the DOMINANT kernel of the step (longest mean launch time, HIP events on the launch stream).
        # encode backward (HBM): algorithmic bytes/sample 4*(P + L*F + 2*L*F*(P+1)) = position + L*F
        #   upstream gradients + read-modify-write of the (P+1)*L*F table entries, plus zero-fill + final read of
        #   the 4*L*T*F gradient table per launch.
        # MLP backward (fp32 MFMA): algorithmic FLOP/sample = 2 x forward = 2 * 2*(C*64 + 64*64*2 + 64) (dX chain + dW);
        #   the in-kernel recomputation of the forward is NOT counted.
        enc_bytes = N * 4 * (P + L_ * F + 2 * L_ * F * (P + 1)) + 2 * 4 * L_ * Tcap * F
        mlp_flops = 2 * 2 * (C_in * 64 + 64 * 64 * 2 + 64) * N
        f16 = path_bwd == 4                        # which MLP backward kernel the timed steps dispatched to
        fwd_f16 = path_fwd == 3                    # and which forward kernel
        mlp_kernel = ("mlp_bwd_split_f16_kernel (+ mlp_split_pack_kernel, mlp_absmax_kernel, mlp_split_reduce_kernel)" if f16 else
                      "mlp_bwd_split_kernel (+ mlp_split_pack_kernel, mlp_split_reduce_kernel)")
        mlp_note = ("fp32-equivalent arithmetic priced against the fp32 matrix peak (157.3 TF): every fp32 operand is two fp16 pieces "
                    "(11 + 11 mantissa bits) and every product three (chains) or four (dW) fp16 MFMA products on "
                    "v_mfma_f32_16x16x32_f16 -- fp32 MFMAs do not overlap with VALU work on gfx950, and gfx950's matrix pipe honours "
                    "fp16 subnormals (attic/prototypes/mlp_fwd_split_f16.hip); the gradient chain of each sample runs on the mantissa of its dY "
                    "(exact rescaling), so accuracy does not depend on the size or spread of dY; the forward recomputation and the "
                    "operand transposes on the matrix pipe are extra, uncounted work; the event bracket also holds the three small "
                    "pack / absmax / reduce launches" if f16 else
                    "fp32-equivalent arithmetic priced against the fp32 matrix peak (157.3 TF): every fp32 product "
                    "is evaluated as six bf16 MFMA products (v_mfma_f32_16x16x32_bf16, three bf16 pieces per operand) "
                    "because fp32 MFMAs do not overlap with VALU work on gfx950; the forward recomputation and the "
                    "operand transposes on the matrix pipe are extra, uncounted work (a dW MFMA carries two piece "
                    "products in its two K halves); the event bracket also holds the two small pack / reduce launches")
        cand = {
            "enc_bwd": {"bound": "hbm", "kernel": "encode_bwd_kernel<3,2,true,false,true> + encode_bwd_reduce_kernel<2>",
                        "achieved": enc_bytes / (ms["enc_bwd"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "avg_launch_ms": ms["enc_bwd"], "algorithmic_bytes_per_launch": enc_bytes,
                        "note": "scatter-add: fp32 global atomics cap at ~21 G/s and ds_add_f32 at ~200 G/s on this chip "
                                "(tools/atomic_bench.hip), so runs are summed in registers, pairs are added in LDS with 64-bit "
                                "CAS and the rest is binned to per-partition queues; HBM is not the limiter"},
            "mlp_bwd": {"bound": "mfma", "kernel": mlp_kernel,
                        "achieved": mlp_flops / (ms["mlp_bwd"] * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                        "avg_launch_ms": ms["mlp_bwd"], "algorithmic_flops_per_launch": mlp_flops, "note": mlp_note},
        }
        if f16:
            # what the matrix pipe really executes: 274 v_mfma_f32_16x16x32_f16 per 16-sample tile (forward recomputation, operand
            # transposes, two-piece products), 16*16*32*2 FLOP each, priced against the dense fp16 peak (MI355X_MICROARCH.md)
            ex = 274 * (N // 16) * 16 * 16 * 32 * 2
            cand["mlp_bwd"]["executed_on_fp16_pipe"] = {"flops_per_launch": ex, "achieved": ex / (ms["mlp_bwd"] * 1e-3) / 1e12,
                                                        "peak": 2500.0, "unit": "TFLOP/s",
                                                        "frac": ex / (ms["mlp_bwd"] * 1e-3) / 1e12 / 2500.0}
        else:
            # 480 v_mfma_f32_16x16x32_bf16 per 16-sample tile (csrc/mlp_bwd_split.hip), priced against the dense bf16 peak
            ex = 480 * (N // 16) * 16 * 16 * 32 * 2
            cand["mlp_bwd"]["executed_on_bf16_pipe"] = {"flops_per_launch": ex, "achieved": ex / (ms["mlp_bwd"] * 1e-3) / 1e12,
                                                        "peak": 2500.0, "unit": "TFLOP/s",
                                                        "frac": ex / (ms["mlp_bwd"] * 1e-3) / 1e12 / 2500.0}
        dom = max(cand, key=lambda k: ms[k])
        roof = cand[dom]
        roof["frac"] = roof["achieved"] / roof["peak"]
        # HBM bytes per launch from the PMC counters: rocprofv3 cannot be driven from inside the timed process, so the
        # number is the committed summary of a separate-pass --pmc run of this same command (profiles/), corrected
        # as MI355X_MICROARCH.md prescribes; null when no summary for the dominant kernel is on file.
        roof["traffic"] = None
        try:
            prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            # the newest committed summary that holds the kernel the timed steps dispatched to
            want = {"enc_bwd": "encode_bwd_kernel", "mlp_bwd": "mlp_bwd_split_f16_kernel" if f16 else "mlp_bwd_split_kernel"}[dom]
            cands = sorted((f for f in os.listdir(prof) if f.endswith("pmc_hbm_traffic.json")), reverse=True)
            src = next((f for f in cands if want in json.load(open(os.path.join(prof, f))).get("kernels", {})), cands[0])
            pmc = json.load(open(os.path.join(prof, src)))
            # the counters are quoted only for the kernels they were taken from: the summary records a hash of csrc/ (tools/
            # pmc_hbm_traffic.sh); a kernel source changed since then -> null + the reason, never yesterday's figure
            import hashlib
            csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "permuto_sdf_amd", "csrc")
            hh = hashlib.sha256()
            for f in sorted(os.listdir(csrc)):
                if f.endswith((".hip", ".h")):
                    hh.update(f.encode())
                    hh.update(open(os.path.join(csrc, f), "rb").read())
            if pmc.get("csrc_sha256") != hh.hexdigest():
                roof["traffic_stale"] = ("profiles/%s was collected from other kernel sources (csrc hash %s..., now %s...): re-run "
                                         "tools/pmc_hbm_traffic.sh" % (src, str(pmc.get("csrc_sha256"))[:12], hh.hexdigest()[:12]))
                raise LookupError("stale")
            key = {"enc_bwd": ["encode_bwd_kernel", "encode_bwd_reduce_kernel"],
                   "mlp_bwd": [next(k for k in ("mlp_bwd_split_f16_kernel" if f16 else "mlp_bwd_split_kernel", "mlp_bwd_split_kernel",
                                                "mlp_bwd_kernel") if k in pmc["kernels"])]}[dom]
            roof["traffic"] = int(sum(pmc["kernels"][k]["hbm_bytes"] for k in key))
            roof["traffic_source"] = "profiles/%s (2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes; kernel(s): %s)" % (src, ", ".join(key))
        except Exception:
            pass
        other = {k: {kk: v[kk] for kk in ("bound", "achieved", "peak", "unit", "avg_launch_ms")} for k, v in cand.items() if k != dom}
        return roof, other, f16, fwd_f16

    if rank == 0:
        ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()}
        roof, other, f16, fwd_f16 = rooflines(ms, path_bwd, path_fwd)
        out = {
            "metric": "ray-samples/sec (encode+MLP+composite)",
            "value": world * N * K / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32 (MLP products on 2 fp16 operand pieces per fp32 value, fp32 accumulation; the 3-piece / 24-bit figure is "
                      "value_fp32_equivalent_24bit)" if (f16 or fwd_f16) else "f32"),
            "data": "synthetic",
            "config": {"workload": "cfg2+composite: 16-level permutohedral encode fwd/bwd + 64x3 SDF MLP fwd/bwd + NeuS "
                                   "compositing fwd/bwd (true gradient of an L1 radiance loss) + AdamW, %d rays x %d samples = %d "
                                   "samples per GPU" % (NR_RAYS, SAMPLES_PER_RAY, N),
                       "pos_dim": 3, "nr_levels": NR_LEVELS, "capacity": Tcap, "feat_per_level": F, "mlp": "36-64-64-64-1 GELU",
                       "mlp_arithmetic": (("forward: fp32 operands as 2 fp16 pieces each (11 + 11 mantissa bits), 3 products kept, fp32 "
                                           "accumulation (max error <= 4e-6 of the largest output against float64 -- measured 2.8e-6 --, "
                                           "tests/test_gpu_mlp.py::test_split_f16_forward_against_float64; inputs beyond +-65504 saturate: "
                                           "PSDF_MLP_FWD_SPLIT=bf16 selects the 3-piece kernel); " if fwd_f16 else
                                           "forward: fp32 operands as 3 bf16 pieces each, 6 products kept (max error 1.3e-6 of the largest "
                                           "output against float64, tests/test_gpu_mlp.py::test_split_bf16_forward_keeps_fp32_accuracy); "
                                           if path_fwd == 2 else "forward: fp32 MFMA; ")
                                          + ("backward: fp32 operands as 2 fp16 pieces each, 3-4 products kept, per-sample exact rescaling "
                                             "of dY (max error of every gradient <= 2e-5 of its largest entry against float64 -- measured "
                                             "5e-7 .. 6e-6 over 32 shapes and dY distributions --, north_star tolerance 1e-4: "
                                             "tests/test_gpu_mlp.py::test_split_f16_backward_matches_float64; PSDF_MLP_BWD_SPLIT=bf16 "
                                             "selects the 3-piece kernel, 1e-6, 1.5x slower)" if f16 else
                                             "backward: 3 bf16 pieces, 6 products (fp32 rounding level, ::test_split_bf16_backward_matches_float64)")
                                          + "; fp32 accumulation everywhere"),
                       "arithmetic_bits": {"mlp_forward": 23 if fwd_f16 else 24, "mlp_backward": 23 if f16 else 24,
                                           "note": "operand mantissa bits the MLP products keep: two fp16 pieces with the high one rounded to "
                                                   "nearest = 11 + sign + 11 (an fp32 operand is reproduced exactly 3 times in 4, else to 2^-23); "
                                                   "three bf16 pieces / fp32 = 24; accumulation, encoding and compositing are fp32"},
                       "kernel_paths": {"mlp_forward": path_fwd, "mlp_backward": path_bwd},
                       "parallelism": "ray-sharded dp%d, RCCL grad all-reduce" % world if world > 1 else "single GPU"},
            "roofline": roof,
            "roofline_other": other,
            "kernel_ms": {"forward_total": ms["fwd"], "mlp_forward": ms["mlp_fwd"], "mlp_backward": ms["mlp_bwd"],
                          "encode_backward": ms["enc_bwd"]},
            "fwd_only_samples_per_s": N / (ms["fwd"] * 1e-3),
            "extra": extra,
        }
        if world > 1:
            from permuto_sdf_amd.parallel import _mode_default
            out["dp"] = {"certificate": cert, "per_rank": per_rank, "reduce": _mode_default() + (" (reduce-scatter + all-gather per bucket)" if _mode_default() == "reduce_scatter" else ""),
                         "optimizer": (hp.last_dp or {}).get("optimizer", "replicated"),
                         "optimizer_note": "sharded: the lattice gradient is reduce-scattered in place, every rank runs AdamW on the "
                                           "1/world of the table it owns and the PARAMETERS are all-gathered (parallel.ShardedUpdate; "
                                           "PSDF_DP_OPTIMIZER=replicated: gradient all-gather, every rank updates everything)",
                         "bucket_bytes": getattr(hp, "last_bucket_bytes", None), "backend": torch.distributed.get_backend(),
                         "note": "buckets: MLP gradients first (overlap the encode backward), then the lattice gradient in two level "
                                 "ranges (the first range travels while the second is computed); comm_wait_ms = what the step still "
                                 "waits for after its last backward kernel"}
        if not args.no_cpu_baseline and world == 1:
            # The vectorised torch-CPU restatement gets SLOWER beyond ~8 threads on the 256-core host (measured:
            # 8 thr 0.7 s, 32 thr 1.1 s, 64 thr 2.2 s per 32k samples), so the baseline uses 8 threads and says so.
            # It runs in a child process under a hard timeout: the baseline must never take the GPU number down.
            import subprocess  # noqa: F811
            cores = min(8, os.cpu_count() or 1)
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", str(cores)],
                                   capture_output=True, text=True, timeout=150)
                out["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:
                out["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": cores, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
    # ---- everything below is EXTRA: the headline (`out`, rank 0) is complete.  A watchdog prints it and ends the process if the
    # extras -- which under N > 1 contain collectives -- do not come back: they must never cost the driver its line.
    import threading
    out_ref = {"out": out if rank == 0 else None}

    def _watchdog():
        if rank == 0:
            o = dict(out_ref["out"])
            o["extra"] = dict(o.get("extra", {}), watchdog="the extra measurements did not finish within their time budget")
            print(json.dumps(o), flush=True)
        os._exit(0)
    dog = threading.Timer(float(os.environ.get("PSDF_BENCH_EXTRA_BUDGET_S", "420")), _watchdog)
    dog.daemon = True
    dog.start()
    # ---- extra row (not the headline): the north_star's stated shape, 24 levels -> 52-64-64-64-1, same batch, same step
    if world == 1 and not args.no_extra:
        try:
            hp24 = SdfHotPath(nr_levels=24, hidden=64, out_channels=1, device=dev, seed=0)
            for _ in range(3):
                hp24.step(rs, rgb, normals, gt)
            torch.cuda.synchronize()
            K24 = max(5, K // 2)
            t24 = time.perf_counter()
            for _ in range(K24):
                hp24.step(rs, rgb, normals, gt)
            torch.cuda.synchronize()
            dt24 = (time.perf_counter() - t24) / K24
            extra["L24_52-64-64-64-1"] = {"ms_per_step": dt24 * 1e3, "samples_per_s": N / dt24, "steps": K24,
                                         "note": "same batch and step with a 24-level encoding (the reference's level count, "
                                                 "models.py:144)"}
            del hp24
        except Exception as e:  # the extra row must never take the headline down
            extra["L24_52-64-64-64-1"] = {"error": repr(e)}
    # ---- the SAME step with 24-bit MLP operands in both directions (three bf16 pieces, six products: operand for operand what the
    # reference's fp32 evaluation keeps), measured exactly like the headline -- K steps between barriers, max over ranks, its own
    # event brackets and roofline -- and reported at the TOP LEVEL beside `value` (the figure nobody can discount).
    # Every rank runs it (the step contains the collectives under N > 1); it comes after the headline is complete and under the
    # watchdog, so it can never cost the driver its line.
    saved_env = {k_: os.environ.get(k_) for k_ in ("PSDF_MLP_FWD_SPLIT", "PSDF_MLP_BWD_SPLIT")}

    def _restore_env():
        for k_, v_ in saved_env.items():
            if v_ is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v_
    if not os.environ.get("PSDF_BENCH_NO_24BIT"):
        try:
            os.environ["PSDF_MLP_FWD_SPLIT"] = "bf16"
            os.environ["PSDF_MLP_BWD_SPLIT"] = "bf16"
            hpb = SdfHotPath(nr_levels=NR_LEVELS, hidden=64, out_channels=1, device=dev, seed=0)
            el_b, ev_b, (pb_bwd, pb_fwd) = measure(hpb, max(3, min(args.warmup, 10)))
            tb_ = torch.tensor([el_b], dtype=torch.float64, device=cdev)
            if world > 1:
                torch.distributed.all_reduce(tb_, op=torch.distributed.ReduceOp.MAX)
            el_b = float(tb_.item())
            if rank == 0:
                ms_b = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev_b.items()}
                roof_b, other_b, _, _ = rooflines(ms_b, pb_bwd, pb_fwd)
                out["value_fp32_equivalent_24bit"] = world * N * K / el_b
                out["ms_per_step_fp32_equivalent_24bit"] = el_b / K * 1e3
                out["roofline_fp32_equivalent_24bit"] = roof_b
                out["roofline_other_fp32_equivalent_24bit"] = other_b
                out["kernel_ms_fp32_equivalent_24bit"] = {"forward_total": ms_b["fwd"], "mlp_forward": ms_b["mlp_fwd"],
                                                          "mlp_backward": ms_b["mlp_bwd"], "encode_backward": ms_b["enc_bwd"]}
                out["fp32_equivalent_24bit_note"] = (
                    "the headline's step, batch, K and timing protocol with PSDF_MLP_FWD_SPLIT=bf16 PSDF_MLP_BWD_SPLIT=bf16 (kernel paths "
                    "fwd %d, bwd %d): every fp32 MLP operand as three bf16 pieces = 24 mantissa bits, six products kept, fp32 accumulation "
                    "(fp32 rounding level against float64: tests/test_gpu_mlp.py::test_split_bf16_backward_matches_float64, "
                    "::test_split_bf16_forward_keeps_fp32_accuracy; whole step per ray at this batch size: "
                    "tests/test_gpu_fullbatch_radiance.py)" % (pb_fwd, pb_bwd))
            del hpb
        except Exception as e:
            if rank == 0:
                out["value_fp32_equivalent_24bit"] = None
                out["fp32_equivalent_24bit_note"] = "failed: %r" % (e,)
        finally:
            _restore_env()
    # ---- cfg 2 as SURVEY.md 8(d) words it -- 2 097 152 points uniform in the radius-0.5 ball instead of ray-ordered samples --,
    # unsorted and Morton-sorted, and the unsorted ball with 24-bit operands (the corner round 5 left unmeasured)
    if world == 1 and not args.no_extra:
        import copy as _copy

        def _timed(h, r, nrm, k):
            for _ in range(3):
                h.step(r, rgb, nrm, gt)
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            for _ in range(k):
                h.step(r, rgb, nrm, gt)
            torch.cuda.synchronize()
            return (time.perf_counter() - t_) / k
        Kx = max(5, K // 2)
        try:
            g = torch.Generator(device="cpu").manual_seed(parallel.rank_seed(11, rank))
            u = torch.randn(N, 3, generator=g)
            ball = (u / u.norm(dim=1, keepdim=True) * 0.5 * torch.rand(N, 1, generator=g) ** (1.0 / 3.0)).to(dev).contiguous()
            rows = {}
            for name in ("unsorted", "morton_sorted", "unsorted_24bit"):
                pts = ball
                if name == "unsorted_24bit":
                    os.environ["PSDF_MLP_FWD_SPLIT"] = "bf16"
                    os.environ["PSDF_MLP_BWD_SPLIT"] = "bf16"
                if name == "morton_sorted":
                    q = ((ball + 0.5).clamp(0, 1 - 1e-7) * 1024).to(torch.int64)

                    def _spread(v):
                        v = (v | (v << 16)) & 0x030000FF
                        v = (v | (v << 8)) & 0x0300F00F
                        v = (v | (v << 4)) & 0x030C30C3
                        return (v | (v << 2)) & 0x09249249
                    code = _spread(q[:, 0]) | (_spread(q[:, 1]) << 1) | (_spread(q[:, 2]) << 2)
                    pts = ball[torch.argsort(code)].contiguous()
                rsb = _copy.copy(rs)
                rsb.samples_pos = pts
                nb = torch.nn.functional.normalize(pts, dim=1).contiguous()
                hpc = SdfHotPath(nr_levels=NR_LEVELS, hidden=64, out_channels=1, device=dev, seed=0)
                dtc = _timed(hpc, rsb, nb, Kx)
                rows[name] = {"ms_per_step": dtc * 1e3, "samples_per_s": N / dtc, "steps": Kx,
                              "kernel_paths": {"mlp_forward": int(_lp(ctypes.c_int(2))), "mlp_backward": int(_lp(ctypes.c_int(1)))}}
                del hpc
            extra["cfg2_ball_points"] = dict(rows, note="the headline's step (same kernels, same arithmetic) on %d points drawn uniformly "
                                                         "in the radius-0.5 ball (SURVEY.md 8(d)'s wording of cfg 2) instead of "
                                                         "ray-ordered samples; the points are grouped 128 to a 'ray' for the compositing "
                                                         "stage; morton_sorted: the same points ordered by a 30-bit Morton code; "
                                                         "unsorted_24bit: the unsorted points with three-piece (24-bit) MLP operands" % N)
        except Exception as e:
            extra["cfg2_ball_points"] = {"error": repr(e)}
        finally:
            _restore_env()
    # ---- the other half of BASELINE.json's metric, "train iters/sec" (cfg 4), and the cfg 3 / cfg 5 figures.  Every one of
    # them is guarded: nothing here can take the headline down.  cfg 4 runs in this process (and this process group: under
    # N > 1 every rank steps its own rays, gradients all-reduced over RCCL); cfg 3 / cfg 5 are one-GPU inference figures and
    # run as child processes under a hard timeout, N = 1 only.
    if not args.no_extra:
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("psdf_train_bench", os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                                         "tools", "train_bench.py"))
            tb = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(tb)
            rows = {}
            for start in (0, 20000):
                if world == 1:      # a fresh process, like the cfg 3 / cfg 5 figures: isolated from this one's state and failures
                    import subprocess
                    cp = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools",
                                                                      "train_bench.py"), "--manual", "--start-iter", str(start)],
                                        capture_output=True, text=True, timeout=180)
                    r = json.loads(cp.stdout.strip().splitlines()[-1])
                else:
                    r = tb.measure(dev, manual=True, steps=60, warmup=20, repeats=3, start_iter=start)
                rows["start_iter_%d" % start] = {k: r[k] for k in ("value", "unit", "ms_per_step", "fg_samples_per_step_per_gpu",
                                                                    "rays_last_step", "steps", "warmup", "repeats_it_per_s", "backward", "dp") if k in r}
            extra["train_iters_per_s"] = {
                "metric": "train iters/sec (cfg 4: train_permuto_sdf.py's full SDF + colour + background step -- occupancy sampling, "
                          "2 rounds of importance sampling, eikonal + curvature + off-surface losses, AdamW, grid refresh every 8th "
                          "step -- on a synthetic 49-image reel; the reference's hyper-parameters, ~49 152 foreground samples per GPU "
                          "per step)",
                "n_gpus": world, "scaling": "weak", "data": "synthetic", "dtype": "f32",
                "value": rows["start_iter_0"]["value"], "value_all_levels_open": rows["start_iter_20000"]["value"],
                "note": "value: iteration counter 0 (coarse-to-fine window of the SDF lattice mostly closed); "
                        "value_all_levels_open: counter 20 000 (every level carries gradient); median of three timed blocks of 60 "
                        "steps, max over ranks", **rows}
        except Exception as e:
            extra["train_iters_per_s"] = {"error": repr(e)}
    if world == 1 and not args.no_extra:
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        for key, tool, env_extra in (("cfg3_render", "cfg3_render.py", {}),
                                     ("cfg5_sphere_trace", "sphere_trace_bench.py", {"PSDF_TRACE_WEIGHTS": "sphere_init"})):
            try:
                r = subprocess.run([sys.executable, os.path.join(here, "tools", tool)], capture_output=True, text=True, timeout=240,
                                   env=dict(os.environ, **env_extra))
                extra[key] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:
                extra[key] = {"error": repr(e)}
        try:
            extra["cfg3_ms_per_image"] = extra["cfg3_render"]["one_pool"]["ms_per_image"]
            extra["cfg5_fps"] = extra["cfg5_sphere_trace"]["fps_graph"]
        except Exception:
            pass
    if rank == 0:
        # BASELINE.json's metric has two halves: "ray-samples/sec (encode+MLP+composite) AND train iters/sec"; the second one
        # (and the cfg 3 / cfg 5 figures) at the top level as well, so that a reader of the line does not have to dig
        tis = extra.get("train_iters_per_s", {})
        if "value" in tis:
            out["train_iters_per_s"] = tis["value"]
            out["train_iters_per_s_all_levels_open"] = tis["value_all_levels_open"]
        for k in ("cfg3_ms_per_image", "cfg5_fps"):
            if k in extra:
                out[k] = extra[k]
        dog.cancel()
        print(json.dumps(out), flush=True)
    dog.cancel()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
