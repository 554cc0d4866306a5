"""GPU: the REAL collective calls of the data-parallel path on a single-GPU box.  A process group of ONE rank with backend nccl
(= RCCL) and PSDF_DP_FORCE_COLLECTIVES=1 sends every gradient bucket through RCCL anyway -- a sum over one rank is the identity --
so `reduce_scatter_tensor` / `all_gather_into_tensor` / `all_reduce` are really enqueued on RCCL's stream, asynchronously, with the
padding, staging, level-split schedule and `finish()` ordering of the N > 1 run; the result must equal the run without them.
(What a single rank cannot show is the arithmetic of the sum itself: tests/test_dp_gloo.py, world 2.)  Runs in a subprocess: the
process group must not leak into the other tests."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import json, os, sys, torch
sys.path.insert(0, os.environ["PSDF_ROOT"])
import bench
from permuto_sdf_amd import parallel
from permuto_sdf_amd.hotpath import SdfHotPath
from permuto_sdf_amd.neus import l1_loss_raw
dev = torch.device("cuda:0")
rs, rgb, aux = bench.make_batch(dev, 11, nr_rays=4096)
normals, gt = aux[4], aux[5]

def run(reduce):
    hp = SdfHotPath(nr_levels=16, hidden=64, out_channels=1, device=dev, seed=0)
    pred, saved = hp.forward(rs, rgb, normals)
    loss, g_pred = l1_loss_raw(pred, gt)
    out = hp.backward(rs, rgb, saved, g_pred, reduce=reduce, optimizer_step=True)
    torch.cuda.synchronize()
    return [g.clone() for g in out["grads"]], [p.detach().clone() for p in hp.params], getattr(hp, "last_bucket_bytes", None)

g0, p0, _ = run(False)
res = {}
rank, world, local = parallel.init(backend="nccl")
assert world == 1 and torch.distributed.get_backend() == "nccl" and parallel.collectives_active()
for mode in ("reduce_scatter", "all_reduce", "sharded", "sharded_inplace"):
    os.environ["PSDF_DP_REDUCE"] = "reduce_scatter" if mode.startswith("sharded") else mode
    os.environ["PSDF_DP_INPLACE"] = "1" if mode == "sharded_inplace" else "0"
    # sharded: in-place reduce_scatter_tensor of the lattice ranges, owner update, in-place all_gather_into_tensor of the
    # PARAMETERS (parallel.ShardedUpdate) -- with one rank the owned range is everything, the calls are the real ones
    os.environ["PSDF_DP_OPTIMIZER"] = "sharded" if mode.startswith("sharded") else "replicated"
    g1, p1, bytes_ = run(True)
    # level-split launches change nothing within a level; a sum over one rank is the identity
    gerr = max(float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)) for a, b in zip(g0, g1))
    perr = max(float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)) for a, b in zip(p0, p1))
    res[mode] = {"grad_rel": gerr, "param_rel": perr, "bucket_bytes": bytes_}
# odd-sized multi-tensor bucket + touched-blocks reduction through the real backend
b = parallel.GradientBuckets(mode="reduce_scatter")
t1, t2 = torch.randn(1001, device=dev), torch.randn(7, 3, device=dev)
c1, c2 = t1.clone(), t2.clone()
b.reduce([t1, t2])
lat = torch.zeros(4, 1024, 2, device=dev)
lat[1, 100:164] = 1.5
touched = (lat.view(-1, 64) != 0).any(1).to(torch.uint8)
want = lat.clone()
b.reduce_blocks(lat, touched, 64)
b.finish()
torch.cuda.synchronize()
res["ragged_bucket_ok"] = bool(torch.equal(t1, c1) and torch.equal(t2, c2) and torch.equal(lat, want))
res["bytes"] = b.bytes
torch.distributed.barrier()
torch.distributed.destroy_process_group()
print("RESULT " + json.dumps(res))
'''


def test_real_rccl_collectives_on_one_rank(dev):
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511",
               PSDF_DP_FORCE_COLLECTIVES="1", PSDF_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[7:])
    print(res)
    for mode in ("reduce_scatter", "all_reduce", "sharded", "sharded_inplace"):
        assert res[mode]["grad_rel"] <= 5e-5 and res[mode]["param_rel"] <= 1e-6, res
        assert len(res[mode]["bucket_bytes"]) == 3            # MLP bucket + two lattice level ranges
    # rows 100..163 of level 1, blocks of 64 floats = 32 rows: three touched blocks travel, not the 32 KB table
    assert res["ragged_bucket_ok"] and res["bytes"][1] == 3 * 64 * 4


def test_training_step_schedule_through_real_rccl_on_one_rank(dev):
    """The cfg-4 hand-written step with its round-6 data-parallel schedule -- early reduce-scatter per lattice, sharded AdamW,
    parameter all-gather left in flight and waited for where the next step reads each table -- through the REAL RCCL calls
    (`reduce_scatter_tensor`, `all_gather_into_tensor`, async) on a one-rank group: the stream ordering between RCCL's stream and
    the compute stream is the real one; tools/train_bench.py asserts finite parameters at the end."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29513",
               PSDF_DP_FORCE_COLLECTIVES="1", PSDF_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_bench.py"), "--manual", "--steps", "12", "--warmup", "4",
                        "--repeats", "1", "--start-iter", "20000"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(d["value"], d["dp"])
    assert d["value"] > 0 and d["dp"] and d["dp"]["optimizer"] == "sharded" and d["dp"]["deferred_gather"] is True
    b = d["dp"]["comm_bytes_per_step_per_gpu"]
    assert b["all_gather_parameters"] == 3 * 24 * (1 << 18) * 2 * 4 and b["reduce_scatter_gradients"] >= b["all_gather_parameters"]
