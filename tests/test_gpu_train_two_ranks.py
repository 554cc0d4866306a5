"""GPU: the cfg-4 trainers with TWO ranks on one device (backend gloo, gradients staged through the host: slow, but the whole
N > 1 code path of the step runs -- per-rank rays, bucketed gradient sum of the dense parameters and of the three lattice
buffers, OR of the touched-block maps, 1 / world scaling in the optimiser).  tools/train_bench.py asserts at the end that every
parameter is bit-identical on both ranks; here both the autograd trainer and the hand-written step (train_manual.py) go through it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("manual", [False, True])
def test_two_rank_training_keeps_replicas_identical(manual):
    env = dict(os.environ)
    env.update(PSDF_BENCH_SINGLE_DEVICE="1", PSDF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29541 + int(manual)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "train_bench.py"), "--steps", "6", "--warmup", "3", "--repeats", "1"]
    if manual:
        cmd.append("--manual")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["fg_samples_per_step_per_gpu"] > 1000


def test_two_rank_bench_line_certifies_itself():
    """bench.py under N = 2 (both ranks on cuda:0, gloo): the line carries dp.certificate -- one device identity per rank, the
    collective backend, and MIN == MAX over ranks of a parameter checksum after the timed steps (replicas bit-identical).  On
    real hardware the same block says `distinct_devices == n_gpus` and names the RCCL version."""
    env = dict(os.environ)
    env.update(PSDF_BENCH_SINGLE_DEVICE="1", PSDF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-extra"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = d["dp"]["certificate"]
    assert d["n_gpus"] == 2 and "error" not in c, c
    assert len(c["devices"]) == 2 and c["single_device_development_run"] and c["distinct_devices"] == 1
    assert c["replicas_bit_identical"] is True and c["backend"] == "gloo"
