"""Ray-sharded data parallelism for the hot path: one process per GPU, torch.distributed (backend "nccl" == RCCL on
ROCm, over xGMI).  The reference has no multi-GPU code at all; rays are independent units of
work, so each rank renders / trains on its own ray batch against REPLICATED parameters and a replicated occupancy
grid, and the only exchange is one sum all-reduce of the gradients per step.

Gradients are reduced in buckets that are launched as soon as they are final, so the reduction of bucket k overlaps
the backward kernels that produce bucket k+1: the small MLP gradients go first (they are ready before the encoding
backward starts), then one bucket per lattice (50 MB each at L=24).  MI355X nodes are fully connected by 7 xGMI links
per GPU; RCCL picks the algorithm, large fp32 buckets keep every link busy.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise torch.distributed from the launcher's environment; returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("PSDF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl" and os.environ.get("PSDF_BENCH_SINGLE_DEVICE") != "1":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def shard_rays(nr_rays_global, rank, world):
    """Contiguous [start, end) slice of a global ray batch owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(nr_rays_global, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def rank_seed(base_seed, rank):
    """Distinct ray-selection stream per rank; the occupancy-grid update uses `base_seed` on every rank so that the
    replicated grids stay bit-identical without communication (hard part 7)."""
    return int(base_seed) * 1000003 + int(rank) + 1


def step_seed(base_seed, rank, iteration, world=None):
    """Seed of (rank, iteration): injective in both (`rank_seed + iteration` made rank r+1 at step t-1 replay
    rank r's batch of step t).  Iterations are spaced `world` apart and the rank fills the gap."""
    world = world_size() if world is None else int(world)
    assert 0 <= int(rank) < world
    return (int(base_seed) * 1000003 + int(iteration)) * world + int(rank)


def all_reduce_max_(t):
    """in-place MAX all-reduce of a small tensor (the touched-block byte maps); no-op for one process"""
    if world_size() == 1:
        return t
    import torch.distributed as dist
    if dist.get_backend() == "gloo" and t.is_cuda:      # test-only path, as in GradientBuckets
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


class GradientBuckets:
    """Async sum all-reduce of gradient tensors, bucket by bucket; `finish()` waits for all of them."""

    def __init__(self):
        self.pending = []

    def reduce(self, tensors):
        """Launch the reduction of one bucket (a list of gradient tensors that are final).  No-op on one rank."""
        if world_size() == 1:
            return
        tensors = [t for t in tensors if t is not None]
        if not tensors:
            return
        if dist.get_backend() == "gloo" and tensors[0].is_cuda:
            # test-only path (gloo has no device collectives here): stage through host memory, synchronously
            for t in tensors:
                h = t.detach().cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                t.copy_(h)
            return
        if len(tensors) == 1:
            self.pending.append((dist.all_reduce(tensors[0], op=dist.ReduceOp.SUM, async_op=True), None, None))
            return
        flat = torch.cat([t.reshape(-1) for t in tensors])
        self.pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, tensors))

    def finish(self):
        for work, flat, tensors in self.pending:
            work.wait()
            if flat is not None:
                off = 0
                for t in tensors:
                    n = t.numel()
                    t.copy_(flat[off:off + n].view_as(t))
                    off += n
        self.pending = []


def allreduce_module_grads(modules, buckets=None):
    """Convenience for autograd-driven training loops: reduce `.grad` of every parameter of `modules`, lattices as
    their own buckets, everything else in one small bucket."""
    own = buckets is None
    buckets = buckets or GradientBuckets()
    small = []
    for m in modules:
        for name, p in m.named_parameters():
            if p.grad is None:
                continue
            if "lattice_values" in name:
                buckets.reduce([p.grad])
            else:
                small.append(p.grad)
    buckets.reduce(small)
    if own:
        buckets.finish()
    return buckets
