"""Ray-sharded data parallelism for the hot path: one process per GPU, torch.distributed (backend "nccl" == RCCL on
ROCm, over xGMI).  The reference has no multi-GPU code at all; rays are independent units of
work, so each rank renders / trains on its own ray batch against REPLICATED parameters and a replicated occupancy
grid, and the only exchange is one sum all-reduce of the gradients per step.

Gradients are reduced in buckets that are launched as soon as they are final, so the reduction of bucket k overlaps
the backward kernels that produce bucket k+1: the small MLP gradients go first (they are ready before the encoding
backward starts), then one bucket per lattice (50 MB each at L=24).  MI355X nodes are fully connected by 7 xGMI links
per GPU (point to point): every bucket is reduced as reduce-scatter + all-gather so that all links carry 1/world of it
(GradientBuckets, PSDF_DP_REDUCE), and a touched-blocks variant sends only the table blocks some rank's batch read
(GradientBuckets.reduce_blocks).  `Loopback` is a single-GPU test double of an asynchronous backend: the overlap schedule of
hotpath.backward can be checked for races without a second device.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise torch.distributed from the launcher's environment; returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    forced = os.environ.get("PSDF_DP_FORCE_COLLECTIVES") == "1"       # a one-rank group that still talks to RCCL (tests)
    if (world > 1 or forced) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("PSDF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl" and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
            # this driver stack needs dmabuf IPC for RCCL (hipIpcGetMemHandle fails otherwise) and the variable only counts before
            # the HIP runtime starts: launch scripts and bench.py export it; importing this module no longer edits the environment
            _warn_once("HSA_ENABLE_IPC_MODE_LEGACY=0 is not set in this process: RCCL may fail with 'hipIpcGetMemHandle: invalid "
                       "argument' on hosts whose driver only supports dmabuf IPC -- export it in the launcher")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl" and os.environ.get("PSDF_BENCH_SINGLE_DEVICE") != "1":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    if _loopback is not None:
        return _loopback.world
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def collectives_active():
    """True when gradient reductions are to be issued: more than one rank -- or ONE rank of an initialised process group with
    PSDF_DP_FORCE_COLLECTIVES=1, which sends every bucket through RCCL anyway (a sum over one rank): the only way to execute the
    real reduce-scatter / all-gather / all-reduce calls, their stream ordering and the level-split schedule on a single-GPU box
    (tests/test_gpu_rccl_single_rank.py)."""
    if world_size() > 1:
        return True
    return dist.is_initialized() and os.environ.get("PSDF_DP_FORCE_COLLECTIVES") == "1"


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


def seed_generators(seed, device):
    """torch.manual_seed(seed) for the two generators a training step draws from -- the CPU default generator and `device`'s --
    without torch.manual_seed's walk over every device (which, per call, goes through torch.cuda._lazy_call and formats a
    Python stack trace: ~90 us, twice per step)"""
    seed = int(seed)
    torch.default_generator.manual_seed(seed)
    dev = torch.device(device)
    if dev.type == "cuda":
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        torch.cuda.default_generators[idx].manual_seed(seed)


def all_reduce_max_(t):
    """in-place MAX all-reduce of a small tensor (the touched-block byte maps); no-op for one process"""
    if world_size() == 1 or _loopback is not None:      # identical replicas: the maximum over the ranks is the value itself
        return t
    import torch.distributed as dist
    if dist.get_backend() == "gloo" and t.is_cuda:      # test-only path, as in GradientBuckets
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def _mode_default():
    """PSDF_DP_REDUCE = all_reduce | reduce_scatter (default): see GradientBuckets"""
    return os.environ.get("PSDF_DP_REDUCE", "reduce_scatter")


class Loopback:
    """TEST DOUBLE of an asynchronous collective backend on ONE GPU (two ranks of RCCL cannot share a device, and gloo's
    device path is synchronous, so neither can show a scheduling race).  It behaves like ProcessGroupNCCL where it matters for
    correctness of the SCHEDULE: a collective is enqueued on a side stream that first waits for what the caller's current stream
    has enqueued so far, runs for a while (`delay_cycles` of device sleep, so that a consumer that forgot to wait reads stale
    data), and `wait()` makes the caller's current stream wait for it.  The 'sum over `world` ranks' of identical replicas is
    t * world.  Activate with `parallel.set_loopback(Loopback(world=2))`; world_size() then reports `world`."""

    def __init__(self, world=2, delay_cycles=2_000_000, serialize=False):
        self.world, self.delay = int(world), int(delay_cycles)
        self.stream = torch.cuda.Stream()
        self.launched = []      # (kind, numel) log for the tests
        self.serialize = bool(serialize)    # every collective is waited for at once: the race-free reference schedule

    class _Work:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    def _run(self, fn, kind, numel):
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            torch.cuda._sleep(self.delay)
            fn()
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.launched.append((kind, numel))
        if self.serialize:
            torch.cuda.current_stream().wait_event(ev)
        return Loopback._Work(ev)

    def gather_params(self, flat):
        """the parameter all-gather of the sharded update: identical virtual replicas already hold every owner's bytes, so the
        result is `flat` itself -- but WHILE the collective runs the buffer holds NaN (a real all-gather overwrites the ranges of
        the other owners while it runs): a reader that did not wait sees them"""
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            keep = flat.clone()
            flat.fill_(float("nan"))
            torch.cuda._sleep(self.delay)
            flat.copy_(keep)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        keep.record_stream(self.stream)
        self.launched.append(("all_gather_params", flat.numel()))
        if self.serialize:
            torch.cuda.current_stream().wait_event(ev)
        return Loopback._Work(ev)

    def all_reduce(self, t, op="sum"):
        return self._run((lambda: t.mul_(self.world)) if op == "sum" else (lambda: None), "all_reduce", t.numel())

    def reduce_scatter(self, shard, flat, rank=0):
        n = shard.numel()
        return self._run(lambda: shard.copy_(flat[rank * n:(rank + 1) * n] * self.world), "reduce_scatter", flat.numel())

    def all_gather(self, flat, shard, rank=0):
        # the other ranks' shards of identical replicas are this rank's own values of those ranges, summed the same way
        n = shard.numel()

        def fn():
            keep = shard.clone()
            flat.mul_(self.world)
            flat[rank * n:(rank + 1) * n].copy_(keep)
        return self._run(fn, "all_gather", flat.numel())


_loopback = None
_rs_warned = False


def set_loopback(lb):
    """install / remove (None) the single-GPU test double; returns the previous one"""
    global _loopback
    prev, _loopback = _loopback, lb
    return prev


class GradientBuckets:
    """Async SUM reduction of gradient tensors over the ranks, bucket by bucket; `finish()` waits for all of them.

    mode "reduce_scatter" (default): every bucket is reduced as reduce-scatter + all-gather.  On the fully
    connected xGMI topology of an MI355X node (7 links per GPU, point to point) each rank then receives 1/world of the bucket
    from every peer over that peer's own link and sends its reduced shard back the same way: all links busy, 2 (w-1)/w of the
    bucket per link direction, instead of a single ring that is bound by one link.  RCCL may well pick the same algorithm for a
    plain all_reduce -- issuing the two halves explicitly does not depend on its tuning tables, and lets the optimiser run on
    the owned shard between the two (not done here: the update is replicated).  mode "all_reduce": one all_reduce per bucket.
    Buckets are padded to a multiple of the world size inside a flat staging buffer when they are multi-tensor or ragged."""

    def __init__(self, mode=None):
        self.pending = []
        self.mode = mode or _mode_default()
        self.bytes = []          # per bucket, for the bench line

    # -- backends ------------------------------------------------------------------------------------------------
    @staticmethod
    def _gloo_device(tensors):
        return dist.is_initialized() and dist.get_backend() == "gloo" and tensors[0].is_cuda

    def _launch(self, flat):
        """-> list of work handles for the SUM reduction of the 1-D contiguous `flat` (numel % world == 0), in place"""
        w = world_size()
        if _loopback is not None:
            if self.mode == "all_reduce":
                return [_loopback.all_reduce(flat)]
            shard = torch.empty(flat.numel() // w, dtype=flat.dtype, device=flat.device)
            shard.record_stream(_loopback.stream)        # used on the side stream after this frame has dropped it
            return [_loopback.reduce_scatter(shard, flat), _loopback.all_gather(flat, shard)]
        if self.mode == "all_reduce":
            return [dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)]
        r = dist.get_rank()
        n = flat.numel() // w
        shard = torch.empty(n, dtype=flat.dtype, device=flat.device)
        if dist.get_backend() == "nccl":
            # both enqueue on the process group's stream, in order: the gather starts when the scatter's result is there
            try:
                a = dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, async_op=True)
            except Exception as e:      # a backend build without the tensor form: say so once and use one all_reduce per bucket
                global _rs_warned
                if not _rs_warned:
                    _rs_warned = True
                    import warnings
                    warnings.warn("GradientBuckets: reduce_scatter_tensor failed (%r); falling back to all_reduce" % (e,))
                self.mode = "all_reduce"
                return [dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)]
            b = dist.all_gather_into_tensor(flat, shard, async_op=True)
            return [a, b, shard]
        # gloo (CPU tests) has no reduce_scatter: one reduce per owner, then the gather
        for o in range(w):
            part = flat[o * n:(o + 1) * n]
            dist.reduce(part, dst=o, op=dist.ReduceOp.SUM)
            if o == r:
                shard.copy_(part)
        parts = [torch.empty_like(shard) for _ in range(w)]
        dist.all_gather(parts, shard)
        flat.copy_(torch.cat(parts))
        return []

    def reduce(self, tensors):
        """Launch the reduction of one bucket (a list of gradient tensors that are final).  No-op on one rank."""
        if not collectives_active():
            return
        tensors = [t for t in tensors if t is not None]
        if not tensors:
            return
        self.bytes.append(sum(t.numel() * t.element_size() for t in tensors))
        if _loopback is None and self._gloo_device(tensors):
            # test-only path (gloo has no device collectives here): stage through host memory, synchronously
            host = [t.detach().cpu().reshape(-1) for t in tensors]
            flat, pad = self._flatten(host)
            for wk in self._launch(flat):
                if hasattr(wk, "wait"):
                    wk.wait()
            off = 0
            for t in tensors:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
            return
        w = world_size()
        if len(tensors) == 1 and tensors[0].is_contiguous() and tensors[0].numel() % w == 0:
            self.pending.append((self._launch(tensors[0].view(-1)), None, None))
            return
        if len(tensors) == 1 and tensors[0].is_contiguous() and tensors[0].numel() >= (1 << 20):
            # a LARGE single tensor that does not cut into `world` equal parts (a 50-MB lattice gradient on 3, 5, 6 or 7 GPUs):
            # one in-place all_reduce instead of a padded staging copy + a copy back that nothing overlaps
            flat = tensors[0].view(-1)
            if _loopback is not None:
                self.pending.append(([_loopback.all_reduce(flat)], None, None))
            else:
                self.pending.append(([dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)], None, None))
            return
        flat, _ = self._flatten([t.reshape(-1) for t in tensors])
        self.pending.append((self._launch(flat), flat, tensors))

    @staticmethod
    def _flatten(parts):
        w = world_size()
        n = sum(p.numel() for p in parts)
        pad = (-n) % w
        if pad:
            parts = list(parts) + [torch.zeros(pad, dtype=parts[0].dtype, device=parts[0].device)]
        return torch.cat(parts), pad

    def reduce_blocks(self, grad, touched, block_elems):
        """Touched-blocks (sparse) reduction of a lattice gradient: `touched` [n_blocks] bytes has ALREADY been OR-reduced over
        the ranks (all_reduce_max_), so every rank holds the same set; only those blocks of `grad` (viewed [n_blocks,
        block_elems]) travel.  Blocks nobody touched carry an all-zero gradient on every rank: skipping them changes nothing.
        Costs one host sync (the number of touched blocks sizes the message).  Worth it when a batch touches a small part of
        the table: the coarse levels always, the hashed levels only for small batches."""
        if not collectives_active():
            return
        idx = torch.nonzero(touched.reshape(-1), as_tuple=False).reshape(-1)
        if idx.numel() == 0:
            return
        g2 = grad.view(-1, block_elems)
        if idx.numel() == g2.shape[0]:
            return self.reduce([grad])
        compact = g2.index_select(0, idx)
        self.reduce([compact])
        # the staging buffer of reduce() is written back into `compact` at finish(); scatter it home afterwards
        self.pending.append(([], None, None, (g2, idx, compact)))

    def finish(self):
        for item in self.pending:
            works, flat, tensors = item[:3]
            for wk in works:
                if hasattr(wk, "wait"):
                    wk.wait()
            if flat is not None:
                off = 0
                for t in tensors:
                    n = t.numel()
                    t.copy_(flat[off:off + n].view_as(t))
                    off += n
            if len(item) == 4:
                g2, idx, compact = item[3]
                g2.index_copy_(0, idx, compact)
        self.pending = []


def shard_bounds(n, unit=1, world=None, rank_=None):
    """[lo, hi) of this rank's share when `n` elements are cut into `world` equal, contiguous, `unit`-aligned parts -- or None
    when they cannot be (n is not a multiple of world * unit): the caller then keeps the replicated update for that tensor."""
    world = world_size() if world is None else int(world)
    r = rank() if rank_ is None else int(rank_)
    if n % (world * unit) != 0:
        return None
    per = n // world
    return r * per, (r + 1) * per


def _inplace_collectives():
    """PSDF_DP_INPLACE=1: reduce_scatter_tensor / all_gather_into_tensor with the shard ALIASING its slot of the full buffer"""
    return os.environ.get("PSDF_DP_INPLACE") == "1"


_warned = set()


def _warn_once(msg):
    if msg not in _warned:
        _warned.add(msg)
        import warnings
        warnings.warn(msg)


def sharded_optimizer_default():
    """PSDF_DP_OPTIMIZER = sharded (default) | replicated"""
    return os.environ.get("PSDF_DP_OPTIMIZER", "sharded") == "sharded"


class ShardedUpdate:
    """The optimiser SHARDED over the ranks for the large tensors (the lattices; ZeRO-1 in the usual vocabulary).

    Replicated update (GradientBuckets): reduce-scatter the gradient, all-gather the GRADIENT, every rank runs AdamW over the
    whole 50-MB table -- world times the same arithmetic and the same 28 B/parameter of HBM traffic.  Sharded: after the
    reduce-scatter a rank already OWNS the sum of 1/world of the table's gradient; it updates exactly that range (its moments
    for the other ranges stay zero and are never read) and the ranks all-gather the PARAMETERS.  Bytes on the links are the same
    (a reduce-scatter and an all-gather of the same size); AdamW work and traffic per rank drop to 1/world, and the gradient's
    all-gather -- which nothing but the replicated update needed -- is gone.  Replicas stay bit-identical by construction:
    every rank holds the owner's bytes.

        su = ShardedUpdate()
        own = su.reduce_scatter(flat_grad)            # async: flat_grad[own[0]:own[1]] becomes the sum over ranks ...
        ...                                           # (more buckets / backward kernels)
        su.wait()                                     # ... once this returns: the caller's stream now sees the sums
        <update param[own] from flat_grad[own]>       # optim.FusedAdamW.step(owned={param: [own]})
        su.all_gather(flat_param, own)                # async; su.wait() before the parameters are read again

    `reduce_scatter` returns None (and does nothing) when the tensor cannot be cut evenly: reduce it through GradientBuckets
    and update it replicated.  Backends: nccl (= RCCL: a separate shard + a copy into the owned range by default, the documented
    in-place forms -- recv buffer = send buffer + rank * count -- with PSDF_DP_INPLACE=1), gloo
    (CPU tests; device tensors are staged through host memory, synchronously), Loopback (one GPU, identical virtual replicas:
    the owner's range is scaled by `world`; `virtual_ranks()` lets the caller play every owner in turn)."""

    def __init__(self):
        self.works = []
        self.copies = []     # (owned range, shard) pairs to copy once their reduce-scatter has completed
        self.keep = []       # sources of collectives in flight
        self.bytes = []

    @staticmethod
    def virtual_ranks():
        """the ranks whose owned ranges THIS process must update: its own -- or all of them under the Loopback double"""
        return list(range(_loopback.world)) if _loopback is not None else [rank()]

    def reduce_scatter(self, flat, unit=1):
        assert flat.dim() == 1 and flat.is_contiguous()
        if not collectives_active():
            return (0, flat.numel())
        own = shard_bounds(flat.numel(), unit)
        if own is None:
            return None
        self.bytes.append(flat.numel() * flat.element_size())
        lo, hi = own
        if _loopback is not None:
            self.works.append(_loopback.all_reduce(flat))          # every virtual owner's range holds its 'sum'
            return own
        w, r = world_size(), dist.get_rank()
        n = hi - lo
        if dist.get_backend() == "nccl":
            if _inplace_collectives():
                # receive buffer = send buffer + rank * count: NCCL / RCCL's documented in-place form (no staging at all)
                self.works.append(dist.reduce_scatter_tensor(flat[lo:hi], flat, op=dist.ReduceOp.SUM, async_op=True))
            else:
                # default: the textbook form -- a separate shard (1/world of the bucket), copied into the owned range once the
                # collective has completed (wait()).  No multi-GPU box has run this code yet: the form
                # every RCCL build is exercised with daily is the safe default, the in-place one an option to measure
                shard = torch.empty(n, dtype=flat.dtype, device=flat.device)
                self.works.append(dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, async_op=True))
                self.copies.append((flat[lo:hi], shard))
            return own
        h = flat.detach().cpu() if flat.is_cuda else flat
        for o in range(w):
            dist.reduce(h[o * n:(o + 1) * n], dst=o, op=dist.ReduceOp.SUM)
        if flat.is_cuda:
            flat[lo:hi].copy_(h[lo:hi])
        return own

    def all_gather(self, flat, own):
        """every rank's [own) range of `flat` (same bounds rule on every rank) -> all of `flat`, in place"""
        if not collectives_active():
            return
        if _loopback is not None:
            self.works.append(_loopback.gather_params(flat))
            return
        lo, hi = own
        if dist.get_backend() == "nccl":
            src = flat[lo:hi] if _inplace_collectives() else flat[lo:hi].clone()
            self.works.append(dist.all_gather_into_tensor(flat, src, async_op=True))
            self.keep.append(src)
            return
        h = flat.detach().cpu() if flat.is_cuda else flat
        parts = [torch.empty(hi - lo, dtype=h.dtype) for _ in range(world_size())]
        dist.all_gather(parts, h[lo:hi].clone())
        with torch.no_grad():
            flat.copy_(torch.cat(parts))

    def wait(self):
        for wk in self.works:
            wk.wait()
        for dst, src in self.copies:      # (the caller's stream has just been ordered behind the collectives)
            dst.copy_(src)
        self.works, self.copies, self.keep = [], [], []


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


def consolidated_state_dict(optimizer):
    """state_dict of an optimiser whose large parameters are updated sharded (ShardedUpdate): the moments of those parameters
    summed over the ranks -- a rank's moments outside the ranges it owns are zero and every element has exactly one owner, so the
    sum IS the full state -- in a copy (the live state keeps its sharded form: a later consolidation must not add gathered copies
    to the owners' values).  Collective: call it on every rank; every rank returns the same dict.  Without sharded parameters or
    with one rank it is optimizer.state_dict()."""
    sd = optimizer.state_dict(allow_partial=True) if "allow_partial" in optimizer.state_dict.__code__.co_varnames else optimizer.state_dict()
    sharded = getattr(optimizer, "_sharded_params", None)
    if not sharded or world_size() <= 1 or _loopback is not None:
        return sd
    # param -> index in the state dict (torch numbers parameters in group order)
    index, i = {}, 0
    for g in optimizer.param_groups:
        for p in g["params"]:
            index[p] = i
            i += 1
    state = dict(sd["state"])
    owned_ranges = getattr(optimizer, "_sharded_ranges", {})
    for p in sorted(sharded, key=lambda q: index[q]):
        st = dict(state[index[p]])
        for k in ("exp_avg", "exp_avg_sq"):
            t = st[k].detach().clone()
            # only the ranges this rank OWNS enter the sum (after load_state_dict of a consolidated checkpoint every
            # rank holds full moments until its next sharded step; stale copies must never be added to the owner's values)
            if p in owned_ranges:
                f, prev = t.view(-1), 0
                for lo, hi in list(owned_ranges[p]) + [(f.numel(), f.numel())]:
                    if lo > prev:
                        f[prev:lo].zero_()
                    prev = max(prev, hi)
            if dist.get_backend() == "nccl":
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
            else:
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                t.copy_(h)
            st[k] = t
        state[index[p]] = st
    sd = dict(sd)
    sd["state"] = state
    return sd
