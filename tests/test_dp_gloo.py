"""CPU, world_size 2 (gloo): the ray-sharded data-parallel contract.  Each rank computes the gradients of ITS ray
shard (with the CPU oracle standing in for the kernels), the buckets are sum-all-reduced through
permuto_sdf_amd.parallel.GradientBuckets, and the result must equal the single-process gradient of the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    from oracle import permuto_oracle as po
    g = torch.Generator().manual_seed(0)
    L_, T = 4, 2 ** 10
    lat, sh = po.make_params(3, T, L_, 2, seed=1, init_scale=0.5)
    pts = torch.rand(400, 3, generator=g) - 0.5
    w1 = torch.randn(po.output_dims(3, L_, 2, True), 1, generator=g)
    return po, lat, sh, pts, w1, np.geomspace(1.0, 0.05, L_), torch.ones(L_)


def _grads(po, lat, sh, pts, w1, sl, win):
    lat = lat.clone().requires_grad_(True)
    w = w1.clone().requires_grad_(True)
    out = torch.tanh(po.encode(pts, lat, sl, sh, win, True, 1e-3) @ w)
    out.sum().backward()            # per-ray losses are summed; the driver divides by the global ray count
    return lat.grad, w.grad


def _worker(rank, world, port, ret, mode, sparse):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from permuto_sdf_amd import parallel
    r, w, _ = parallel.init(backend="gloo")
    assert (r, w) == (rank, world) and parallel.world_size() == world
    po, lat, sh, pts, w1, sl, win = _problem()
    s, e = parallel.shard_rays(pts.shape[0], rank, world)
    g_lat, g_w = _grads(po, lat, sh, pts[s:e], w1, sl, win)
    b = parallel.GradientBuckets(mode=mode)
    b.reduce([g_w, torch.zeros(3)])     # small multi-tensor bucket (flattened, padded to a multiple of the world size)
    if sparse:
        # touched-blocks reduction: blocks of 16 table rows; a block is touched where THIS rank's gradient is non-zero, the
        # byte maps are OR-reduced (MAX) first so that every rank sends the same set of blocks
        block_elems = 16 * 2
        touched = (g_lat.view(-1, block_elems) != 0).any(1).to(torch.uint8)
        parallel.all_reduce_max_(touched)
        ret["touched_frac_%d" % rank] = float(touched.float().mean())
        b.reduce_blocks(g_lat, touched, block_elems)
    else:
        b.reduce([g_lat])                   # one bucket per lattice
    b.finish()
    ret["bytes_%d" % rank] = list(b.bytes)
    if rank == 0:
        ret["g_lat"], ret["g_w"] = g_lat.numpy(), g_w.numpy()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,sparse", [("all_reduce", False), ("reduce_scatter", False), ("reduce_scatter", True),
                                         ("all_reduce", True)])
def test_ray_sharded_gradients_equal_full_batch(mode, sparse):
    """both bucket algorithms (one all_reduce; reduce-scatter + all-gather) and the touched-blocks variant
    give the full-batch gradient"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, mode, sparse), nprocs=world, join=True)
    po, lat, sh, pts, w1, sl, win = _problem()
    g_lat, g_w = _grads(po, lat, sh, pts, w1, sl, win)
    assert np.abs(ret["g_lat"] - g_lat.numpy()).max() <= 1e-5 * g_lat.abs().max().item()
    assert np.abs(ret["g_w"] - g_w.numpy()).max() <= 1e-5 * g_w.abs().max().item()
    if sparse:
        assert ret["touched_frac_0"] == ret["touched_frac_1"] and 0.0 < ret["touched_frac_0"] < 1.0
        assert ret["bytes_0"][1] < g_lat.numel() * 4          # fewer bytes travelled than the dense bucket
        assert ret["bytes_0"] == ret["bytes_1"]


def _sharded_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from permuto_sdf_amd import parallel
    parallel.init(backend="gloo")
    po, lat, sh, pts, w1, sl, win = _problem()
    s, e = parallel.shard_rays(pts.shape[0], rank, world)
    g_lat, _ = _grads(po, lat, sh, pts[s:e], w1, sl, win)
    param = lat.clone()
    su = parallel.ShardedUpdate()
    flat_g = g_lat.contiguous().view(-1)
    own = su.reduce_scatter(flat_g, unit=4)                 # in place: MY range of flat_g is now the sum over the ranks
    su.wait()
    ret["sharded_%d" % rank] = own is not None
    if own is None:       # 4 * 1024 * 2 elements do not cut into 3 aligned parts: what every caller does then -- replicated
        b = parallel.GradientBuckets()
        b.reduce([flat_g])
        b.finish()
        with torch.no_grad():
            param.view(-1).sub_(0.1 * flat_g)
    else:
        assert own == parallel.shard_bounds(flat_g.numel(), 4) and own[1] - own[0] == flat_g.numel() // world
        lo, hi = own
        with torch.no_grad():
            param.view(-1)[lo:hi] -= 0.1 * flat_g[lo:hi]       # the owner's update (plain SGD stands in for the AdamW kernel)
        su.all_gather(param.view(-1), own)                      # the owners' bytes to everybody
        su.wait()
    ret["param_%d" % rank] = param.numpy()
    assert su.reduce_scatter(torch.zeros(world * 4 + 2), unit=4) is None      # cannot be cut evenly: the caller keeps it replicated
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_optimizer_protocol_equals_the_full_batch_update(world):
    """parallel.ShardedUpdate: in-place reduce-scatter of the lattice gradient, the owner updates its 1/world of the table,
    all-gather of the PARAMETERS -- every rank ends with the single-process update of the whole batch, bit-identical replicas"""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sharded_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    po, lat, sh, pts, w1, sl, win = _problem()
    g_lat, _ = _grads(po, lat, sh, pts, w1, sl, win)
    want = (lat - 0.1 * g_lat).numpy()
    assert all(ret["sharded_%d" % r] == (world in (2, 4)) for r in range(world))      # 8 192 elements do not cut into 3 x 4k
    for r in range(world):
        assert np.abs(ret["param_%d" % r] - want).max() <= 1e-5 * np.abs(want).max()
        assert np.array_equal(ret["param_%d" % r], ret["param_0"])          # replicas: the same bytes


def test_single_process_is_a_noop():
    from permuto_sdf_amd import parallel
    assert parallel.world_size() == 1
    t = torch.ones(4)
    b = parallel.GradientBuckets()
    b.reduce([t])
    b.finish()
    assert torch.equal(t, torch.ones(4))


def _consolidate_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from permuto_sdf_amd import parallel
    parallel.init(backend="gloo")
    n = 4096
    big, small = torch.nn.Parameter(torch.zeros(n)), torch.nn.Parameter(torch.zeros(8))
    opt = torch.optim.Adam([big, small], lr=1e-3)
    lo, hi = parallel.shard_bounds(n, 4)
    m, v = torch.zeros(n), torch.zeros(n)
    m[lo:hi] = torch.arange(lo, hi, dtype=torch.float32) + 1.0           # the owner's moments; zero elsewhere (never written)
    v[lo:hi] = (torch.arange(lo, hi, dtype=torch.float32) + 1.0) ** 2
    opt.state[big] = {"step": torch.tensor(3.0), "exp_avg": m, "exp_avg_sq": v}
    opt.state[small] = {"step": torch.tensor(3.0), "exp_avg": torch.full((8,), 0.5), "exp_avg_sq": torch.full((8,), 0.25)}
    opt._sharded_params = {big}
    sd = parallel.consolidated_state_dict(opt)
    ret["m_%d" % rank] = sd["state"][0]["exp_avg"].numpy()
    ret["v_%d" % rank] = sd["state"][0]["exp_avg_sq"].numpy()
    ret["small_%d" % rank] = sd["state"][1]["exp_avg"].numpy()
    ret["live_%d" % rank] = float(opt.state[big]["exp_avg"].abs().sum())   # the live state keeps its sharded form
    ret["own_%d" % rank] = float(m.abs().sum())
    dist.barrier()
    dist.destroy_process_group()


def test_consolidated_state_dict_gathers_the_sharded_moments():
    """parallel.consolidated_state_dict: the moments of a sharded parameter summed over the ranks (zero outside a rank's own
    range, one owner per element) = the full state, on every rank, in a COPY; replicated parameters untouched."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_consolidate_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    full = np.arange(4096, dtype=np.float32) + 1.0
    for r in range(world):
        assert np.array_equal(ret["m_%d" % r], full) and np.array_equal(ret["v_%d" % r], full ** 2)
        assert np.array_equal(ret["small_%d" % r], np.full(8, 0.5, np.float32))
        assert ret["live_%d" % r] == ret["own_%d" % r]


def _roundtrip_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from permuto_sdf_amd import parallel
    parallel.init(backend="gloo")
    n = 4096
    big = torch.nn.Parameter(torch.zeros(n))
    opt = torch.optim.Adam([big], lr=1e-3)
    lo, hi = parallel.shard_bounds(n, 4)
    full_m = torch.arange(n, dtype=torch.float32) + 1.0
    # the state right after load_state_dict() of a CONSOLIDATED checkpoint: every rank holds the FULL moments ...
    opt.state[big] = {"step": torch.tensor(3.0), "exp_avg": full_m.clone(), "exp_avg_sq": full_m.clone() ** 2}
    opt._sharded_params = {big}
    opt._sharded_ranges = {big: [(lo, hi)]}
    # ... the owner then takes a step on its range (stands in for FusedAdamW.step(owned=...)); the stale copies elsewhere stay
    with torch.no_grad():
        opt.state[big]["exp_avg"][lo:hi] += 0.5
    sd = parallel.consolidated_state_dict(opt)
    ret["m_%d" % rank] = sd["state"][0]["exp_avg"].numpy()
    ret["v_%d" % rank] = sd["state"][0]["exp_avg_sq"].numpy()
    dist.barrier()
    dist.destroy_process_group()


def test_consolidation_after_a_resume_counts_every_element_once():
    """save -> load -> step -> save: after a resume every rank holds full moments; the second consolidated save must
    hold the OWNER's updated value of every element, not owner + (world - 1) stale copies."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_roundtrip_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    full = np.arange(4096, dtype=np.float32) + 1.0
    for r in range(world):
        assert np.array_equal(ret["m_%d" % r], full + 0.5) and np.array_equal(ret["v_%d" % r], full ** 2)
