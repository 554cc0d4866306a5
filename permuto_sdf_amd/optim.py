"""Fused AdamW over flat fp32 tensors (csrc/optim.hip).  Same update rule as torch.optim.AdamW, which the reference
uses with betas=(0.9, 0.99), eps=1e-15 (permuto_sdf_py/train_permuto_sdf.py:293-304)."""
import torch

from . import _lib as L


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.generation = 0     # bumped by every step(): the kernels write the parameters behind torch's version counters

    SMALL = 1 << 16  # tensors below this many elements are batched into one launch (64 per launch)

    def attach(self, param, touched_rows):
        """`param` (a lattice) is updated from `touched_rows` (encoding.TouchedRows): its persistent gradient buffer, only in
        blocks of rows that a batch has touched or that carry non-zero moments; the gradient is cleared in the same launch.
        Result: bit-identical to the dense update (untouched rows' moments keep decaying exactly as torch.optim.AdamW
        makes them, train_permuto_sdf.py:293-304); blocks that no batch ever touched are never read.  Needs weight_decay 0
        for that group (with decay every row moves every step: the dense kernel is used then)."""
        if not hasattr(self, "_touched"):
            self._touched = {}
        self._touched[param] = touched_rows
        self._rebuild_active(param)

    def _rebuild_active(self, param):
        """`active[b]` must be 1 for every block whose moments are not exactly zero (the dense update moves those rows): after
        an attach() that follows dense steps, and after load_state_dict(), it is rebuilt from exp_avg / exp_avg_sq -- otherwise
        such blocks would be skipped until their next touch and the 'same result as dense AdamW' contract would not hold."""
        tr = self._touched[param]
        st = self.state.get(param)
        if not st or "exp_avg" not in st:
            tr.active.zero_()
            return
        nz = (st["exp_avg"] != 0) | (st["exp_avg_sq"] != 0)
        tr.active.copy_(nz.reshape(tr.touched.numel(), tr.block_elems).any(dim=1).view_as(tr.active).to(torch.uint8))

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.__dict__.pop("_small_ok", None)
        for p in getattr(self, "_touched", {}):
            self._rebuild_active(p)
        # a (consolidated) checkpoint holds FULL moments; under the sharded update a rank must hold zeros outside the ranges it
        # owns (parallel.consolidated_state_dict sums over the ranks): the next step(owned=...) clears what it does not own
        self._moments_loaded = True

    def state_dict(self, allow_partial=False):
        """Under the sharded data-parallel update (step(owned=...), parallel.ShardedUpdate) a rank holds the moments of the
        ranges it owns and ZEROS elsewhere: saving that as it is and resuming -- on one rank, with another world size, or in
        replicated mode -- would silently restart most of the table from zero moments.  So it is refused: use
        `parallel.consolidated_state_dict(optimizer)` (a sum over the ranks of the moments of the sharded parameters: every
        element has exactly one owner) or pass allow_partial=True for this rank's partial view."""
        sharded = getattr(self, "_sharded_params", None)
        if sharded and not allow_partial:
            from . import parallel
            if parallel.world_size() > 1:
                raise L.PsdfError("FusedAdamW.state_dict(): %d parameter(s) are updated sharded over %d ranks -- this rank holds only "
                                  "its own ranges' moments; save parallel.consolidated_state_dict(optimizer) instead"
                                  % (len(sharded), parallel.world_size()))
        return super().state_dict()

    @torch.no_grad()
    def step(self, grad_scale=1.0, owned=None):
        """owned (data parallel, parallel.ShardedUpdate): {param: [(lo, hi), ...]} -- element ranges of the FLAT parameter that
        this rank updates; the rest of such a parameter is left alone (its owner's bytes arrive with the all-gather that follows)
        and, for a touched-rows parameter, the gradient buffer and touched map outside the ranges are cleared (they hold this
        rank's local contributions, which the reduce-scatter has already delivered to their owners).  Ranges are 4-element
        aligned (touched-rows parameters: block aligned).  Moments of ranges a rank never owns stay zero and are never read."""
        import ctypes
        owned = owned or {}
        if owned:
            if not hasattr(self, "_sharded_params"):
                self._sharded_params = set()
                self._sharded_ranges = {}
            self._sharded_params.update(owned.keys())
            for p, ranges in owned.items():
                ranges = sorted((int(lo), int(hi)) for lo, hi in ranges)
                if self._sharded_ranges.get(p) != ranges or getattr(self, "_moments_loaded", False):
                    # first sharded step of this parameter (or the first one after load_state_dict / a change of ownership):
                    # moments outside the owned ranges are somebody else's -- zero them, so that "a rank holds zeros outside
                    # its ranges" (what consolidated_state_dict relies on) is a fact and not an assumption
                    st = self.state.get(p)
                    if st and "exp_avg" in st:
                        for k in ("exp_avg", "exp_avg_sq"):
                            f, prev = st[k].view(-1), 0
                            for lo, hi in ranges + [(f.numel(), f.numel())]:
                                if lo > prev:
                                    f[prev:lo].zero_()
                                prev = max(prev, hi)
                    self._sharded_ranges[p] = ranges
            self._moments_loaded = False
        self.generation += 1
        blocks_pending = []
        touched_rows = getattr(self, "_touched", {})
        small_cache = self.__dict__.setdefault("_small_ok", {})
        for group in self.param_groups:
            b1, b2 = group["betas"]
            small = {}   # step count -> [(p, g, m, v)]
            for p in group["params"]:
                tr = touched_rows.get(p)
                if tr is not None:
                    st = self.state[p]
                    if not st:
                        st["step"] = 0
                        st["exp_avg"] = torch.zeros_like(p)
                        st["exp_avg_sq"] = torch.zeros_like(p)
                    st["step"] += 1
                    if p.grad is not None:
                        # a backward that ran OUTSIDE `tr.accumulate()` handed its lattice gradient to autograd: fold it into
                        # the buffer (the forward marked its row blocks already), so that no gradient is ever silently lost
                        tr.grad.add_(p.grad)
                        p.grad = None
                    be = tr.block_elems
                    ranges = owned.get(p, [(0, p.numel())])
                    pf, gf, mf, vf = p.view(-1), tr.grad.view(-1), st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1)
                    tf, af = tr.touched.view(-1), tr.active.view(-1)
                    for lo, hi in ranges:
                        if lo % be or hi % be:
                            raise L.PsdfError("owned ranges of a touched-rows parameter must be aligned to its row blocks")
                        blo, bhi = lo // be, hi // be
                        if lo == 0 and hi == pf.numel() and group["weight_decay"] == 0.0:   # the whole tensor (no data parallel
                            blocks_pending.append((bhi, be, pf, gf, mf, vf, tf, af, group["lr"], b1, b2, group["eps"], st["step"]))
                            continue                                                        # sharding): no slicing, eight views less
                        if group["weight_decay"] != 0.0:       # every row moves: dense update from the same buffer
                            L.call("psdf_adamw_step", L.c_l(hi - lo), L.ptr(pf[lo:hi]), L.ptr(gf[lo:hi]), L.ptr(mf[lo:hi]),
                                   L.ptr(vf[lo:hi]), L.c_f(group["lr"]), L.c_f(b1), L.c_f(b2), L.c_f(group["eps"]),
                                   L.c_f(group["weight_decay"]), L.c_i(st["step"]), L.c_f(float(grad_scale)), L.stream())
                            gf[lo:hi].zero_()
                            af[blo:bhi].fill_(1)
                            tf[blo:bhi].zero_()
                        else:
                            # collected: ONE launch for all touched-rows tensors of the step (the three lattices of cfg 4 live
                            # in different parameter groups; three ~42-us launches that do not fill the chip, round 4's trace)
                            blocks_pending.append((bhi - blo, be, pf[lo:hi], gf[lo:hi], mf[lo:hi], vf[lo:hi], tf[blo:bhi], af[blo:bhi],
                                                   group["lr"], b1, b2, group["eps"], st["step"]))
                    if p in owned:      # what lies outside: local contributions whose sums live with their owners
                        prev = 0
                        for lo, hi in sorted(ranges) + [(p.numel(), p.numel())]:
                            if lo > prev:
                                gf[prev:lo].zero_()
                                tf[prev // be:lo // be].zero_()
                            prev = max(prev, hi)
                    continue
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise L.PsdfError("FusedAdamW handles contiguous fp32 parameters only")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if p in owned:          # sharded dense update: this rank's ranges only
                    pf, gf, mf, vf = p.view(-1), g.view(-1), st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1)
                    for lo, hi in owned[p]:
                        if lo % 4 or hi % 4:
                            raise L.PsdfError("owned ranges must be aligned to 4 elements")
                        L.call("psdf_adamw_step", L.c_l(hi - lo), L.ptr(pf[lo:hi]), L.ptr(gf[lo:hi]), L.ptr(mf[lo:hi]),
                               L.ptr(vf[lo:hi]), L.c_f(group["lr"]), L.c_f(b1), L.c_f(b2), L.c_f(group["eps"]),
                               L.c_f(group["weight_decay"]), L.c_i(st["step"]), L.c_f(float(grad_scale)), L.stream())
                    continue
                # (parameter and moments do not move between steps: their alignment is checked once per (moment tensors, address of
                #  the parameter) -- p.data = ..., .to() or a reloaded state are seen)
                c = small_cache.get(p)
                if c is None or c[1] is not st["exp_avg"] or c[2] is not st["exp_avg_sq"] or c[3] != p.data_ptr():
                    c = small_cache[p] = (p.numel() < self.SMALL
                                          and all(t.data_ptr() % 16 == 0 for t in (p, st["exp_avg"], st["exp_avg_sq"])),
                                          st["exp_avg"], st["exp_avg_sq"], p.data_ptr())
                small_ok = c[0]
                if small_ok and g.data_ptr() % 16 == 0:
                    small.setdefault(st["step"], []).append((p, g, st["exp_avg"], st["exp_avg_sq"]))
                    continue
                L.call("psdf_adamw_step", L.c_l(p.numel()), L.ptr(p), L.ptr(g), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]),
                       L.c_f(group["lr"]), L.c_f(b1), L.c_f(b2), L.c_f(group["eps"]), L.c_f(group["weight_decay"]),
                       L.c_i(st["step"]), L.c_f(float(grad_scale)), L.stream())
            for step, items in small.items():
                for i in range(0, len(items), 64):
                    chunk = items[i:i + 64]
                    n = len(chunk)
                    sizes = (ctypes.c_int64 * n)(*[c[0].numel() for c in chunk])
                    arrs = [(ctypes.c_void_p * n)(*[c[k].data_ptr() for c in chunk]) for k in range(4)]
                    L.call("psdf_adamw_step_multi", L.c_i(n), sizes, *arrs, L.c_f(group["lr"]), L.c_f(b1), L.c_f(b2),
                           L.c_f(group["eps"]), L.c_f(group["weight_decay"]), L.c_i(step), L.c_f(float(grad_scale)),
                           L.stream())
        for i in range(0, len(blocks_pending), 8):
            chunk = blocks_pending[i:i + 8]
            n = len(chunk)
            if n == 1:
                c = chunk[0]
                L.call("psdf_adamw_step_blocks", L.c_l(c[0]), L.c_i(c[1]), L.ptr(c[2]), L.ptr(c[3]), L.ptr(c[4]), L.ptr(c[5]),
                       L.ptr(c[6]), L.ptr(c[7]), L.c_f(c[8]), L.c_f(c[9]), L.c_f(c[10]), L.c_f(c[11]), L.c_i(c[12]),
                       L.c_f(float(grad_scale)), L.c_i(1), L.stream())
                continue
            nb = (ctypes.c_int64 * n)(*[c[0] for c in chunk])
            be = (ctypes.c_int * n)(*[c[1] for c in chunk])
            ptrs = [(ctypes.c_void_p * n)(*[c[k].data_ptr() for c in chunk]) for k in range(2, 8)]
            fl = [(ctypes.c_float * n)(*[float(c[k]) for c in chunk]) for k in range(8, 12)]
            steps = (ctypes.c_int * n)(*[int(c[12]) for c in chunk])
            L.call("psdf_adamw_step_blocks_multi", L.c_i(n), nb, be, *ptrs, *fl, steps, L.c_f(float(grad_scale)), L.c_i(1),
                   L.stream())
