"""Host-side mirror of the reference's pybind11 module `permuto_sdf` (boundary #1, src/PyBridge.cxx:36-166).

Same class names, method names, argument order, return tuples and in-place/aliasing behaviour as the reference:
``Sphere`` (:49-57), ``OccupancyGrid`` (:60-81), ``RaySamplesPacked`` (:83-103), ``VolumeRendering`` (:105-122),
``RaySampler`` (:124-128), ``PermutoSDF`` statics (:36-47), ``TrainParams`` (:133-144).  Every method allocates its
outputs with torch on the inputs' device (the reference hard-codes cuda:0) and launches hand-written HIP kernels
through the C ABI on torch's CURRENT stream (the reference uses the NULL stream).  Argument errors raise
``ValueError`` where the reference aborts the process through loguru CHECK.
"""
import math

import numpy as np

import torch

from . import _lib as L

_COUNTS_ON_HOST_MAX_RAYS = 32768      # compact_to_valid_samples: up to here the march's per-ray counts travel instead of its total

MASK64 = (1 << 64) - 1
PCG_DEFAULT_STATE = 0x853C49E6748FEA9B
PCG_DEFAULT_STREAM = 0xDA3E39CB94B95BDB
PCG_MULT = 0x5851F42D4C957F2D


class Pcg32:
    """Host copy of the generator the kernels receive by value (reference kernels/permuto_sdf/pcg32.h:45-206:
    default seed, and ``advance()`` by 2^32 on the host after every jittered launch, e.g. src/OccupancyGrid.cu:252-254)."""

    def __init__(self, state=PCG_DEFAULT_STATE, inc=PCG_DEFAULT_STREAM):
        self.state, self.inc = state & MASK64, inc & MASK64

    def advance(self, delta=1 << 32):
        delta &= MASK64
        cur_mult, cur_plus, acc_mult, acc_plus = PCG_MULT, self.inc, 1, 0
        while delta > 0:
            if delta & 1:
                acc_mult = (acc_mult * cur_mult) & MASK64
                acc_plus = (acc_plus * cur_mult + cur_plus) & MASK64
            cur_plus = ((cur_mult + 1) * cur_plus) & MASK64
            cur_mult = (cur_mult * cur_mult) & MASK64
            delta >>= 1
        self.state = (acc_mult * self.state + acc_plus) & MASK64

    def next_uint(self):
        old = self.state
        self.state = (old * PCG_MULT + self.inc) & MASK64
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF

    def args(self):
        return L.c_u64(self.state), L.c_u64(self.inc)


def _f32c(t):
    if t.dtype is torch.float32 and t.is_contiguous():      # (the usual case, without two dispatcher round trips)
        return t
    return t.to(torch.float32).contiguous()


def _check2d(t, cols, name):
    if t.dim() != 2 or (cols is not None and t.shape[1] != cols):
        raise ValueError("%s should have shape [N, %s], got %s" % (name, cols, tuple(t.shape)))


def _vec3(v):
    if type(v) is list and len(v) == 3 and type(v[0]) is float:
        return v
    # a tensor this package made from a host vector (Sphere.m_center_tensor: what train_permuto_sdf.py hands to
    # compute_samples_bg every iteration) carries that vector: no read-back -- a stream synchronisation -- while it is unmodified
    h = getattr(v, "_psdf_host3", None)
    if h is not None and h[1] == v._version:
        return h[0]
    v = [float(x) for x in (v.tolist() if hasattr(v, "tolist") else v)]
    if len(v) != 3:
        raise ValueError("expected a 3-vector")
    return v


def _host3(v):
    return (L.c_f * 3)(*v)


# ---------------------------------------------------------------------------------------------- container
class RaySamplesPacked:
    """SoA container of per-sample tensors + per-ray [start,end) ranges (include/permuto_sdf/RaySamplesPacked.cuh:6-46)."""

    def __init__(self, nr_rays, nr_samples_maximum, device=None, _alloc=True):
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        # (straight into __dict__: a training step builds nine of these, and every attribute through __setattr__ below is host
        #  time of a host-bound step; the values are exactly what the assignments would store)
        d = self.__dict__
        d["m_nr_rays"] = int(nr_rays)
        d["max_nr_samples"] = int(nr_samples_maximum)
        d["rays_have_equal_nr_of_samples"] = False
        d["fixed_nr_of_samples_per_ray"] = 0
        d["has_sdf"] = False
        d["_exact"] = False  # set by producers whose output is already hole-free and ray ordered
        d["_dense"] = False  # set by compaction: every slot of the sample tensors belongs to a ray (outputs need no zero fill)
        # host-side knowledge that spares later compactions their sync (round 4): the per-ray counts of a march (device tensor),
        # the number of non-empty rays once those counts have been on the host, the exact total of a merge derived from them
        d["_ray_counts"] = None
        d["_host_nonempty"] = None
        d["_known_total"] = None
        if _alloc:
            M, R = d["max_nr_samples"], d["m_nr_rays"]
            f32 = torch.float32
            d["cur_nr_samples"] = L.zeroed_int(dev)      # a [1] int32 slice of a pooled, pre-zeroed buffer
            d["samples_pos"] = torch.empty((M, 3), dtype=f32, device=dev)
            d["samples_pos_4d"] = torch.empty((M, 4), dtype=f32, device=dev)
            d["samples_dirs"] = torch.empty((M, 3), dtype=f32, device=dev)
            d["samples_z"] = torch.empty((M, 1), dtype=f32, device=dev)
            d["samples_dt"] = torch.empty((M, 1), dtype=f32, device=dev)
            d["samples_sdf"] = torch.empty((M, 1), dtype=f32, device=dev)
            d["ray_fixed_dt"] = torch.empty((R, 1), dtype=f32, device=dev)
            d["ray_start_end_idx"] = torch.empty((R, 2), dtype=torch.int32, device=dev)

    def __setattr__(self, name, value):
        """Python may overwrite any attribute (the reference does: sdf_utils.py:216 assigns samples_pos).  A container whose
        ray ranges or sample count were re-assigned from outside is no longer known to be densely packed: the `_exact`
        shortcut of compact_to_valid_samples / compute_exact_nr_samples is dropped and the generic path recounts.
        Producers set `_exact = True` AFTER filling the container."""
        if name in ("ray_start_end_idx", "cur_nr_samples"):
            object.__setattr__(self, "_exact", False)
            object.__setattr__(self, "_dense", False)
            for k in ("_ray_counts", "_host_nonempty", "_known_total"):
                object.__setattr__(self, k, None)
        object.__setattr__(self, name, value)

    # -- ray-index arguments shared by every per-ray kernel
    def _ri(self):
        se = self.ray_start_end_idx
        if se.dtype != torch.int32 or not se.is_contiguous():
            se = se.to(torch.int32).contiguous()
            object.__setattr__(self, "ray_start_end_idx", se)   # same values: packing knowledge is unchanged
        return (L.c_i(se.shape[0]), L.ptr(se), L.c_i(int(self.rays_have_equal_nr_of_samples)),
                L.c_i(int(self.fixed_nr_of_samples_per_ray)), L.c_i(int(self.max_nr_samples)))

    def compute_exact_nr_samples(self):
        """Host sync (the reference's only one: src/RaySamplesPacked.cu:44-54)."""
        if self._exact:
            return int(self.cur_nr_samples.item())
        se = self.ray_start_end_idx
        return int((se[:, 1] - se[:, 0]).sum().item())

    def compact_to_valid_samples(self, known_nr_samples=None):
        """known_nr_samples (not part of the reference's API): the exact sample count when the caller already has it on the
        host -- the container of a producer that packs densely is then narrowed WITHOUT the host sync that reading
        `cur_nr_samples` costs (train_step.Trainer._samples derives the counts after the importance rounds from the march's)"""
        R = self.m_nr_rays
        out = RaySamplesPacked(R, 0, device=self.samples_pos.device, _alloc=False)
        out.has_sdf = self.has_sdf
        out.rays_have_equal_nr_of_samples = self.rays_have_equal_nr_of_samples
        out.fixed_nr_of_samples_per_ray = self.fixed_nr_of_samples_per_ray
        if self._exact:
            # producer already packed the samples densely in ray order: the compaction is a narrow view
            nonempty = self._host_nonempty
            if known_nr_samples is not None:
                cur = int(known_nr_samples)
            elif self._known_total is not None:       # a merge whose total followed from counts that were already on the host
                cur = int(self._known_total)
            elif self._ray_counts is not None and R <= _COUNTS_ON_HOST_MAX_RAYS:
                # a march: its per-ray counts instead of the total -- the same one sync, and the number of non-empty rays comes
                # with it (see combine_uniform_samples_with_imp).  Training-sized batches only: a whole image's counts are a
                # 1 MB pageable copy, and a torch CPU reduction over them wakes the intra-op thread pool (measured: +12 ms per
                # 512x512 image in tools/cfg3_render.py on the GPU box) -- numpy, single thread, 16 KB at 4096 rays
                c = self._ray_counts.cpu().numpy()
                cur = int(c.sum(dtype=np.int64))
                nonempty = int(np.count_nonzero(c))
            else:
                cur = int(self.cur_nr_samples.item())      # the one host sync of this call
            n = min(cur, self.max_nr_samples)
            out.max_nr_samples = n
            for name in ("samples_pos", "samples_pos_4d", "samples_dirs", "samples_z", "samples_dt", "samples_sdf"):
                setattr(out, name, getattr(self, name)[:n])
            out.ray_fixed_dt = self.ray_fixed_dt
            out.ray_start_end_idx = self.ray_start_end_idx
            if n < cur:  # pool overflow: drop the rays that did not fit
                se = self.ray_start_end_idx
                bad = se[:, 1] > n
                out.ray_start_end_idx = torch.where(bad[:, None], torch.zeros_like(se), se)
                out.ray_fixed_dt = torch.where(bad[:, None], torch.zeros_like(self.ray_fixed_dt), self.ray_fixed_dt)
            # (the producer's counter IS the count unless the pool overflowed: no new tensor, no fill launch)
            out.cur_nr_samples = (self.cur_nr_samples if n == cur else
                                  torch.full((1,), n, dtype=torch.int32, device=self.samples_pos.device))
            out._exact = True
            out._dense = cur <= self.max_nr_samples     # (an overflowing pool leaves slots of dropped rays behind)
            if n == cur:                                # (after the assignments above: they reset the host-side knowledge)
                object.__setattr__(out, "_host_nonempty", nonempty)
            return out
        dev = self.samples_pos.device
        se = self.ray_start_end_idx.to(torch.int32).contiguous()
        scratch = torch.empty(2 * R, dtype=torch.int32, device=dev)
        total = L.zeroed_int(dev)
        L.call("psdf_compact_offsets", L.c_i(R), L.ptr(se), L.ptr(scratch), L.ptr(total), L.stream())
        n = int(total.item())
        res = RaySamplesPacked(R, n, device=dev)
        res.has_sdf, res.rays_have_equal_nr_of_samples = self.has_sdf, self.rays_have_equal_nr_of_samples
        res.fixed_nr_of_samples_per_ray = self.fixed_nr_of_samples_per_ray
        src = [_f32c(getattr(self, k).reshape(-1, c)) for k, c in (("samples_pos", 3), ("samples_pos_4d", 4),
                                                                     ("samples_dirs", 3), ("samples_z", 1),
                                                                     ("samples_dt", 1), ("samples_sdf", 1))]
        L.call("psdf_compact_copy", L.c_i(R), L.ptr(se), L.ptr(scratch[R:]), *[L.ptr(t) for t in src],
               L.ptr(_f32c(self.ray_fixed_dt)), L.ptr(res.samples_pos), L.ptr(res.samples_pos_4d), L.ptr(res.samples_dirs),
               L.ptr(res.samples_z), L.ptr(res.samples_dt), L.ptr(res.samples_sdf), L.ptr(res.ray_fixed_dt),
               L.ptr(res.ray_start_end_idx), L.stream())
        res.cur_nr_samples = total
        res._exact = True
        res._dense = True
        return res

    def initialize_with_one_sample_per_ray(self, one_sample_per_ray, dirs):
        """Sphere-tracing helper (src/RaySamplesPacked.cu:97-122); int32 ranges on the samples' device and the
        members (not shadowing locals) are updated -- SURVEY.md App. B3."""
        n = one_sample_per_ray.shape[0]
        dev = one_sample_per_ray.device
        self.samples_pos = one_sample_per_ray
        self.samples_dirs = dirs
        self.samples_z = torch.zeros((n, 1), dtype=torch.float32, device=dev)
        self.samples_dt = torch.zeros((n, 1), dtype=torch.float32, device=dev)
        self.ray_fixed_dt = torch.zeros((n, 1), dtype=torch.float32, device=dev)
        start = torch.arange(n, dtype=torch.int32, device=dev).view(-1, 1)
        self.ray_start_end_idx = torch.cat([start, start + 1], 1).contiguous()
        self.max_nr_samples = n
        self.m_nr_rays = n
        self.cur_nr_samples = torch.full((1,), n, dtype=torch.int32, device=dev)
        self.rays_have_equal_nr_of_samples = True
        self.fixed_nr_of_samples_per_ray = 1
        self.has_sdf = False
        self._exact = True

    def set_sdf(self, sdf):
        self.samples_sdf = sdf.view(-1, 1)
        self.has_sdf = True

    def remove_sdf(self):
        self.has_sdf = False

    @staticmethod
    def compute_per_sample_ray_idx(ray_start_end_idx, nr_samples):
        se = ray_start_end_idx.to(torch.int32).contiguous()
        out = torch.empty(int(nr_samples), dtype=torch.int32, device=se.device)
        L.call("psdf_per_sample_ray_idx", L.c_i(se.shape[0]), L.c_i(int(nr_samples)), L.ptr(se), L.ptr(out), L.stream())
        return out


# ---------------------------------------------------------------------------------------------- sphere
class Sphere:
    def __init__(self, radius, center):
        self.m_radius = float(radius)
        self.m_center = _vec3(center)
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.m_center_tensor = torch.tensor(self.m_center, dtype=torch.float32, device=dev)
        self.m_center_tensor._psdf_host3 = (self.m_center, self.m_center_tensor._version)

    def ray_intersection(self, ray_origins, ray_dirs):
        _check2d(ray_origins, 3, "ray_origins")
        _check2d(ray_dirs, 3, "ray_dirs")
        L.require_cuda(ray_origins, ray_dirs)
        o, d = _f32c(ray_origins), _f32c(ray_dirs)
        n, dev = o.shape[0], o.device
        f = dict(dtype=torch.float32, device=dev)
        p0, t0 = torch.empty((n, 3), **f), torch.empty((n, 1), **f)
        p1, t1 = torch.empty((n, 3), **f), torch.empty((n, 1), **f)
        hit = torch.empty((n, 1), dtype=torch.bool, device=dev)
        L.call("psdf_sphere_ray_intersection", L.c_i(n), L.c_f(self.m_radius), _host3(_vec3(self.m_center)), L.ptr(o),
               L.ptr(d), L.ptr(p0), L.ptr(t0), L.ptr(p1), L.ptr(t1), L.ptr(hit), L.stream())
        return p0, t0, p1, t1, hit

    def rand_points_inside(self, nr_points):
        dev = self.m_center_tensor.device
        L.require_cuda(self.m_center_tensor)
        n = int(nr_points)
        phi = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 2 * math.pi)
        costheta = torch.empty(n, dtype=torch.float32, device=dev).uniform_(-1, 1)
        u = torch.rand(n, dtype=torch.float32, device=dev)
        pts = torch.empty((n, 3), dtype=torch.float32, device=dev)
        L.call("psdf_sphere_rand_points_inside", L.c_i(n), L.c_f(self.m_radius), L.ptr(phi), L.ptr(costheta), L.ptr(u),
               L.ptr(pts), L.stream())
        return pts

    def check_point_inside_primitive(self, points):
        c = self.m_center_tensor.to(points.device).view(1, 3)
        return (points - c).norm(2, 1, True) < self.m_radius


# ---------------------------------------------------------------------------------------------- occupancy grid
class OccupancyGrid:
    _rng = Pcg32()  # process-global, like the reference's static member (include/permuto_sdf/OccupancyGrid.cuh:58)
    POOL = 1024 * 1024 * 2  # the reference's fixed sample pool (src/OccupancyGrid.cu:216)
    use_coarse_mask = True  # DDA probes of empty 8x8x8 blocks answered from an LDS bit mask (same results; _coarse())

    def __init__(self, nr_voxels_per_dim, grid_extent, grid_translation, device=None):
        self.m_nr_voxels_per_dim = int(nr_voxels_per_dim)
        self.m_grid_extent = float(grid_extent)
        self.m_grid_translation = _vec3(grid_translation)
        self._dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.m_grid_translation_tensor = torch.tensor(self.m_grid_translation, dtype=torch.float32, device=self._dev)
        self.m_grid_values = self.make_grid_values(self.m_nr_voxels_per_dim, self._dev)
        self.m_grid_occupancy = self.make_grid_occupancy(self.m_nr_voxels_per_dim, self._dev)
        self.max_nr_samples = OccupancyGrid.POOL

    @staticmethod
    def _check_dim(n):
        if n % 2 != 0 or (n & (n - 1)) != 0 or n > 1024:
            raise ValueError("nr_voxels_per_dim must be a power of two <= 1024 (Morton codes), got %d" % n)

    @staticmethod
    def make_grid_values(nr_voxels_per_dim, device=None):
        OccupancyGrid._check_dim(nr_voxels_per_dim)
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        return torch.ones(nr_voxels_per_dim ** 3, dtype=torch.float32, device=dev)

    @staticmethod
    def make_grid_occupancy(nr_voxels_per_dim, device=None):
        OccupancyGrid._check_dim(nr_voxels_per_dim)
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        return torch.ones(nr_voxels_per_dim ** 3, dtype=torch.bool, device=dev)

    def set_grid_values(self, grid_values):
        self.m_grid_values = grid_values

    def set_grid_occupancy(self, grid_occupancy):
        self.m_grid_occupancy = grid_occupancy

    def get_grid_values(self):
        return self.m_grid_values

    def get_grid_occupancy(self):
        return self.m_grid_occupancy

    def get_nr_voxels(self):
        return self.m_nr_voxels_per_dim ** 3

    def get_nr_voxels_per_dim(self):
        return self.m_nr_voxels_per_dim

    def _grid_args(self):
        return L.c_i(self.m_nr_voxels_per_dim), L.c_f(self.m_grid_extent), _host3(self.m_grid_translation)

    def _occ(self):
        occ = self.m_grid_occupancy
        if occ.dtype != torch.bool or not occ.is_contiguous():
            raise ValueError("grid occupancy must be a contiguous bool tensor")
        L.require_cuda(occ)
        return occ

    COARSE_MIN_RAYS = 4096

    def _coarse(self, count):
        """Coarse mask of the CURRENT occupancy for the DDA kernels (csrc/sampling.hip, struct Occ): one bit per 8x8x8
        block, rebuilt on every use because Python may have written into the occupancy tensor since (one 16-MiB read).
        Measured (tools/dda_march_bench.py, 256^3 shell grid): the march of 16 384 rays 707 -> 600 us, of 262 144 rays
        1.29 -> 1.18 ms; for the ~700 rays of a training step the march is a latency chain per ray and the extra LDS
        hop makes it 8 % slower, so small batches (and grids without a mask) get None."""
        words = L.lib().psdf_occupancy_coarse_words(self.m_nr_voxels_per_dim)
        if words == 0 or not OccupancyGrid.use_coarse_mask or count < OccupancyGrid.COARSE_MIN_RAYS:
            return None
        buf = getattr(self, "_coarse_buf", None)
        occ = self._occ()
        if buf is None or buf.numel() != words or buf.device != occ.device:
            buf = self._coarse_buf = torch.empty(words, dtype=torch.int32, device=occ.device)
        L.call("psdf_occupancy_coarse_mask", L.c_i(self.m_nr_voxels_per_dim), L.ptr(occ), L.ptr(buf), L.stream())
        return buf

    def compute_grid_points(self, randomize_position):
        n = self.get_nr_voxels()
        pts = torch.empty((n, 3), dtype=torch.float32, device=self._dev)
        rng = OccupancyGrid._rng
        L.call("psdf_grid_points", L.c_i(n), *self._grid_args(), None, *rng.args(), L.c_i(int(randomize_position)),
               L.ptr(pts), L.stream())
        if randomize_position:
            rng.advance()
        return pts

    def compute_random_sample_of_grid_points(self, nr_voxels_to_select, randomize_position):
        k = int(nr_voxels_to_select)
        idx = torch.randint(0, self.get_nr_voxels(), (k,), dtype=torch.int32, device=self._dev)
        pts = torch.empty((k, 3), dtype=torch.float32, device=self._dev)
        rng = OccupancyGrid._rng
        L.call("psdf_grid_points", L.c_i(k), *self._grid_args(), L.ptr(idx), *rng.args(), L.c_i(int(randomize_position)),
               L.ptr(pts), L.stream())
        if randomize_position:
            rng.advance()
        return pts, idx

    def check_occupancy(self, points):
        _check2d(points, 3, "points")
        if points.dtype != torch.float32:
            raise ValueError("positions should be of type float")
        p = _f32c(points)
        out = torch.empty((p.shape[0], 1), dtype=torch.bool, device=p.device)
        L.call("psdf_grid_check_occupancy", L.c_i(p.shape[0]), *self._grid_args(), L.ptr(self._occ()), L.ptr(p), L.ptr(out),
               L.stream())
        return out

    def update_with_density(self, density, decay, occupancy_tresh):
        _check2d(density, None, "density")
        if not decay < 1.0:
            raise ValueError("We except the decay to be <1.0 but it is %s" % decay)
        L.call("psdf_grid_update_with_density", L.c_i(self.get_nr_voxels()), None, L.ptr(_f32c(density)), L.c_f(decay),
               L.c_f(occupancy_tresh), L.ptr(self.m_grid_values), L.ptr(self._occ()), L.stream())

    def update_with_density_random_sample(self, point_indices, density, decay, occupancy_tresh):
        _check2d(density, None, "density")
        if point_indices.dim() != 1:
            raise ValueError("point_indices should have dim 1")
        if not decay < 1.0:
            raise ValueError("We except the decay to be <1.0 but it is %s" % decay)
        idx = point_indices.to(torch.int32).contiguous()
        L.call("psdf_grid_update_with_density", L.c_i(idx.shape[0]), L.ptr(idx), L.ptr(_f32c(density)), L.c_f(decay),
               L.c_f(occupancy_tresh), L.ptr(self.m_grid_values), L.ptr(self._occ()), L.stream())

    def update_with_sdf(self, sdf, inv_s, max_eikonal_abs, occupancy_thresh):
        _check2d(sdf, None, "sdf")
        L.call("psdf_grid_update_with_sdf", L.c_i(self.get_nr_voxels()), None, L.ptr(_f32c(sdf)), *self._grid_args(),
               L.c_f(float(inv_s)), None, L.c_i(1), L.c_f(occupancy_thresh), L.ptr(self.m_grid_values), L.ptr(self._occ()),
               L.stream())

    def update_with_sdf_random_sample(self, point_indices, sdf, inv_s, occupancy_thresh):
        _check2d(sdf, None, "sdf")
        if point_indices.dim() != 1:
            raise ValueError("point_indices should have dim 1")
        if inv_s.dim() != 1 or inv_s.shape[0] != 1:
            raise ValueError("Inv_s should be a tensor of 1 but it has sizes: %s" % (tuple(inv_s.shape),))
        idx = point_indices.to(torch.int32).contiguous()
        L.call("psdf_grid_update_with_sdf", L.c_i(idx.shape[0]), L.ptr(idx), L.ptr(_f32c(sdf)), *self._grid_args(),
               L.c_f(0.0), L.ptr(_f32c(inv_s.detach())), L.c_i(0), L.c_f(occupancy_thresh), L.ptr(self.m_grid_values),
               L.ptr(self._occ()), L.stream())

    def compute_samples_in_occupied_regions(self, ray_origins, ray_dirs, ray_t_entry, ray_t_exit,
                                            min_dist_between_samples, max_nr_samples_per_ray, jitter_samples):
        o, d = _f32c(ray_origins), _f32c(ray_dirs)
        te, tx = _f32c(ray_t_entry), _f32c(ray_t_exit)
        L.require_cuda(o, d, te, tx)
        R = o.shape[0]
        rs = RaySamplesPacked(R, self.max_nr_samples, device=o.device)
        scratch = torch.empty(max(R, 1) * (3 + int(max_nr_samples_per_ray)), dtype=torch.int32, device=o.device)
        rng = OccupancyGrid._rng
        L.call("psdf_march_samples", L.c_i(1), L.c_i(R), *self._grid_args(), L.ptr(self._occ()), L.ptr(o), L.ptr(d),
               L.ptr(te), L.ptr(tx), L.c_f(min_dist_between_samples), L.c_i(int(max_nr_samples_per_ray)),
               L.c_i(rs.max_nr_samples), *rng.args(), L.c_i(int(jitter_samples)), L.ptr(rs.samples_pos),
               L.ptr(rs.samples_dirs), L.ptr(rs.samples_z), L.ptr(rs.samples_dt), L.ptr(rs.ray_fixed_dt),
               L.ptr(rs.ray_start_end_idx), L.ptr(rs.cur_nr_samples), L.ptr(scratch), L.ptr(self._coarse(R)), L.stream())
        if jitter_samples:
            rng.advance()
        rs._exact = True
        object.__setattr__(rs, "_ray_counts", scratch[:R])     # per-ray sample counts (0 or >= 3), still on the device
        return rs

    def compute_first_sample_start_of_occupied_regions(self, ray_origins, ray_dirs, ray_t_entry, ray_t_exit):
        o, d = _f32c(ray_origins), _f32c(ray_dirs)
        te, tx = _f32c(ray_t_entry), _f32c(ray_t_exit)
        L.require_cuda(o, d, te, tx)
        R = o.shape[0]
        rs = RaySamplesPacked(R, max(self.max_nr_samples, R), device=o.device)
        scratch = torch.empty(2 * max(R, 1), dtype=torch.int32, device=o.device)
        L.call("psdf_first_hit_samples", L.c_i(R), *self._grid_args(), L.ptr(self._occ()), L.ptr(o), L.ptr(d), L.ptr(te),
               L.ptr(tx), L.c_i(rs.max_nr_samples), L.ptr(rs.samples_pos), L.ptr(rs.samples_dirs), L.ptr(rs.samples_z),
               L.ptr(rs.samples_dt), L.ptr(rs.ray_fixed_dt), L.ptr(rs.ray_start_end_idx), L.ptr(rs.cur_nr_samples),
               L.ptr(scratch), L.ptr(self._coarse(R)), L.stream())
        rs._exact = True
        return rs

    def advance_sample_to_next_occupied_voxel(self, samples_dirs, samples_pos):
        """Returns (new_samples_pos, is_within_bounds); like the reference the positions are updated IN PLACE when
        `samples_pos` is a contiguous fp32 tensor (src/OccupancyGrid.cu:311)."""
        d = _f32c(samples_dirs)
        p = samples_pos if (samples_pos.dtype == torch.float32 and samples_pos.is_contiguous()) else _f32c(samples_pos)
        L.require_cuda(p, d)
        n = p.shape[0]
        within = torch.ones((n, 1), dtype=torch.bool, device=p.device)
        L.call("psdf_advance_to_next_occupied_voxel", L.c_i(n), *self._grid_args(), L.ptr(self._occ()), L.ptr(d), L.ptr(p),
               L.ptr(within), L.ptr(self._coarse(n)), L.stream())
        return p, within

    def create_cubes_for_occupied_voxels(self):
        raise NotImplementedError("viewer-only helper (EasyPBR mesh); out of scope of the hot path")


# ---------------------------------------------------------------------------------------------- samplers
class RaySampler:
    _rng = Pcg32()

    @staticmethod
    def compute_samples_bg(ray_origins, ray_dirs, ray_t_exit, nr_samples_per_ray, sphere_radius, sphere_center,
                           randomize_position, contract_3d_samples):
        _check2d(ray_origins, 3, "ray_origins")
        _check2d(ray_dirs, 3, "ray_dirs")
        _check2d(ray_t_exit, 1, "ray_t_exit")
        o, d, tx = _f32c(ray_origins), _f32c(ray_dirs), _f32c(ray_t_exit)
        L.require_cuda(o, d, tx)
        R, n = o.shape[0], int(nr_samples_per_ray)
        rs = RaySamplesPacked(R, R * n, device=o.device)
        rs.rays_have_equal_nr_of_samples = True
        rs.fixed_nr_of_samples_per_ray = n
        rng = RaySampler._rng
        L.call("psdf_samples_bg", L.c_i(R), L.c_i(n), L.ptr(o), L.ptr(d), L.ptr(tx), L.c_f(float(sphere_radius)),
               _host3(_vec3(sphere_center)), *rng.args(), L.c_i(int(randomize_position)), L.c_i(int(contract_3d_samples)),
               L.ptr(rs.samples_pos), L.ptr(rs.samples_pos_4d), L.ptr(rs.samples_dirs), L.ptr(rs.samples_z),
               L.ptr(rs.samples_dt), L.ptr(rs.ray_fixed_dt), L.ptr(rs.ray_start_end_idx), L.stream())
        if randomize_position:
            rng.advance()
        rs.cur_nr_samples.fill_(R * n)
        rs._exact = True
        return rs

    @staticmethod
    def compute_samples_fg(ray_origins, ray_dirs, ray_t_entry, ray_t_exit, min_dist_between_samples,
                           max_nr_samples_per_ray, sphere_radius, sphere_center, randomize_position):
        _check2d(ray_origins, 3, "ray_origins")
        _check2d(ray_dirs, 3, "ray_dirs")
        _check2d(ray_t_entry, 1, "ray_t_entry")
        _check2d(ray_t_exit, 1, "ray_t_exit")
        o, d = _f32c(ray_origins), _f32c(ray_dirs)
        te, tx = _f32c(ray_t_entry), _f32c(ray_t_exit)
        L.require_cuda(o, d, te, tx)
        R = o.shape[0]
        rs = RaySamplesPacked(R, R * int(max_nr_samples_per_ray), device=o.device)
        scratch = torch.empty(max(R, 1) * (3 + int(max_nr_samples_per_ray)), dtype=torch.int32, device=o.device)
        rng = RaySampler._rng
        L.call("psdf_march_samples", L.c_i(0), L.c_i(R), L.c_i(1), L.c_f(1.0), _host3([0.0, 0.0, 0.0]), None, L.ptr(o),
               L.ptr(d), L.ptr(te), L.ptr(tx), L.c_f(min_dist_between_samples), L.c_i(int(max_nr_samples_per_ray)),
               L.c_i(rs.max_nr_samples), *rng.args(), L.c_i(int(randomize_position)), L.ptr(rs.samples_pos),
               L.ptr(rs.samples_dirs), L.ptr(rs.samples_z), L.ptr(rs.samples_dt), L.ptr(rs.ray_fixed_dt),
               L.ptr(rs.ray_start_end_idx), L.ptr(rs.cur_nr_samples), L.ptr(scratch), None, L.stream())
        if randomize_position:
            rng.advance()
        rs._exact = True
        return rs


def _per_sample(rs, shape, dev):
    """Output tensor of a per-sample kernel: the kernels write every sample of every valid ray and nothing else, so a
    container whose every slot belongs to a ray (`_dense`, set by compaction) needs no zero fill; pools with free slots
    get the zeros the reference's torch::zeros gives them."""
    dense = getattr(rs, "_dense", False) or (
        rs.rays_have_equal_nr_of_samples and
        rs.fixed_nr_of_samples_per_ray * rs.ray_start_end_idx.shape[0] == shape[0] and
        rs.max_nr_samples >= shape[0])   # equal-count mode: the implicit ranges tile the pool
    if dense:
        return torch.empty(shape, dtype=torch.float32, device=dev)
    return torch.zeros(shape, dtype=torch.float32, device=dev)


# ---------------------------------------------------------------------------------------------- compositing
class VolumeRendering:
    _rng = Pcg32()
    # the reference's integrate_with_weights_backward reads the G channel where B is meant
    # (VolumeRenderingGPU.cuh:1247).  True reproduces the reference's training gradients.
    reference_compat = True

    @staticmethod
    def _vals(t, n, c=None, name="tensor"):
        if t.dim() != 2 or t.shape[0] != n or (c is not None and t.shape[1] != c):
            raise ValueError("%s should have shape [%d, %s], got %s" % (name, n, c, tuple(t.shape)))
        return _f32c(t)

    @staticmethod
    def volume_render_nerf(ray_samples_packed, rgb_samples, radiance_samples, ray_t_exit, use_ray_t_exit):
        rs = ray_samples_packed
        _check2d(rgb_samples, 3, "rgb_samples")
        _check2d(radiance_samples, 1, "radiance_samples")
        R, M, dev = rs.ray_start_end_idx.shape[0], rs.samples_z.shape[0], rgb_samples.device
        f = dict(dtype=torch.float32, device=dev)
        pred_rgb, pred_depth = torch.zeros((R, 3), **f), torch.zeros((R, 1), **f)
        bg, w = torch.zeros((R, 1), **f), torch.zeros((M, 1), **f)
        L.call("psdf_volume_render_nerf", *rs._ri(), L.ptr(_f32c(rgb_samples)), L.ptr(_f32c(radiance_samples)),
               L.ptr(_f32c(rs.samples_z)), L.ptr(_f32c(rs.samples_dt)), L.ptr(pred_rgb), L.ptr(pred_depth), L.ptr(bg),
               L.ptr(w), L.stream())
        return pred_rgb, pred_depth, bg, w

    @staticmethod
    def volume_render_nerf_backward(grad_pred_rgb, grad_bg_transmittance, grad_weight_per_sample, pred_rgb,
                                    ray_samples_packed, rgb_samples, radiance_samples, ray_t_exit, use_ray_t_exit,
                                    bg_transmittance):
        rs = ray_samples_packed
        M, dev = rgb_samples.shape[0], rgb_samples.device
        g_rgb = torch.zeros((M, 3), dtype=torch.float32, device=dev)
        g_sig = torch.zeros((M, 1), dtype=torch.float32, device=dev)
        L.call("psdf_volume_render_nerf_backward", *rs._ri(), L.ptr(_f32c(grad_pred_rgb)),
               L.ptr(_f32c(grad_bg_transmittance)), L.ptr(_f32c(pred_rgb)), L.ptr(_f32c(bg_transmittance)),
               L.ptr(_f32c(rgb_samples)), L.ptr(_f32c(radiance_samples)), L.ptr(_f32c(rs.samples_dt)), L.ptr(g_rgb),
               L.ptr(g_sig), L.stream())
        return g_rgb, g_sig

    @staticmethod
    def compute_dt(ray_samples_packed, ray_t_exit, use_ray_t_exit):
        rs = ray_samples_packed
        M = rs.samples_z.shape[0]
        dt = torch.zeros((M, 1), dtype=torch.float32, device=rs.samples_z.device)
        L.call("psdf_compute_dt", *rs._ri(), L.ptr(_f32c(rs.samples_z)), L.ptr(_f32c(ray_t_exit)),
               L.c_i(int(use_ray_t_exit)), L.ptr(dt), L.stream())
        return dt

    @staticmethod
    def cumprod_alpha2transmittance(ray_samples_packed, alpha_samples):
        rs = ray_samples_packed
        R, M, dev = rs.ray_start_end_idx.shape[0], rs.samples_z.shape[0], alpha_samples.device
        a = VolumeRendering._vals(alpha_samples, M, 1, "alpha_samples")
        T = _per_sample(rs, (M, 1), dev)
        bg = torch.ones((R, 1), dtype=torch.float32, device=dev)
        L.call("psdf_cumprod_alpha2transmittance", *rs._ri(), L.ptr(a), L.ptr(T), L.ptr(bg), L.stream())
        return T, bg

    @staticmethod
    def integrate_with_weights(ray_samples_packed, rgb_samples, weights_samples):
        rs = ray_samples_packed
        R, dev = rs.ray_start_end_idx.shape[0], rgb_samples.device
        _check2d(rgb_samples, 3, "rgb_samples")
        pred = torch.zeros((R, 3), dtype=torch.float32, device=dev)
        L.call("psdf_integrate_with_weights", *rs._ri(), L.ptr(_f32c(rgb_samples)), L.ptr(_f32c(weights_samples)),
               L.ptr(pred), L.stream())
        return pred

    @staticmethod
    def sdf2alpha(ray_samples_packed, sdf_samples, inv_s, dynamic_inv_s, inv_s_multiplier):
        rs = ray_samples_packed
        M = rs.samples_z.shape[0]
        alpha = torch.zeros((M, 1), dtype=torch.float32, device=sdf_samples.device)   # the last sample of a ray keeps alpha 0
        L.call("psdf_sdf2alpha", *rs._ri(), L.ptr(_f32c(rs.ray_fixed_dt)), L.ptr(_f32c(rs.samples_dt)),
               L.ptr(_f32c(sdf_samples)), L.c_f(float(inv_s)), L.c_i(int(dynamic_inv_s)), L.c_f(float(inv_s_multiplier)),
               L.ptr(alpha), L.stream())
        return alpha

    @staticmethod
    def sdf_importance_cdf(ray_samples_packed, sdf_samples, inv_s, dynamic_inv_s, inv_s_multiplier):
        """NOT part of the reference's API: the cdf `importance_sampling_sdf_model` (sdf_utils.py:383-423) draws from, in one launch
        instead of the nine of
            alpha = sdf2alpha(rs, sdf, inv_s, dynamic, mult).clip(0, 1); T, _ = cumprod_alpha2transmittance(rs, 1 - alpha + 1e-7)
            w = alpha * T; _, s = sum_over_each_ray(rs, w); cdf = compute_cdf(rs, w / clamp(s, min=1e-6))
        bit-identical to that chain (for trainers that own their sampling loop; the reference's Python calls the operators)."""
        rs = ray_samples_packed
        M = rs.samples_z.shape[0]
        sdf = VolumeRendering._vals(sdf_samples, M, 1, "sdf_samples")
        cdf = _per_sample(rs, (M, 1), sdf.device)
        L.call("psdf_sdf_importance_cdf", *rs._ri(), L.ptr(_f32c(rs.ray_fixed_dt)), L.ptr(_f32c(rs.samples_dt)), L.ptr(sdf),
               L.c_f(float(inv_s)), L.c_i(int(dynamic_inv_s)), L.c_f(float(inv_s_multiplier)), L.ptr(cdf), L.stream())
        return cdf

    @staticmethod
    def sum_over_each_ray(ray_samples_packed, sample_values):
        rs = ray_samples_packed
        R, M = rs.ray_start_end_idx.shape[0], rs.samples_z.shape[0]
        v = VolumeRendering._vals(sample_values, M, None, "sample_values")
        C = v.shape[1]
        if not (C <= 3 or C == 32):
            raise ValueError("sample_values should have 1, 2, 3 or 32 channels, got %d" % C)
        s_ray = torch.zeros((R, C), dtype=torch.float32, device=v.device)
        s_smp = _per_sample(rs, (M, C), v.device)
        L.call("psdf_sum_over_each_ray", *rs._ri(), L.c_i(C), L.ptr(v), L.ptr(s_ray), L.ptr(s_smp), L.stream())
        return s_ray, s_smp

    @staticmethod
    def cumsum_over_each_ray(ray_samples_packed, sample_values, inverse):
        rs = ray_samples_packed
        M = rs.samples_z.shape[0]
        v = VolumeRendering._vals(sample_values, M, 1, "sample_values")
        out = _per_sample(rs, (M, 1), v.device)
        L.call("psdf_cumsum_over_each_ray", *rs._ri(), L.ptr(v), L.c_i(int(inverse)), L.c_i(0), L.ptr(out), L.stream())
        return out

    @staticmethod
    def compute_cdf(ray_samples_packed, sample_weights):
        rs = ray_samples_packed
        M = rs.samples_z.shape[0]
        w = VolumeRendering._vals(sample_weights, M, 1, "sample_weights")
        out = _per_sample(rs, (M, 1), w.device)
        L.call("psdf_cumsum_over_each_ray", *rs._ri(), L.ptr(w), L.c_i(0), L.c_i(1), L.ptr(out), L.stream())
        return out

    @staticmethod
    def importance_sample(ray_origins, ray_dirs, ray_samples_packed, sample_cdf, nr_importance_samples, jitter_samples):
        rs = ray_samples_packed
        R, M = rs.ray_start_end_idx.shape[0], rs.samples_z.shape[0]
        cdf = VolumeRendering._vals(sample_cdf, M, 1, "sample_cdf")
        n = int(nr_importance_samples)
        imp = RaySamplesPacked(R, R * n, device=cdf.device)
        imp.rays_have_equal_nr_of_samples = True
        imp.fixed_nr_of_samples_per_ray = n
        rng = VolumeRendering._rng
        L.call("psdf_importance_sample", *rs._ri(), L.ptr(_f32c(ray_origins)), L.ptr(_f32c(ray_dirs)),
               L.ptr(_f32c(rs.ray_fixed_dt)), L.ptr(_f32c(rs.samples_z)), L.ptr(cdf), L.c_i(n), *rng.args(),
               L.c_i(int(jitter_samples)), L.ptr(imp.samples_pos), L.ptr(imp.samples_dirs), L.ptr(imp.samples_z),
               L.stream())
        if jitter_samples:
            rng.advance()
        return imp

    @staticmethod
    def combine_uniform_samples_with_imp(ray_origins, ray_dirs, ray_t_exit, ray_samples_packed, ray_samples_imp):
        uni, imp = ray_samples_packed, ray_samples_imp
        if not imp.rays_have_equal_nr_of_samples:
            raise ValueError("the importance samples must have an equal nr of samples per ray")
        if uni.has_sdf != imp.has_sdf:
            raise ValueError("both sample sets must either have or not have sdf")
        R = uni.ray_start_end_idx.shape[0]
        n_uni, n_imp = uni.samples_z.shape[0], imp.max_nr_samples
        if R * imp.fixed_nr_of_samples_per_ray != n_imp:
            raise ValueError("importance sample count mismatch")
        dev = uni.samples_z.device
        out = RaySamplesPacked(R, n_uni + n_imp, device=dev)
        out.has_sdf = uni.has_sdf
        scratch = torch.empty(2 * max(R, 1), dtype=torch.int32, device=dev)
        L.call("psdf_combine_uniform_samples_with_imp", *uni._ri(), L.ptr(_f32c(ray_origins)), L.ptr(_f32c(ray_dirs)),
               L.ptr(_f32c(ray_t_exit)), L.ptr(_f32c(uni.ray_fixed_dt)), L.ptr(_f32c(uni.samples_z)),
               L.ptr(_f32c(uni.samples_sdf)), L.c_i(int(uni.has_sdf)), L.c_i(int(imp.fixed_nr_of_samples_per_ray)),
               L.ptr(_f32c(imp.samples_z)), L.ptr(_f32c(imp.samples_sdf)), L.c_i(out.max_nr_samples), L.ptr(out.samples_pos),
               L.ptr(out.samples_dirs), L.ptr(out.samples_z), L.ptr(out.samples_dt), L.ptr(out.samples_sdf),
               L.ptr(out.ray_fixed_dt), L.ptr(out.ray_start_end_idx), L.ptr(out.cur_nr_samples), L.ptr(scratch), L.stream())
        out._exact = True
        # the reference syncs again when this container is compacted (sdf_utils.py:383-423 -> src/RaySamplesPacked.cu:44-54).  When
        # the number of non-empty rays of `uni` is known on the host -- it came with the march's counts -- the total is too: a ray
        # with n <= 1 samples yields none, every other ray n + nr_imp (combine_count_kernel), and a march leaves 0 or >= 3 per ray
        if uni._host_nonempty is not None and uni._exact:
            object.__setattr__(out, "_known_total", n_uni + uni._host_nonempty * int(imp.fixed_nr_of_samples_per_ray))
            object.__setattr__(out, "_host_nonempty", uni._host_nonempty)
        return out

    # ---- backward passes
    @staticmethod
    def cumprod_alpha2transmittance_backward(grad_transmittance, grad_bg_transmittance, ray_samples_packed, alpha,
                                             transmittance, bg_transmittance, cumsumLV):
        rs = ray_samples_packed
        M = rs.samples_z.shape[0]
        if grad_transmittance.shape[0] != M:
            raise ValueError("grad_transmittance should have size nr_samples_total x 1")
        g = _per_sample(rs, (M, 1), alpha.device)
        L.call("psdf_cumprod_alpha2transmittance_backward", *rs._ri(), L.ptr(_f32c(grad_bg_transmittance)),
               L.ptr(_f32c(alpha)), L.ptr(_f32c(bg_transmittance)), L.ptr(_f32c(cumsumLV)), L.ptr(g), L.stream())
        return g

    @staticmethod
    def integrate_with_weights_backward(grad_pred_rgb, ray_samples_packed, rgb_samples, weights_samples, pred_rgb):
        rs = ray_samples_packed
        R, M = rs.ray_start_end_idx.shape[0], rs.samples_z.shape[0]
        if grad_pred_rgb.shape[0] != R or grad_pred_rgb.shape[1] != 3:
            raise ValueError("grad_pred_rgb should have size nr_rays x 3")
        dev = rgb_samples.device
        g_rgb = _per_sample(rs, (M, 3), dev)
        g_w = _per_sample(rs, (M, 1), dev)
        L.call("psdf_integrate_with_weights_backward", *rs._ri(), L.ptr(_f32c(grad_pred_rgb)), L.ptr(_f32c(rgb_samples)),
               L.ptr(_f32c(weights_samples)), L.ptr(g_rgb), L.ptr(g_w), L.c_i(int(VolumeRendering.reference_compat)),
               L.stream())
        return g_rgb, g_w

    @staticmethod
    def sum_over_each_ray_backward(grad_values_sum_per_ray, grad_values_sum_per_sample, ray_samples_packed, sample_values):
        rs = ray_samples_packed
        R, M = rs.ray_start_end_idx.shape[0], rs.samples_z.shape[0]
        C = sample_values.shape[1]
        if C > 3:
            raise ValueError("sum_over_each_ray_backward supports 1, 2 or 3 channels (reference: src/VolumeRendering.cu:621-663)")
        if grad_values_sum_per_ray.shape[0] != R or grad_values_sum_per_sample.shape[0] != M:
            raise ValueError("gradient shapes do not match the sample container")
        g = _per_sample(rs, (M, C), sample_values.device)
        L.call("psdf_sum_over_each_ray_backward", *rs._ri(), L.c_i(C), L.ptr(_f32c(grad_values_sum_per_ray)),
               L.ptr(_f32c(grad_values_sum_per_sample)), L.ptr(g), L.stream())
        return g


# ---------------------------------------------------------------------------------------------- misc statics
_SH_CHANNELS = {1: 1, 2: 4, 3: 9, 4: 16, 5: 25, 6: 36, 7: 49}


class PermutoSDF:
    @staticmethod
    def spherical_harmonics(dirs, degree):
        if dirs.dim() != 2 or dirs.shape[1] != 3:
            raise ValueError("We are assuming that dirs should be Nx3")
        if degree not in _SH_CHANNELS:
            raise ValueError("Nr channels encoded is not valid. Maybe you gave a degree number that is not supported.")
        d = _f32c(dirs)
        L.require_cuda(d)
        out = torch.empty((d.shape[0], _SH_CHANNELS[degree]), dtype=torch.float32, device=d.device)
        L.call("psdf_spherical_harmonics", L.c_i(d.shape[0]), L.c_i(int(degree)), L.ptr(d), L.ptr(out), L.stream())
        return out

    @staticmethod
    def random_rays_from_reel(reel, nr_rays):
        """reel: object with rgb_reel [I,3,H,W], mask_reel [I,1,H,W], K_reel [I,3,3], tf_world_cam_reel [I,4,4], has_mask
        (DataLoaders TensorReel, src/PermutoSDF.cu:67-112)."""
        rgb = _f32c(reel.rgb_reel)
        L.require_cuda(rgb)
        I, _, H, W = rgb.shape
        dev, n = rgb.device, int(nr_rays)
        has_mask = bool(getattr(reel, "has_mask", False))
        mask = _f32c(reel.mask_reel) if has_mask else rgb
        f = dict(dtype=torch.float32, device=dev)
        o, d = torch.empty((n, 3), **f), torch.empty((n, 3), **f)
        gt, gm = torch.empty((n, 3), **f), torch.empty((n, 1), **f)
        pix = torch.randint(0, H * W, (n,), dtype=torch.int32, device=dev)
        img = torch.randint(0, I, (n,), dtype=torch.int32, device=dev)
        L.call("psdf_random_rays_from_reel", L.c_i(n), L.c_i(I), L.c_i(H), L.c_i(W), L.ptr(rgb), L.ptr(mask),
               L.ptr(_f32c(reel.K_reel)), L.ptr(_f32c(reel.tf_world_cam_reel)), L.ptr(pix), L.ptr(img), L.c_i(int(has_mask)),
               L.ptr(o), L.ptr(d), L.ptr(gt), L.ptr(gm), L.stream())
        return o, d, gt, gm, img

    @staticmethod
    def _unsupported(name):
        raise NotImplementedError("PermutoSDF.%s is not on the training/rendering hot path (no caller in "
                                  "train_permuto_sdf.py); out of scope, see SURVEY.md section 2" % name)

    @staticmethod
    def rays_from_reprojection_reel(*a):
        PermutoSDF._unsupported("rays_from_reprojection_reel")

    @staticmethod
    def update_errors_of_matching_indices(*a):
        PermutoSDF._unsupported("update_errors_of_matching_indices")

    @staticmethod
    def meshgrid3d(*a):
        PermutoSDF._unsupported("meshgrid3d")

    @staticmethod
    def low_discrepancy2d_sampling(*a):
        PermutoSDF._unsupported("low_discrepancy2d_sampling")


class TrainParams:
    """Stand-in for the configuru-backed TrainParams (src/TrainParams.cxx:36-43): four logging switches read from
    the ``train: { ... }`` block of a config file."""

    def __init__(self):
        self.m_with_visdom = self.m_with_tensorboard = self.m_with_wandb = self.m_save_checkpoint = False

    @staticmethod
    def create(config_file):
        import re
        tp = TrainParams()
        try:
            txt = open(config_file).read()
        except OSError:
            return tp
        m = re.search(r"train\s*:\s*\{(.*?)\}", txt, re.S)
        if m:
            for key in ("with_visdom", "with_tensorboard", "with_wandb", "save_checkpoint"):
                k = re.search(key + r"\s*:\s*(true|false)", m.group(1))
                if k:
                    setattr(tp, "m_" + key, k.group(1) == "true")
        return tp

    def with_visdom(self):
        return self.m_with_visdom

    def with_tensorboard(self):
        return self.m_with_tensorboard

    def with_wandb(self):
        return self.m_with_wandb

    def save_checkpoint(self):
        return self.m_save_checkpoint

    def set_with_visdom(self, v):
        self.m_with_visdom = bool(v)

    def set_with_tensorboard(self, v):
        self.m_with_tensorboard = bool(v)

    def set_with_wandb(self, v):
        self.m_with_wandb = bool(v)

    def set_save_checkpoint(self, v):
        self.m_save_checkpoint = bool(v)
