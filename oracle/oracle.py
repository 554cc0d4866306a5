"""ctypes front-end of the CPU oracle.  TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py cpu_baseline).

Two back-ends with the same numpy call surface:
  * ``Oracle("port")``: oracle/libpsdf_oracle.so, our plain-C restatement (oracle/psdf_oracle.c), always buildable
    with gcc (``build()`` does it on demand);
  * ``Oracle("ref")``:  oracle/_ref/libpsdf_ref.so, the REFERENCE's own kernel headers compiled for the CPU
    (``make -C oracle ref``; only where /root/reference exists, the built .so travels to the GPU box);
  * ``Oracle("ref_fma")``: the same headers built with floating-point contraction on (``make -C oracle ref_fma``), the
    way nvcc builds them by default -- used only by tools/fma_contraction_gap.py to COUNT what contraction changes.
Every method takes / returns numpy arrays (float32 / int32 / bool) and mirrors one reference kernel launch
including the host-side allocation semantics of src/*.cu (zeros / ones / empty initial values).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_LIB = os.path.join(HERE, "libpsdf_oracle.so")
REF_LIB = os.path.join(HERE, "_ref", "libpsdf_ref.so")
REF_FMA_LIB = os.path.join(HERE, "_ref", "libpsdf_ref_fma.so")     # the same headers, floating-point contraction on (Makefile)
PCG_STATE = 0x853C49E6748FEA9B
PCG_INC = 0xDA3E39CB94B95BDB

c_i, c_f, c_u64, c_l = ctypes.c_int, ctypes.c_float, ctypes.c_uint64, ctypes.c_int64


def build(ref=True, quiet=True):
    """(Re)build the oracle libraries; the reference build is attempted only when /root/reference exists."""
    out = subprocess.DEVNULL if quiet else None
    subprocess.run(["make", "-C", HERE, "oracle"], check=True, stdout=out)
    if ref and os.path.isdir("/root/reference/kernels"):
        subprocess.run(["make", "-C", HERE, "ref"], check=True, stdout=out)
    return os.path.exists(PORT_LIB), os.path.exists(REF_LIB)


def have_ref():
    return os.path.exists(REF_LIB)


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be C contiguous"
    return a.ctypes.data_as(ctypes.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Samples:
    """numpy mirror of RaySamplesPacked."""

    def __init__(self, R, M):
        self.R, self.max_nr_samples = R, M
        self.pos = np.zeros((M, 3), np.float32)
        self.pos4 = np.zeros((M, 4), np.float32)
        self.dirs = np.zeros((M, 3), np.float32)
        self.z = np.zeros((M, 1), np.float32)
        self.dt = np.zeros((M, 1), np.float32)
        self.sdf = np.zeros((M, 1), np.float32)
        self.fixed_dt = np.zeros((R, 1), np.float32)
        self.start_end = np.zeros((R, 2), np.int32)
        self.cur = np.zeros(1, np.int32)
        self.equal, self.fixed, self.has_sdf = False, 0, False

    def ri(self):
        return (_p(self.start_end), c_i(int(self.equal)), c_i(int(self.fixed)), c_i(int(self.max_nr_samples)))

    def counts(self):
        return self.start_end[:, 1] - self.start_end[:, 0]

    def total(self):
        return int(self.counts().sum())


class Oracle:
    def __init__(self, kind="port"):
        self.kind = kind
        if kind == "port":
            if not os.path.exists(PORT_LIB):
                build(ref=False)
            self.lib = ctypes.CDLL(PORT_LIB)
            self.pre = "orc_"
        elif kind == "ref":
            if not have_ref():
                raise FileNotFoundError(REF_LIB)
            self.lib = ctypes.CDLL(REF_LIB)
            self.pre = "ref_"
        elif kind == "ref_fma":
            if not os.path.exists(REF_FMA_LIB):
                if os.path.isdir("/root/reference/kernels"):
                    subprocess.run(["make", "-C", HERE, "ref_fma"], check=True, stdout=subprocess.DEVNULL)
                else:
                    raise FileNotFoundError(REF_FMA_LIB)
            self.lib = ctypes.CDLL(REF_FMA_LIB)
            self.pre = "ref_"
        else:
            raise ValueError(kind)
        for n in ("morton3D", "morton3D_invert"):
            getattr(self.lib, self.pre + n).restype = ctypes.c_uint32

    def _fn(self, name):
        f = getattr(self.lib, self.pre + name)
        return f

    # ---- scalars
    def morton3D(self, x, y, z):
        return int(self._fn("morton3D")(ctypes.c_uint32(x), ctypes.c_uint32(y), ctypes.c_uint32(z)))

    def morton3D_invert(self, x):
        return int(self._fn("morton3D_invert")(ctypes.c_uint32(x)))

    def pcg32(self, n, advance=0, state=PCG_STATE, inc=PCG_INC):
        st, ic = c_u64(state), c_u64(inc)
        u, f = np.zeros(n, np.uint32), np.zeros(n, np.float32)
        # draw the uint stream and the float stream from two copies of the same generator
        self._fn("pcg32")(ctypes.byref(st), ctypes.byref(ic), c_l(advance), c_i(n), _p(u), None)
        st2, ic2 = c_u64(state), c_u64(inc)
        self._fn("pcg32")(ctypes.byref(st2), ctypes.byref(ic2), c_l(advance), c_i(n), None, _p(f))
        return u, f, st.value

    # ---- occupancy grid
    def grid_points(self, n, extent, tr, indices=None, randomize=False, rng=(PCG_STATE, PCG_INC)):
        count = n ** 3 if indices is None else len(indices)
        out = np.zeros((count, 3), np.float32)
        tr = f32(tr)
        idx = None if indices is None else i32(indices)
        if self.kind == "port":
            self._fn("grid_points")(c_i(count), c_i(n), c_f(extent), _p(tr), _p(idx), c_u64(rng[0]), c_u64(rng[1]),
                                    c_i(int(randomize)), _p(out))
        else:
            self._fn("compute_grid_points")(c_i(count), c_i(n), c_f(extent), _p(tr), _p(idx), c_u64(rng[0]), c_u64(rng[1]),
                                            c_i(int(randomize)), _p(out))
        return out

    def update_with_density(self, values, occ, density, decay, thresh, indices=None):
        values, occ = f32(values).copy(), np.ascontiguousarray(occ, dtype=np.bool_).copy()
        density = f32(density).reshape(-1, 1)
        idx = None if indices is None else i32(indices)
        count = len(density)
        if self.kind == "port":
            self._fn("update_with_density")(c_i(count), _p(idx), _p(density), c_f(decay), c_f(thresh), _p(values), _p(occ))
        else:
            self._fn("update_with_density")(c_i(count), c_i(len(values)), _p(idx), _p(density), c_i(0), c_f(decay),
                                            c_f(thresh), _p(values), _p(occ))
        return values, occ

    def update_with_sdf(self, values, occ, sdf, n, extent, inv_s, thresh, indices=None):
        values, occ = f32(values).copy(), np.ascontiguousarray(occ, dtype=np.bool_).copy()
        sdf = f32(sdf).reshape(-1, 1)
        idx = None if indices is None else i32(indices)
        inv_t = f32([inv_s])
        count = len(sdf)
        if self.kind == "port":
            self._fn("update_with_sdf")(c_i(count), _p(idx), _p(sdf), c_f(extent), c_i(n), c_f(inv_s), _p(inv_t),
                                        c_f(thresh), _p(values), _p(occ))
        else:
            self._fn("update_with_sdf")(c_i(count), c_i(len(values)), _p(idx), _p(sdf), c_f(extent), c_i(n), c_f(inv_s),
                                        _p(inv_t), c_f(thresh), _p(values), _p(occ))
        return values, occ

    def check_occupancy(self, n, extent, tr, occ, pts):
        pts, tr = f32(pts), f32(tr)
        occ = np.ascontiguousarray(occ, dtype=np.bool_)
        out = np.zeros((len(pts), 1), np.bool_)
        self._fn("check_occupancy")(c_i(len(pts)), c_i(n), c_f(extent), _p(tr), _p(occ), _p(pts), _p(out))
        return out

    def march_samples(self, o, d, t_entry, t_exit, min_dist, max_per_ray, M, grid=None, jitter=False,
                      rng=(PCG_STATE, PCG_INC)):
        """grid = (n, extent, tr, occ) -> compute_samples_in_occupied_regions; grid=None -> compute_samples_fg."""
        o, d, te, tx = f32(o), f32(d), f32(t_entry).reshape(-1, 1), f32(t_exit).reshape(-1, 1)
        R = len(o)
        s = Samples(R, M)
        if self.kind == "port":
            if grid is not None:
                n, extent, tr, occ = grid
                tr, occ = f32(tr), np.ascontiguousarray(occ, dtype=np.bool_)
            else:
                n, extent, tr, occ = 1, 1.0, f32([0, 0, 0]), None
            self._fn("march_samples")(c_i(int(grid is not None)), c_i(R), c_i(n), c_f(extent), _p(tr), _p(occ), _p(o), _p(d),
                                      _p(te), _p(tx), c_f(min_dist), c_i(max_per_ray), c_i(M), c_u64(rng[0]), c_u64(rng[1]),
                                      c_i(int(jitter)), _p(s.pos), _p(s.dirs), _p(s.z), _p(s.dt), _p(s.fixed_dt),
                                      _p(s.start_end), _p(s.cur))
        elif grid is not None:
            n, extent, tr, occ = grid
            tr, occ = f32(tr), np.ascontiguousarray(occ, dtype=np.bool_)
            self._fn("compute_samples_in_occupied_regions")(
                c_i(R), c_i(n), c_f(extent), _p(tr), _p(o), _p(d), _p(te), _p(tx), _p(occ), c_f(min_dist), c_i(max_per_ray),
                c_i(M), c_u64(rng[0]), c_u64(rng[1]), c_i(int(jitter)), _p(s.pos), _p(s.dirs), _p(s.z), _p(s.dt),
                _p(s.fixed_dt), _p(s.start_end), _p(s.cur))
        else:
            c = f32([0, 0, 0])
            self._fn("samples_fg")(c_i(R), _p(o), _p(d), _p(te), _p(tx), c_f(0.5), _p(c), c_f(min_dist), c_i(max_per_ray),
                                   c_i(M), c_u64(rng[0]), c_u64(rng[1]), c_i(int(jitter)), _p(s.pos), _p(s.dirs), _p(s.z),
                                   _p(s.dt), _p(s.fixed_dt), _p(s.start_end), _p(s.cur))
        return s

    def first_hit_samples(self, o, d, t_entry, t_exit, M, grid):
        o, d, te, tx = f32(o), f32(d), f32(t_entry).reshape(-1, 1), f32(t_exit).reshape(-1, 1)
        n, extent, tr, occ = grid
        tr, occ = f32(tr), np.ascontiguousarray(occ, dtype=np.bool_)
        R = len(o)
        s = Samples(R, M)
        if self.kind == "port":
            self._fn("first_hit_samples")(c_i(R), c_i(n), c_f(extent), _p(tr), _p(occ), _p(o), _p(d), _p(te), _p(tx), c_i(M),
                                          _p(s.pos), _p(s.dirs), _p(s.z), _p(s.dt), _p(s.fixed_dt), _p(s.start_end), _p(s.cur))
        else:
            self._fn("compute_first_sample")(c_i(R), c_i(n), c_f(extent), _p(tr), _p(o), _p(d), _p(te), _p(tx), _p(occ),
                                             c_i(M), _p(s.pos), _p(s.dirs), _p(s.z), _p(s.dt), _p(s.fixed_dt),
                                             _p(s.start_end), _p(s.cur))
        return s

    def advance_samples(self, dirs, pos, grid):
        dirs, pos = f32(dirs), f32(pos)
        n, extent, tr, occ = grid
        tr, occ = f32(tr), np.ascontiguousarray(occ, dtype=np.bool_)
        new_pos = pos.copy()
        within = np.ones((len(pos), 1), np.bool_)
        if self.kind == "port":
            self._fn("advance_samples")(c_i(len(pos)), c_i(n), c_f(extent), _p(tr), _p(occ), _p(dirs), _p(pos), _p(new_pos),
                                        _p(within))
        else:
            self._fn("advance_sample")(c_i(len(pos)), c_i(n), c_f(extent), _p(tr), _p(dirs), _p(pos), _p(occ), _p(new_pos),
                                       _p(within))
        return new_pos, within

    def compact(self, s):
        total = s.total()
        o = Samples(s.R, total)
        o.equal, o.fixed, o.has_sdf = s.equal, s.fixed, s.has_sdf
        if self.kind == "port":
            self._fn("compact")(c_i(s.R), _p(s.pos), _p(s.pos4), _p(s.dirs), _p(s.z), _p(s.dt), _p(s.sdf), _p(s.fixed_dt),
                                _p(s.start_end), _p(o.pos), _p(o.pos4), _p(o.dirs), _p(o.z), _p(o.dt), _p(o.sdf),
                                _p(o.fixed_dt), _p(o.start_end), _p(o.cur))
        else:
            self._fn("compact")(c_i(s.R), c_i(s.max_nr_samples), c_i(total), _p(s.pos), _p(s.pos4), _p(s.dirs), _p(s.z),
                                _p(s.dt), _p(s.sdf), _p(s.fixed_dt), _p(s.start_end), _p(o.pos), _p(o.pos4), _p(o.dirs),
                                _p(o.z), _p(o.dt), _p(o.sdf), _p(o.fixed_dt), _p(o.start_end), _p(o.cur))
        return o

    def per_sample_ray_idx(self, start_end, M):
        se = i32(start_end)
        out = np.full(M, -1, np.int32)
        self._fn("per_sample_ray_idx")(c_i(len(se)), c_i(M), _p(se), _p(out))
        return out

    def samples_bg(self, o, d, t_exit, per_ray, radius, center, randomize=False, contract=False, rng=(PCG_STATE, PCG_INC)):
        o, d, tx, center = f32(o), f32(d), f32(t_exit).reshape(-1, 1), f32(center)
        R = len(o)
        s = Samples(R, R * per_ray)
        s.equal, s.fixed = True, per_ray
        self._fn("samples_bg")(c_i(R), c_i(per_ray), _p(o), _p(d), _p(tx), c_f(radius), _p(center), c_u64(rng[0]),
                               c_u64(rng[1]), c_i(int(randomize)), c_i(int(contract)), _p(s.pos), _p(s.pos4), _p(s.dirs),
                               _p(s.z), _p(s.dt), _p(s.fixed_dt), _p(s.start_end))
        s.cur[0] = R * per_ray
        return s

    # ---- sphere
    def sphere_intersect(self, radius, center, o, d):
        o, d, center = f32(o), f32(d), f32(center)
        R = len(o)
        p0, t0 = np.zeros((R, 3), np.float32), np.zeros((R, 1), np.float32)
        p1, t1 = np.zeros((R, 3), np.float32), np.zeros((R, 1), np.float32)
        hit = np.zeros((R, 1), np.bool_)
        self._fn("sphere_intersect")(c_i(R), c_f(radius), _p(center), _p(o), _p(d), _p(p0), _p(t0), _p(p1), _p(t1), _p(hit))
        return p0, t0, p1, t1, hit

    def rand_points_inside(self, radius, phi, costheta, u):
        phi, costheta, u = f32(phi), f32(costheta), f32(u)
        out = np.zeros((len(phi), 3), np.float32)
        if self.kind == "port":
            self._fn("rand_points_inside")(c_i(len(phi)), c_f(radius), _p(phi), _p(costheta), _p(u), _p(out))
        else:
            c = f32([0, 0, 0])
            self._fn("rand_points_inside")(c_i(len(phi)), c_f(radius), _p(c), _p(phi), _p(costheta), _p(u), _p(out))
        return out

    def spherical_harmonics(self, dirs, degree):
        dirs = f32(dirs)
        out = np.zeros((len(dirs), degree * degree), np.float32)
        self._fn("spherical_harmonics")(c_i(len(dirs)), c_i(degree), _p(dirs), _p(out))
        return out

    def random_rays_from_reel(self, rgb, mask, K, tf, pix, img, has_mask):
        rgb, mask, K, tf = f32(rgb), f32(mask), f32(K), f32(tf)
        pix, img = i32(pix), i32(img)
        I, _, H, W = rgb.shape
        R = len(pix)
        o, d = np.zeros((R, 3), np.float32), np.zeros((R, 3), np.float32)
        gt, gm = np.zeros((R, 3), np.float32), np.zeros((R, 1), np.float32)
        if self.kind == "port":
            self._fn("random_rays_from_reel")(c_i(R), c_i(H), c_i(W), _p(rgb), _p(mask), _p(K), _p(tf), _p(pix), _p(img),
                                              c_i(int(has_mask)), _p(o), _p(d), _p(gt), _p(gm))
        else:
            self._fn("random_rays_from_reel")(c_i(R), c_i(I), c_i(H), c_i(W), _p(rgb), _p(mask), _p(K), _p(tf), _p(pix),
                                              _p(img), c_i(int(has_mask)), _p(o), _p(d), _p(gt), _p(gm))
        return o, d, gt, gm

    # ---- volume rendering; s is a Samples
    def _head(self, s):
        M = len(s.z)
        return (c_i(s.R),) + ((c_i(M),) if self.kind != "port" else ()) + s.ri()

    def volume_render_nerf(self, s, rgb, sigma):
        rgb, sigma = f32(rgb), f32(sigma).reshape(-1, 1)
        R, M = s.R, len(s.z)
        pred, depth = np.zeros((R, 3), np.float32), np.zeros((R, 1), np.float32)
        bg, w = np.zeros((R, 1), np.float32), np.zeros((M, 1), np.float32)
        self._fn("volume_render_nerf")(*self._head(s), _p(rgb), _p(sigma), _p(s.z), _p(s.dt), _p(pred), _p(depth), _p(bg), _p(w))
        return pred, depth, bg, w

    def volume_render_nerf_backward(self, s, g_pred, g_bg, pred, bg, rgb, sigma):
        g_pred, g_bg, pred, bg = f32(g_pred), f32(g_bg), f32(pred), f32(bg)
        rgb, sigma = f32(rgb), f32(sigma).reshape(-1, 1)
        M = len(s.z)
        g_rgb, g_sigma = np.zeros((M, 3), np.float32), np.zeros((M, 1), np.float32)
        if self.kind == "port":
            self._fn("volume_render_nerf_backward")(*self._head(s), _p(g_pred), _p(g_bg), _p(pred), _p(bg), _p(rgb), _p(sigma),
                                                    _p(s.dt), _p(g_rgb), _p(g_sigma))
        else:
            g_w = np.zeros((M, 1), np.float32)
            self._fn("volume_render_nerf_backward")(*self._head(s), _p(g_pred), _p(g_bg), _p(g_w), _p(pred), _p(bg), _p(rgb),
                                                    _p(sigma), _p(s.dt), _p(g_rgb), _p(g_sigma))
        return g_rgb, g_sigma

    def compute_dt(self, s, t_exit, use_t_exit):
        t_exit = f32(t_exit).reshape(-1, 1)
        dt = np.zeros((len(s.z), 1), np.float32)
        self._fn("compute_dt")(*self._head(s), _p(s.z), _p(t_exit), c_i(int(use_t_exit)), _p(dt))
        return dt

    def cumprod(self, s, alpha):
        alpha = f32(alpha).reshape(-1, 1)
        T, bg = np.zeros((len(s.z), 1), np.float32), np.ones((s.R, 1), np.float32)
        self._fn("cumprod")(*self._head(s), _p(alpha), _p(T), _p(bg))
        return T, bg

    def cumprod_backward(self, s, g_T, g_bg, alpha, T, bg, cumsumLV):
        g_T, g_bg, alpha, T, bg, cumsumLV = [f32(a) for a in (g_T, g_bg, alpha, T, bg, cumsumLV)]
        g = np.zeros((len(s.z), 1), np.float32)
        if self.kind == "port":
            self._fn("cumprod_backward")(*self._head(s), _p(g_bg), _p(alpha), _p(bg), _p(cumsumLV), _p(g))
        else:
            self._fn("cumprod_backward")(*self._head(s), _p(g_T), _p(g_bg), _p(alpha), _p(T), _p(bg), _p(cumsumLV), _p(g))
        return g

    def integrate(self, s, rgb, w):
        rgb, w = f32(rgb), f32(w).reshape(-1, 1)
        pred = np.zeros((s.R, 3), np.float32)
        self._fn("integrate")(*self._head(s), _p(rgb), _p(w), _p(pred))
        return pred

    def integrate_backward(self, s, g_pred, rgb, w, compat=True):
        g_pred, rgb, w = f32(g_pred), f32(rgb), f32(w).reshape(-1, 1)
        M = len(s.z)
        g_rgb, g_w = np.zeros((M, 3), np.float32), np.zeros((M, 1), np.float32)
        if self.kind == "port":
            self._fn("integrate_backward")(*self._head(s), _p(g_pred), _p(rgb), _p(w), _p(g_rgb), _p(g_w), c_i(int(compat)))
        else:
            pred = np.zeros((s.R, 3), np.float32)
            self._fn("integrate_backward")(*self._head(s), _p(g_pred), _p(rgb), _p(w), _p(pred), _p(g_rgb), _p(g_w))
        return g_rgb, g_w

    def sdf2alpha(self, s, sdf, inv_s, dynamic, mult):
        sdf = f32(sdf).reshape(-1, 1)
        alpha = np.zeros((len(s.z), 1), np.float32)
        self._fn("sdf2alpha")(*self._head(s), _p(s.fixed_dt), _p(s.dt), _p(sdf), c_f(inv_s), c_i(int(dynamic)), c_f(mult),
                              _p(alpha))
        return alpha

    def sum_over_each_ray(self, s, v):
        v = f32(v)
        C = v.shape[1]
        s_ray, s_smp = np.zeros((s.R, C), np.float32), np.zeros((len(s.z), C), np.float32)
        self._fn("sum_over_each_ray")(*self._head(s), c_i(C), _p(v), _p(s_ray), _p(s_smp))
        return s_ray, s_smp

    def sum_over_each_ray_backward(self, s, g_ray, g_smp, v):
        g_ray, g_smp, v = f32(g_ray), f32(g_smp), f32(v)
        C = v.shape[1]
        g = np.zeros((len(s.z), C), np.float32)
        if self.kind == "port":
            self._fn("sum_over_each_ray_backward")(*self._head(s), c_i(C), _p(g_ray), _p(g_smp), _p(g))
        else:
            self._fn("sum_over_each_ray_backward")(*self._head(s), c_i(C), _p(g_ray), _p(g_smp), _p(v), _p(g))
        return g

    def cumsum(self, s, v, inverse):
        v = f32(v).reshape(-1, 1)
        out = np.zeros((len(s.z), 1), np.float32)
        if self.kind == "port":
            self._fn("cumsum")(*self._head(s), _p(v), c_i(int(inverse)), c_i(0), _p(out))
        else:
            self._fn("cumsum")(*self._head(s), _p(v), c_i(int(inverse)), _p(out))
        return out

    def compute_cdf(self, s, w):
        w = f32(w).reshape(-1, 1)
        out = np.zeros((len(s.z), 1), np.float32)
        if self.kind == "port":
            self._fn("cumsum")(*self._head(s), _p(w), c_i(0), c_i(1), _p(out))
        else:
            self._fn("compute_cdf")(*self._head(s), _p(w), _p(out))
        return out

    def importance_sample(self, s, o, d, cdf, nimp, jitter=False, rng=(PCG_STATE, PCG_INC)):
        o, d, cdf = f32(o), f32(d), f32(cdf).reshape(-1, 1)
        imp = Samples(s.R, s.R * nimp)
        imp.equal, imp.fixed = True, nimp
        args = (_p(o), _p(d), _p(s.fixed_dt), _p(s.z), _p(cdf), c_i(nimp), c_u64(rng[0]), c_u64(rng[1]), c_i(int(jitter)),
                _p(imp.pos), _p(imp.dirs), _p(imp.z))
        if self.kind == "port":
            self._fn("importance_sample")(*self._head(s), *args)
        else:
            self._fn("importance_sample")(*self._head(s), *args, _p(imp.start_end))
        return imp

    def combine(self, s, imp, o, d, t_exit):
        o, d, t_exit = f32(o), f32(d), f32(t_exit).reshape(-1, 1)
        Mc = len(s.z) + imp.max_nr_samples
        c = Samples(s.R, Mc)
        c.has_sdf = s.has_sdf
        self._fn("combine")(*self._head(s), _p(o), _p(d), _p(t_exit), _p(s.fixed_dt), _p(s.z), _p(s.sdf), c_i(int(s.has_sdf)),
                            c_i(imp.fixed), _p(imp.z), _p(imp.sdf), c_i(Mc), _p(c.pos), _p(c.dirs), _p(c.z), _p(c.dt),
                            _p(c.sdf), _p(c.fixed_dt), _p(c.start_end), _p(c.cur))
        return c
