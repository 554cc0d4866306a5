"""The cross-lane arithmetic of the encode kernels, emulated lane by lane in numpy (no GPU): the DPP moves they are built on
(row_shr / row_shl inside a 16-lane row with bound_ctrl: a lane that would read across the edge of its row receives 0;
row_bcast:15 / row_bcast:31 with a row mask) and on top of them
  * combine_runs16 (csrc/encode.hip): segmented inclusive scan over runs of equal rows with a FLOAT run flag -- the conditional
    add is v = shifted(v) * j + v, the flag update j = j * shifted(j) -- and the rule for the lane that owns a run;
  * psdf::wave_incl_scan_add_i (csrc/psdf_common.h): the integer scan over the 64 lanes of a wave;
  * the integer identities compute_simplex / vertex_rows (csrc/encode_device.h) rely on since round 5: the rank wrap as a two's
    complement remainder, and the vertex rows from per-rank gathered terms.
The GPU parity tests (tests/test_gpu_encoding.py, test_gpu_hotpath_parity.py) hold the kernels themselves against the oracle;
here the constructions are held against their plain definitions on adversarial lane patterns."""
import numpy as np

NONE = 0xFFFFFFFF
LANE = np.arange(64)
ROW, COL = LANE >> 4, LANE & 15


def row_shr(x, d, fill=0):
    """v_mov_b32_dpp row_shr:d bound_ctrl:1 -- lane l reads lane l - d of its own 16-lane row, `fill` across the edge"""
    out = np.full_like(x, fill)
    ok = COL >= d
    out[ok] = x[LANE[ok] - d]
    return out


def row_shl(x, d, fill=0):
    out = np.full_like(x, fill)
    ok = COL + d <= 15
    out[ok] = x[LANE[ok] + d]
    return out


def combine_runs16(key, v):
    """the kernel's construction: returns (v_out, own)"""
    key = key.astype(np.uint32)
    v = v.astype(np.float32).copy()
    kprev = row_shr(key.astype(np.int64), 1)            # 0 shifted in (may equal a real key 0: see below)
    joined = (kprev == key.astype(np.int64)).astype(np.float32)
    j = joined.copy()
    for d in (1, 2, 4, 8):
        v = (row_shr(v, d) * j[:, None] + v).astype(np.float32)   # one rounding of an exact product by 0 or 1: an add
        j = j * row_shr(j, d)
    next_joined = row_shl(joined, 1)
    own = (key != NONE) & (next_joined == 0)
    return v, own


def reference_runs(key, v):
    """plain definition: inside each 16-lane row, maximal runs of equal keys; the last lane of a run holds its sum"""
    sums = {}
    for r in range(4):
        l = 16 * r
        while l < 16 * r + 16:
            e = l
            while e + 1 < 16 * r + 16 and key[e + 1] == key[l]:
                e += 1
            if key[l] != NONE:
                acc = np.zeros(v.shape[1], np.float32)
                for q in range(l, e + 1):               # same association as the scan is NOT required: compare with tolerance 0 on
                    acc = acc + v[q]                    # integer-valued inputs below
                sums[e] = acc
            l = e + 1
    return sums


def patterns(rng):
    yield np.arange(64, dtype=np.uint32) * 7 % 50                       # no runs
    yield np.zeros(64, np.uint32)                                       # key 0 everywhere: the shifted-in 0 at a row edge
    yield np.repeat(np.arange(8, dtype=np.uint32), 8)                   # runs of 8 (two per row)
    yield np.repeat(np.arange(2, dtype=np.uint32), 32)                  # runs crossing row edges: must be cut at the edge
    k = np.array([3, 5, 3, 3, 5, 5, 5, 3] * 8, np.uint32)               # A B A: equal keys that are NOT adjacent
    yield k
    for _ in range(200):
        k = rng.integers(0, 4, 64).astype(np.uint32)
        k[rng.random(64) < 0.2] = NONE                                  # lanes without a contribution, also in runs
        yield k
    k = np.full(64, NONE, np.uint32)
    yield k


def test_run_combine_matches_the_plain_run_sums():
    rng = np.random.default_rng(5)
    for key in patterns(rng):
        v = rng.integers(-64, 64, (64, 2)).astype(np.float32)           # integer valued: every order of additions is exact
        v[key == NONE] = 0.0                                            # what the kernel hands in for such lanes
        out, own = combine_runs16(key, v)
        ref = reference_runs(key, v)
        assert set(np.nonzero(own)[0]) == set(ref.keys()), (key, own)
        for lane, s in ref.items():
            assert np.array_equal(out[lane], s), (key, lane, out[lane], s)


def test_a_lane_joined_to_the_shifted_in_zero_adds_only_zeros():
    """first lane of a row with key 0: kprev = 0 (bound_ctrl) == key, so it counts as joined -- everything it adds is shifted-in 0"""
    key = np.zeros(64, np.uint32)
    key[1:] = np.arange(1, 64)
    v = np.ones((64, 2), np.float32)
    out, own = combine_runs16(key, v)
    assert np.array_equal(out, v) and own.all()


def wave_incl_scan_add_i(v):
    v = v.astype(np.int64).copy()
    for d in (1, 2, 4, 8):
        v = v + row_shr(v, d)
    # row_bcast:15 row_mask:0xa -- lane 15 of row r-1 into every lane of rows 1 and 3 (other rows: old = 0)
    b = np.zeros_like(v)
    for r in (1, 3):
        b[ROW == r] = v[16 * (r - 1) + 15]
    v = v + b
    # row_bcast:31 row_mask:0xc -- lane 31 into every lane of rows 2 and 3
    b = np.zeros_like(v)
    b[ROW >= 2] = v[31]
    return v + b


def test_integer_wave_scan():
    rng = np.random.default_rng(6)
    for _ in range(100):
        v = rng.integers(0, 1 << 20, 64)
        assert np.array_equal(wave_incl_scan_add_i(v), np.cumsum(v))
    assert np.array_equal(wave_incl_scan_add_i(np.ones(64, np.int64)), np.arange(1, 65))


def test_rank_wrap_is_the_twos_complement_remainder():
    """compute_simplex, P + 1 = 4: r = rank + sum lies in [-4, 7]; the reference's wrap by +-4 is r & 3 in 32-bit two's complement"""
    for r in range(-4, 8):
        adj_ref = 4 if r < 0 else (-4 if r > 3 else 0)
        r32 = np.int32(r)
        adj = np.int32(np.int32(r32 & np.int32(3)) - r32)
        assert adj == adj_ref and 0 <= r + adj <= 3


def test_vertex_rows_from_per_rank_terms():
    """vertex_rows: h_r = h0 + r geom - sum_i [rank_i > P - r] t_i  ==  h0 + (r geom - (T_P + .. + T_(P+1-r))),  T_k = t_i of the
    hashed coordinate whose rank is k (0 if that rank belongs to the last, unhashed coordinate); 32-bit wrap-around arithmetic"""
    import itertools
    rng = np.random.default_rng(7)
    P = 3
    M = 1 << 32
    for perm in itertools.permutations(range(P + 1)):
        c = int(rng.integers(1, M))
        pw = [pow(c, e, M) for e in range(P + 1)]
        geom = sum(pw[1:]) % M
        h0 = int(rng.integers(0, M))
        t = [(P + 1) * pw[P - i] % M for i in range(P)]
        rank = list(perm)
        ref = [(h0 + r * geom - sum(t[i] for i in range(P) if rank[i] > P - r)) % M for r in range(P + 1)]
        T = [0] * (P + 1)
        for k in range(1, P + 1):
            for i in range(P):
                if rank[i] == k:
                    T[k] = t[i]
        rows, S = [h0], 0
        for r in range(1, P + 1):
            S = (S + T[P + 1 - r]) % M
            rows.append((h0 + (r * geom - S)) % M)
        assert rows == ref


def test_nearest_remainder_rule_equals_the_reference_rule_on_floats():
    """compute_simplex, P + 1 = 4: rem = (E > down + 2) ? down + 4 : down against the reference's (up - E) < (E - down) ? up : down
    with up = 4 ceil(E / 4), down = 4 floor(E / 4), all in float32 -- swept over +-2000 ulps around every multiple of 1/2 up to
    +-300 (the ties and their neighbours), the denormals, every float of a few unit intervals, and random values of all scales."""
    f32 = np.float32

    def ref(E):
        v = (E * f32(0.25)).astype(f32)
        up, down = (np.ceil(v) * f32(4)).astype(f32), (np.floor(v) * f32(4)).astype(f32)
        return np.where((up - E).astype(f32) < (E - down).astype(f32), up, down).astype(f32)

    def new(E):
        down = (np.floor((E * f32(0.25)).astype(f32)) * f32(4)).astype(f32)
        return np.where(E > (down + f32(2)).astype(f32), (down + f32(4)).astype(f32), down).astype(f32)

    def check(E):
        E = E.astype(f32)
        E = E[np.isfinite(E)]
        assert np.array_equal(ref(E), new(E))

    rng = np.random.default_rng(8)
    offs = np.arange(-2000, 2001)
    for b0 in np.arange(-600, 601) * 0.5:
        bits = np.array([b0], f32).view(np.int32)[0]
        check((bits + offs).astype(np.int32).view(f32))
    check(np.arange(-3000, 3001).astype(np.int32).view(f32))                       # +-denormals and the smallest normals
    check((np.arange(0, 3000) | 0x80000000).astype(np.uint32).view(f32))
    for lo, hi in [(1.5, 2.5), (5.5, 6.5), (2.0 ** 20 + 1.5, 2.0 ** 20 + 2.5), (2.0 ** 24 - 8, 2.0 ** 24 + 64)]:
        lb, hb = np.array([lo, hi], f32).view(np.int32)
        arr = np.arange(lb, hb + 1, dtype=np.int32).view(f32)
        check(arr)
        check(-arr)
    for scale in [1e-30, 1e-10, 1e-3, 1, 10, 1e3, 1e5, 1e6, 3e7, 1e9]:
        check(rng.standard_normal(1_000_000) * scale)
