"""Lane-level numpy emulation of ONE 32-sample tile of tools/prototypes/mlp_bwd_split_bf16_v2.hip (the split-bf16 MLP backward with
transposes on the matrix pipe), statement by statement, on the MFMA lane maps of tools/mfma_lane_maps.py: checks the whole
data flow (operand images, chaining, 0/1-operand transposes, piece packing, sample order of the X / dY row loads, bias
sums, accumulator layout) against a float64 backward of the same net.  CPU only; it does not model timing.
usage: python tools/prototypes/emulate_bwd_v2.py"""
import os
import sys

import numpy as np
from scipy.special import erf

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mfma_lane_maps import row_of  # noqa: E402

K0, HID, S0, SH = 36, 64, 3, 4
LANES = np.arange(64)
H_OF, SL_OF = LANES >> 5, LANES & 31
F32 = np.float32


def split3(x):
    r = x.astype(F32)
    out = []
    for _ in range(3):
        top = (r.view(np.uint32) & np.uint32(0xFFFF0000)).view(F32)
        out.append(top)
        r = (r - top).astype(F32)
    return out


def mfma(A, B, C):
    """A, B: [64][8] (bf16-valued), C: [64][16] fp32 -> C + A B with the 32x32x16 lane maps; fp32 accumulate"""
    Am, Bm = np.zeros((32, 16)), np.zeros((16, 32))
    for lane in range(64):
        m, hh = lane & 31, lane >> 5
        Am[m, 8 * hh:8 * hh + 8] = A[lane]
        Bm[8 * hh:8 * hh + 8, m] = B[lane]
    Dm = Am @ Bm
    D = C.astype(np.float64).copy()
    for lane in range(64):
        n, h = lane & 31, lane >> 5
        for r in range(16):
            D[lane, r] += Dm[row_of(r, h), n]
    return D.astype(F32)


def image(Wl, ns, col_of):
    """[to][s][piece][lane][j] operand image; col_of(s, hh, j) -> k index of W's second axis, rows 32 to + m"""
    rows, cols = Wl.shape
    img = np.zeros((2, ns, 3, 64, 8), dtype=F32)
    for to in range(2):
        for s in range(ns):
            for lane in range(64):
                m, hh = lane & 31, lane >> 5
                for j in range(8):
                    r_, c_ = 32 * to + m, col_of(s, hh, j)
                    v = Wl[r_, c_] if (r_ < rows and c_ < cols) else 0.0
                    for p, piece in enumerate(split3(np.array([v], dtype=F32))):
                        img[to, s, p, lane, j] = piece[0]
    return img


def mac(out, x, img, s):
    """split_mac: out[to] += W-step x (six products, smallest first); returns the pieces of x"""
    b = split3(x)
    for to in range(2):
        a1, a2, a3 = img[to, s, 0], img[to, s, 1], img[to, s, 2]
        for A, Bp in ((a3, b[0]), (a2, b[1]), (a1, b[2]), (a2, b[0]), (a1, b[1]), (a1, b[0])):
            out[to] = mfma(A, Bp, out[to])
    return b


def ident_op(sp):
    o = np.zeros((64, 8), dtype=F32)
    for lane in range(64):
        n, hh = lane & 31, lane >> 5
        for j in range(8):
            o[lane, j] = 1.0 if n == row_of(8 * sp + j, hh) else 0.0
    return o


ID = [ident_op(0), ident_op(1)]


def chain_tiles(inp, out, img, per_tile):
    for ti in range(2):
        kp = []
        for sp in range(2):
            kp.append(mac(out, inp[ti][:, 8 * sp:8 * sp + 8], img, 2 * ti + sp))
        per_tile(ti, kp)        # kp[sp][piece] : [64][8]


def transpose_f32(kp):
    o = np.zeros((64, 16), dtype=F32)
    for sp in range(2):
        for piece in (2, 1, 0):
            o = mfma(kp[sp][piece], ID[sp], o)
    return o


def transpose_pieces(kp):
    """-> pieces[ks][piece] : [64][8] in feature-lane order, and the per-lane fp32 sum of the tile"""
    out = [[None] * 3 for _ in range(2)]
    total = np.zeros(64, dtype=F32)
    for piece in range(3):
        o = np.zeros((64, 16), dtype=F32)
        for sp in range(2):
            o = mfma(kp[sp][piece], ID[sp], o)
        total = (total + o.sum(axis=1)).astype(F32)
        assert not np.any(o.view(np.uint32) & np.uint32(0xFFFF)), "transposed piece is not bf16-valued"
        out[0][piece], out[1][piece] = o[:, 0:8], o[:, 8:16]
    return out, total


def split_tile(t):
    return [split3(t[:, 8 * ks:8 * ks + 8]) for ks in range(2)]


def dw_mac(acc, A, B):
    for ks in range(2):
        for pa, pb in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):
            acc = mfma(A[ks][pa], B[ks][pb], acc)
    return acc


def gelu_both(z):
    z = z.astype(np.float64)
    cdf = 0.5 * (1 + erf(z * 0.7071067811865476))
    return (z * cdf).astype(F32), (cdf + z * np.exp(-0.5 * z * z) * 0.3989422804014327).astype(F32)


def bias_init(b):
    out = np.zeros((2, 64, 16), dtype=F32)
    for to in range(2):
        for lane in range(64):
            for r in range(16):
                out[to, lane, r] = b[32 * to + row_of(r, lane >> 5)]
    return out


def load_row_tile(row):
    """row: 32 floats of one feature-major row -> per lane-half h: registers 4q..4q+3 = samples 8q + 4h .. + 3"""
    o = np.zeros((64, 16), dtype=F32)
    for lane in range(64):
        h = lane >> 5
        for q in range(4):
            o[lane, 4 * q:4 * q + 4] = row[8 * q + 4 * h:8 * q + 4 * h + 4]
    return o


def from_d_layout(t, rows):
    """[to][lane][r] D tiles (rows = features 32 to + row_of, cols = lane & 31) -> matrix [rows][32]"""
    M = np.zeros((rows, 32))
    for to in range(t.shape[0]):
        for lane in range(64):
            for r in range(16):
                k = 32 * to + row_of(r, lane >> 5)
                if k < rows:
                    M[k, lane & 31] = t[to, lane, r]
    return M


def main():
    rng = np.random.default_rng(5)
    W = [rng.standard_normal((HID, K0)) * (2 / K0) ** 0.5, rng.standard_normal((HID, HID)) * (2 / HID) ** 0.5,
         rng.standard_normal((HID, HID)) * (2 / HID) ** 0.5, rng.standard_normal((1, HID)) * (2 / HID) ** 0.5]
    W = [w.astype(F32) for w in W]
    Bv = [(rng.standard_normal(HID) * 0.1).astype(F32) for _ in range(3)]
    X = rng.standard_normal((K0, 32)).astype(F32)          # [feature][sample] of one tile
    dY = rng.standard_normal(32).astype(F32)

    chained = lambda s, hh, j: 32 * (s >> 1) + row_of(8 * (s & 1) + j, hh)
    F0 = image(W[0], S0, lambda s, hh, j: 16 * s + 8 * hh + j)
    F1, F2 = image(W[1], SH, chained), image(W[2], SH, chained)
    T2, T1 = image(W[2].T.copy(), SH, chained), image(W[1].T.copy(), SH, chained)
    T0 = image(W[0].T.copy(), SH, chained)                  # rows = input neurons (36 of 64), k = output neurons

    # ---------------- forward
    a = bias_init(Bv[0])
    for s in range(S0):
        xs = np.zeros((64, 8), dtype=F32)
        for lane in range(64):
            for j in range(8):
                k = 16 * s + 8 * (lane >> 5) + j
                xs[lane, j] = X[k, lane & 31] if k < K0 else 0.0
        mac(a, xs, F0, s)
    a, g1 = gelu_both(a)
    b = bias_init(Bv[1])
    h1T = [None, None]
    chain_tiles(a, b, F1, lambda ti, kp: h1T.__setitem__(ti, transpose_f32(kp)))
    b, g2 = gelu_both(b)
    c = bias_init(Bv[2])
    h2T = [None, None]
    chain_tiles(b, c, F2, lambda ti, kp: h2T.__setitem__(ti, transpose_f32(kp)))
    c, dz = gelu_both(c)
    # ---------------- output layer
    dyT = load_row_tile(dY)
    db4 = dyT.sum(axis=1)
    dw4 = np.zeros((2, 64))
    for to in range(2):
        kp = [split3(c[to][:, 8 * sp:8 * sp + 8]) for sp in range(2)]
        h3T = transpose_f32(kp)
        dw4[to] = (h3T.astype(np.float64) * dyT).sum(axis=1)
    dy_lane = dY[SL_OF]
    for to in range(2):
        for lane in range(64):
            for r in range(16):
                dz[to, lane, r] *= W[3][0, 32 * to + row_of(r, lane >> 5)] * dy_lane[lane]
    # ---------------- layers 3, 2, 1
    dW = {l: np.zeros((2, 2, 64, 16), dtype=F32) for l in (1, 2, 3)}
    db = {l: np.zeros((2, 64), dtype=F32) for l in (1, 2, 3)}

    def layer(l, dzin, out, Timg, hT):
        A = [None, None]

        def per(to, kp):
            A[to], s_ = transpose_pieces(kp)
            db[l][to] += s_
        chain_tiles(dzin, out, Timg, per)
        for ti in range(2):
            Bp = split_tile(hT[ti])
            for to in range(2):
                dW[l][to, ti] = dw_mac(dW[l][to, ti], A[to], Bp)

    c = np.zeros((2, 64, 16), dtype=F32)
    layer(3, dz, c, T2, h2T)
    c = (c * g2).astype(F32)
    dz = np.zeros((2, 64, 16), dtype=F32)
    layer(2, c, dz, T1, h1T)
    dz = (dz * g1).astype(F32)
    c = np.zeros((2, 64, 16), dtype=F32)
    xT = []
    for ti in range(2):
        t = np.zeros((64, 16), dtype=F32)
        for lane in range(64):
            feat = 32 * ti + (lane & 31)
            if feat < K0:
                t[lane] = load_row_tile(X[feat])[lane]
        xT.append(t)
    layer(1, dz, c, T0, xT)
    dX = from_d_layout(c, K0)

    # ---------------- float64 reference
    x = X.astype(np.float64)
    z1 = W[0].astype(np.float64) @ x + Bv[0][:, None]
    gel = lambda z: 0.5 * z * (1 + erf(z * 0.7071067811865476))
    gp = lambda z: 0.5 * (1 + erf(z * 0.7071067811865476)) + z * np.exp(-0.5 * z * z) * 0.3989422804014327
    h1 = gel(z1)
    z2 = W[1].astype(np.float64) @ h1 + Bv[1][:, None]
    h2 = gel(z2)
    z3 = W[2].astype(np.float64) @ h2 + Bv[2][:, None]
    h3 = gel(z3)
    dz3 = W[3].astype(np.float64).T * dY[None, :] * gp(z3)
    dz2 = (W[2].astype(np.float64).T @ dz3) * gp(z2)
    dz1 = (W[1].astype(np.float64).T @ dz2) * gp(z1)
    ref = {"dX": W[0].astype(np.float64).T @ dz1, "dW1": dz1 @ x.T, "dW2": dz2 @ h1.T, "dW3": dz3 @ h2.T,
           "db1": dz1.sum(1), "db2": dz2.sum(1), "db3": dz3.sum(1), "dW4": h3 @ dY, "db4": dY.sum()}

    def dw_matrix(l, cols):
        M = np.zeros((HID, 64))
        for to in range(2):
            for ti in range(2):
                for lane in range(64):
                    for r in range(16):
                        M[32 * to + row_of(r, lane >> 5), 32 * ti + (lane & 31)] = dW[l][to, ti, lane, r]
        return M[:, :cols]

    def lane_vec(v):          # per-lane partials of lane (f, h): add the two halves
        return np.array([[v[to][f] + v[to][f + 32] for f in range(32)] for to in range(2)]).reshape(-1)

    got = {"dX": dX, "dW1": dw_matrix(1, K0), "dW2": dw_matrix(2, HID), "dW3": dw_matrix(3, HID), "db1": lane_vec(db[1]),
           "db2": lane_vec(db[2]), "db3": lane_vec(db[3]), "dW4": lane_vec(dw4), "db4": db4[0] + db4[32]}
    ok = True
    for k in ref:
        e = np.abs(np.asarray(got[k]) - ref[k]).max()
        m = np.abs(ref[k]).max()
        print("%-4s max |err| %.3e  max |ref| %.3e  rel %.1e" % (k, e, m, e / m))
        ok &= e <= 2e-5 * m
    print("emulated tile of the v2 backward matches float64" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
