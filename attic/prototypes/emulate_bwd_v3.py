"""Lane-level numpy emulation of ONE 16-sample tile of tools/prototypes/mlp_bwd_split_bf16_v3.hip: the split-bf16 MLP backward on
v_mfma_f32_16x16x32_bf16 (16-sample tiles halve the per-sample register state that makes the 32-sample versions spill).
Lane maps of the 16x16x32 instruction (c = lane & 15, g = lane >> 4):
    A[m][k]: lane (m = c, g) holds k = 8 g + j;   B[k][n]: lane (n = c, g) holds k = 8 g + j;   D[m][n]: lane (n = c, g),
    register r holds row m = 4 g + r.
Chained order of a k-step s (32 features = the D tiles 2s and 2s+1 of the previous layer): k = (g, j) <-> feature
    kf(s, g, j) = 32 s + 16 (j >> 2) + 4 g + (j & 3).
Transposes are products with a 0/1 operand (two per k-step, one per 16-feature tile); a feature-lane tile holds samples
4 g + r in register r, which sit in k-slots 8 g + r of a dW operand (slots 8 g + 4 .. + 7 are zero: half-filled k).
Checks all nine gradients against float64.  CPU only.
usage: python tools/prototypes/emulate_bwd_v3.py"""
import sys

import numpy as np
from scipy.special import erf

K0, HID = 36, 64
NT = HID // 16          # 16-feature tiles per hidden layer
F32 = np.float32


def kf(s, g, j):
    return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3)


def split3(x):
    r = x.astype(F32)
    out = []
    for _ in range(3):
        top = (r.view(np.uint32) & np.uint32(0xFFFF0000)).view(F32)
        out.append(top)
        r = (r - top).astype(F32)
    return out


def mfma16(A, B, C):
    """A, B: [64][8]; C: [64][4] -> C + A B (16x16x32 lane maps), fp32 accumulate"""
    Am, Bm = np.zeros((16, 32)), np.zeros((32, 16))
    for lane in range(64):
        c, g = lane & 15, lane >> 4
        Am[c, 8 * g:8 * g + 8] = A[lane]
        Bm[8 * g:8 * g + 8, c] = B[lane]
    Dm = Am @ Bm
    D = C.astype(np.float64).copy()
    for lane in range(64):
        c, g = lane & 15, lane >> 4
        for r in range(4):
            D[lane, r] += Dm[4 * g + r, c]
    return D.astype(F32)


def image(Wl, ntile, nstep, col_of):
    """[t][s][piece][lane][j]: rows 16 t + c of Wl, columns col_of(s, g, j)"""
    rows, cols = Wl.shape
    img = np.zeros((ntile, nstep, 3, 64, 8), dtype=F32)
    for t in range(ntile):
        for s in range(nstep):
            for lane in range(64):
                c, g = lane & 15, lane >> 4
                for j in range(8):
                    r_, c_ = 16 * t + c, col_of(s, g, j)
                    v = Wl[r_, c_] if (r_ < rows and c_ < cols) else 0.0
                    for p, piece in enumerate(split3(np.array([v], dtype=F32))):
                        img[t, s, p, lane, j] = piece[0]
    return img


PRODUCTS = ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0))   # (weight piece, activation piece), smallest first


def mac(out, bp, img, s):
    """out[t] += W(tile t, k-step s) x pieces bp (six products), t interleaved inside each product"""
    for pa, pb in PRODUCTS:
        for t in range(out.shape[0]):
            out[t] = mfma16(img[t, s, pa], bp[pb], out[t])


def step_operand(act, s):
    """B operand of k-step s from the D tiles 2s, 2s+1 of an activation [tiles][64][4] -> [64][8]"""
    return np.concatenate([act[2 * s], act[2 * s + 1]], axis=1)


def ident_op(u):
    """0/1 operand selecting the 16 features of tile 2s+u out of a k-step: independent of s"""
    o = np.zeros((64, 8), dtype=F32)
    for lane in range(64):
        c, g = lane & 15, lane >> 4
        for j in range(8):
            o[lane, j] = 1.0 if ((j >> 2) == u and 4 * g + (j & 3) == c) else 0.0
    return o


ID = [ident_op(0), ident_op(1)]


def chain(act, out, img, per_step):
    for s in range(act.shape[0] // 2):
        bp = split3(step_operand(act, s))
        mac(out, bp, img, s)
        per_step(s, bp)


def transpose_f32(bp, u):
    o = np.zeros((64, 4), dtype=F32)
    for piece in (2, 1, 0):
        o = mfma16(bp[piece], ID[u], o)
    return o                     # lane (f = c, g): register r = feature 16(2s+u)+c of sample 4g + r


def dw_operand(regs4):
    """[64][4] bf16-valued -> [64][8] with the four samples in k-slots 8g .. 8g+3 and zeros above"""
    return np.concatenate([regs4, np.zeros((64, 4), dtype=F32)], axis=1)


def transpose_pieces(bp, u):
    ops, total = [], np.zeros(64, dtype=F32)
    for piece in range(3):
        o = mfma16(bp[piece], ID[u], np.zeros((64, 4), dtype=F32))
        assert not np.any(o.view(np.uint32) & np.uint32(0xFFFF))
        total = (total + o.sum(axis=1)).astype(F32)
        ops.append(dw_operand(o))
    return ops, total


def split_fl(tile4):
    return [dw_operand(p) for p in split3(tile4)]


def dw_mac(acc, A, B):
    for pa, pb in PRODUCTS:
        acc = mfma16(A[pa], B[pb], acc)
    return acc


def gelu_both(z):
    z = z.astype(np.float64)
    cdf = 0.5 * (1 + erf(z * 0.7071067811865476))
    return (z * cdf).astype(F32), (cdf + z * np.exp(-0.5 * z * z) * 0.3989422804014327).astype(F32)


def bias_init(b, ntile=NT):
    out = np.zeros((ntile, 64, 4), dtype=F32)
    for t in range(ntile):
        for lane in range(64):
            for r in range(4):
                out[t, lane, r] = b[16 * t + 4 * (lane >> 4) + r]
    return out


def main():
    rng = np.random.default_rng(11)
    W = [rng.standard_normal((HID, K0)) * (2 / K0) ** 0.5, rng.standard_normal((HID, HID)) * (2 / HID) ** 0.5,
         rng.standard_normal((HID, HID)) * (2 / HID) ** 0.5, rng.standard_normal((1, HID)) * (2 / HID) ** 0.5]
    W = [w.astype(F32) for w in W]
    Bv = [(rng.standard_normal(HID) * 0.1).astype(F32) for _ in range(3)]
    X = rng.standard_normal((K0, 16)).astype(F32)          # [feature][sample], one tile
    dY = rng.standard_normal(16).astype(F32)

    F0 = image(W[0], NT, 2, lambda s, g, j: 32 * s + 8 * g + j)          # layer 0: natural k order, K0 padded to 64
    F1, F2 = image(W[1], NT, 2, kf), image(W[2], NT, 2, kf)
    T2, T1 = image(W[2].T.copy(), NT, 2, kf), image(W[1].T.copy(), NT, 2, kf)
    T0 = image(W[0].T.copy(), 3, 2, kf)                                  # rows = input neurons: 3 tiles cover 36

    # ---------------- forward
    a = bias_init(Bv[0])
    for s in range(2):
        xs = np.zeros((64, 8), dtype=F32)
        for lane in range(64):
            c, g = lane & 15, lane >> 4
            for j in range(8):
                k = 32 * s + 8 * g + j
                xs[lane, j] = X[k, c] if k < K0 else 0.0
        mac(a, split3(xs), F0, s)
    a, g1 = gelu_both(a)
    h1T, h2T = [None] * NT, [None] * NT

    def keep(dst):
        def f(s, bp):
            for u in range(2):
                dst[2 * s + u] = transpose_f32(bp, u)
        return f
    b = bias_init(Bv[1])
    chain(a, b, F1, keep(h1T))
    b, g2 = gelu_both(b)
    c = bias_init(Bv[2])
    chain(b, c, F2, keep(h2T))
    c, dz = gelu_both(c)                                                  # c = h3, dz = gelu'(z3)
    # ---------------- output layer
    dyT = np.zeros((64, 4), dtype=F32)
    for lane in range(64):
        dyT[lane] = dY[4 * (lane >> 4):4 * (lane >> 4) + 4]
    db4 = dyT.sum(axis=1)                                                 # lanes c == 0 carry it (4 groups)
    dw4 = np.zeros((NT, 64))
    for s in range(2):
        bp = split3(step_operand(c, s))
        for u in range(2):
            h3T = transpose_f32(bp, u)
            dw4[2 * s + u] = (h3T.astype(np.float64) * dyT).sum(axis=1)
    for t in range(NT):
        for lane in range(64):
            for r in range(4):
                dz[t, lane, r] *= W[3][0, 16 * t + 4 * (lane >> 4) + r] * dY[lane & 15]
    # ---------------- layers
    dW = {1: np.zeros((NT, 3, 64, 4), dtype=F32), 2: np.zeros((NT, NT, 64, 4), dtype=F32), 3: np.zeros((NT, NT, 64, 4), dtype=F32)}
    db = {l: np.zeros((NT, 64), dtype=F32) for l in (1, 2, 3)}

    def layer(l, dzin, out, Timg, hT):
        A = [None] * NT

        def per(s, bp):
            for u in range(2):
                A[2 * s + u], s_ = transpose_pieces(bp, u)
                db[l][2 * s + u] += s_
        chain(dzin, out, Timg, per)
        for ti in range(len(hT)):
            Bp = split_fl(hT[ti])
            for to in range(NT):
                dW[l][to, ti] = dw_mac(dW[l][to, ti], A[to], Bp)

    c = np.zeros((NT, 64, 4), dtype=F32)
    layer(3, dz, c, T2, h2T)
    c = (c * g2).astype(F32)
    dz = np.zeros((NT, 64, 4), dtype=F32)
    layer(2, c, dz, T1, h1T)
    dz = (dz * g1).astype(F32)
    xT = []
    for u in range(3):
        t = np.zeros((64, 4), dtype=F32)
        for lane in range(64):
            feat = 16 * u + (lane & 15)
            if feat < K0:
                t[lane] = X[feat, 4 * (lane >> 4):4 * (lane >> 4) + 4]
        xT.append(t)
    cx = np.zeros((3, 64, 4), dtype=F32)
    layer(1, dz, cx, T0, xT)
    dX = np.zeros((K0, 16))
    for t in range(3):
        for lane in range(64):
            for r in range(4):
                k = 16 * t + 4 * (lane >> 4) + r
                if k < K0:
                    dX[k, lane & 15] = cx[t, lane, r]

    # ---------------- float64 reference
    x = X.astype(np.float64)
    gel = lambda z: 0.5 * z * (1 + erf(z * 0.7071067811865476))
    gp = lambda z: 0.5 * (1 + erf(z * 0.7071067811865476)) + z * np.exp(-0.5 * z * z) * 0.3989422804014327
    z1 = W[0].astype(np.float64) @ x + Bv[0][:, None]
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
        nti = dW[l].shape[1]
        M = np.zeros((HID, 16 * nti))
        for to in range(NT):
            for ti in range(nti):
                for lane in range(64):
                    for r in range(4):
                        M[16 * to + 4 * (lane >> 4) + r, 16 * ti + (lane & 15)] = dW[l][to, ti, lane, r]
        return M[:, :cols]

    def lane_vec(v):          # lane (f = c, g): add the four sample groups
        return np.array([[sum(v[t][f + 16 * g] for g in range(4)) for f in range(16)] for t in range(len(v))]).reshape(-1)

    got = {"dX": dX, "dW1": dw_matrix(1, K0), "dW2": dw_matrix(2, HID), "dW3": dw_matrix(3, HID), "db1": lane_vec(db[1]),
           "db2": lane_vec(db[2]), "db3": lane_vec(db[3]), "dW4": lane_vec(dw4), "db4": sum(db4[16 * g] for g in range(4))}
    ok = True
    for k in ref:
        e = np.abs(np.asarray(got[k]) - ref[k]).max()
        m = np.abs(ref[k]).max()
        print("%-4s max |err| %.3e  max |ref| %.3e  rel %.1e" % (k, e, m, e / m))
        ok &= e <= 2e-5 * m
    print("emulated tile of the v3 backward matches float64" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
