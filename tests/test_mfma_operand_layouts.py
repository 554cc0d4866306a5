"""The operand layouts csrc/mlp_bwd_split_f16.hip is built on, emulated lane by lane in numpy (no GPU): v_mfma_f32_16x16x32_f16
as D[i][j] += sum_k A[i][k] B[k][j] with
    A operand: lane l holds A[l & 15][8 (l >> 4) .. + 7],   B operand: lane l holds B[8 (l >> 4) .. + 7][l & 15],
    C / D:     lane l holds D[4 (l >> 4) + r][l & 15], r = 0 .. 3,
and on top of it the four things the kernel does with it: a chain layer out^T = W in^T with the k order kf() of its weight image
(mlp_split_pack_kernel), the transposition of an operand by an MFMA against a 0/1 matrix (ident_op / transpose_f32), the
parameter-gradient block with two piece products per MFMA in the two halves of K (transpose_pieces / split4 / dw_mac), and the
layout of the dX rows a lane stores.  Every emulated result is compared with the plain matrix product.  The kernel's own
arithmetic (fp16 pieces) is tests/test_split_f16_numerics.py; here values are small integers, exact in every format."""
import numpy as np

LANES = np.arange(64)
C_, G_ = LANES & 15, LANES >> 4


def mfma(a_op, b_op, c):
    """a_op, b_op [64, 8], c [64, 4] (lane-major registers) -> d [64, 4]"""
    A = np.zeros((16, 32))
    B = np.zeros((32, 16))
    for l in range(64):
        A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = a_op[l]
        B[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = b_op[l]
    D = A @ B
    d = c.copy()
    for l in range(64):
        d[l] += D[4 * (l >> 4):4 * (l >> 4) + 4, l & 15]
    return d


def kf(s, g, j):
    """feature index of element j of the k-step-s operand of lane group g (the kernel's kf())"""
    return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3)


def chain_operand(act, s):
    """B operand of k-step s from activation tiles (D layout: act[t][lane, r] = feature 16 t + 4 g + r of sample c)"""
    return np.concatenate([act[2 * s], act[2 * s + 1]], axis=1)          # [64, 8]: step_operand()


def weight_record(W, tile, s):
    """A operand of output tile `tile`, k-step s, as mlp_split_pack_kernel lays it out: lane (c, g), element j = W[16 tile + c][kf]"""
    rec = np.zeros((64, 8))
    for l in range(64):
        for j in range(8):
            rec[l, j] = W[16 * tile + (l & 15), kf(s, l >> 4, j)]
    return rec


def to_tiles(Xfs):
    """[64 features, 16 samples] -> 4 D-layout tiles [64 lanes, 4]"""
    tiles = []
    for t in range(4):
        d = np.zeros((64, 4))
        for l in range(64):
            d[l] = Xfs[16 * t + 4 * (l >> 4):16 * t + 4 * (l >> 4) + 4, l & 15]
        tiles.append(d)
    return tiles


def ident_op(u):
    """0/1 B operand selecting the 16 features of tile 2 s + u out of a k-step (the kernel's ident_op)"""
    op = np.zeros((64, 8))
    for l in range(64):
        for j in range(8):
            op[l, j] = 1.0 if ((j >> 2) == u and 4 * (l >> 4) + (j & 3) == (l & 15)) else 0.0
    return op


def test_chain_layer_with_the_image_k_order():
    rng = np.random.default_rng(0)
    W = rng.integers(-4, 5, (64, 64)).astype(float)
    H = rng.integers(-4, 5, (64, 16)).astype(float)                      # [feature][sample]
    act = to_tiles(H)
    out = [np.zeros((64, 4)) for _ in range(4)]
    for s in range(2):
        b = chain_operand(act, s)
        for t in range(4):
            out[t] = mfma(weight_record(W, t, s), b, out[t])
    want = to_tiles(W @ H)
    for t in range(4):
        assert np.array_equal(out[t], want[t])


def test_transposition_by_an_mfma_against_a_zero_one_operand():
    """transpose_f32: the chain operand of a k-step goes in as the A operand, the 0/1 matrix as B; out comes the feature-lane tile:
    register r of lane (f, g) = feature f of tile 2 s + u, sample 4 g + r"""
    rng = np.random.default_rng(1)
    H = rng.integers(-9, 10, (64, 16)).astype(float)
    act = to_tiles(H)
    for s in range(2):
        b = chain_operand(act, s)
        for u in range(2):
            hT = mfma(b, ident_op(u), np.zeros((64, 4)))
            for l in range(64):
                f, g = l & 15, l >> 4
                assert np.array_equal(hT[l], H[16 * (2 * s + u) + f, 4 * g:4 * g + 4])


def test_parameter_gradient_block_two_piece_products_per_mfma():
    """dW[to][ti] += dZ(to) H(ti)^T over the 16 samples of a tile.  A lane of a transposed tile owns four samples (k slots 0 .. 3 of
    its group); slots 4 .. 7 carry the OTHER PIECE of the same samples on the dZ side and a copy of the same piece on the H side:
    [a0|a1] x [b1|b1] then [a0|a1] x [b0|b0] = (a0 + a1)(b0 + b1), two MFMAs per 16 x 16 block"""
    rng = np.random.default_rng(2)
    dz0, dz1 = rng.integers(-3, 4, (16, 16)).astype(float), rng.integers(-3, 4, (16, 16)).astype(float)   # pieces, [feature][sample]
    h0, h1 = rng.integers(-3, 4, (16, 16)).astype(float), rng.integers(-3, 4, (16, 16)).astype(float)

    def feature_lane(P):                  # [64 lanes, 4]: lane (f, g) -> samples 4 g .. + 3 of feature f
        return np.stack([P[l & 15, 4 * (l >> 4):4 * (l >> 4) + 4] for l in range(64)])
    a_t01 = np.concatenate([feature_lane(dz0), feature_lane(dz1)], axis=1)     # AT.t01
    b_t00 = np.concatenate([feature_lane(h0), feature_lane(h0)], axis=1)       # BT.t00
    b_t11 = np.concatenate([feature_lane(h1), feature_lane(h1)], axis=1)       # BT.t11
    acc = mfma(a_t01, b_t11, np.zeros((64, 4)))                                # smallest first
    acc = mfma(a_t01, b_t00, acc)
    want = (dz0 + dz1) @ (h0 + h1).T                                           # [out feature][in feature]
    for l in range(64):                                                        # the epilogue's mapping: row 4 g + r, column c
        assert np.array_equal(acc[l], want[4 * (l >> 4):4 * (l >> 4) + 4, l & 15])


def test_data_gradient_chain_uses_the_transposed_image_and_lands_in_row_order():
    """dX^T = W0^T dZ1^T with the T0 image (pack case 5: row = INPUT feature, k = output neuron in kf order); a lane then stores
    rows 16 t + 4 g + r of its sample c -- the layout of the dX stores"""
    rng = np.random.default_rng(3)
    K0 = 36
    W0 = rng.integers(-4, 5, (64, K0)).astype(float)
    dZ = rng.integers(-4, 5, (64, 16)).astype(float)
    W0T = np.zeros((64, 64))
    W0T[:K0] = W0.T                                                            # rows past K0 are zero in the image
    act = to_tiles(dZ)
    dx = [np.zeros((64, 4)) for _ in range(3)]
    for s in range(2):
        b = chain_operand(act, s)
        for t in range(3):
            dx[t] = mfma(weight_record(W0T, t, s), b, dx[t])
    want = W0.T @ dZ                                                           # [K0][sample]
    for t in range(3):
        for l in range(64):
            for r in range(4):
                k = 16 * t + 4 * (l >> 4) + r
                assert dx[t][l, r] == (want[k, l & 15] if k < K0 else 0.0)
