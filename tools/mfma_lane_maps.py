"""numpy model of the v_mfma_f32_32x32x16_bf16 lane maps, used to check the index logic of the split-bf16 MLP kernels
before they reach the GPU (tools/mlp_*_split_bf16*.hip, csrc/mlp_device.h):
  A[m][k]: lane m + 32 hh holds k = 8 hh + j (j = 0..7);  B[k][n]: lane n + 32 hh holds k = 8 hh + j;
  D[m][n]: lane n + 32 h, register r holds row m = row_of(r, h) = (r & 3) + 8 (r >> 2) + 4 h.
Checks (all exact):
  1. chaining: the D tile of one product is the B operand of the next when k-step s' of a tile takes registers 8 s' .. + 7
     and the A operand is permuted to k <-> row_of(8 s' + j, hh);
  2. transpose by a 0/1 operand: D = H I puts feature f of sample row_of(r, h) in register r of lane (f, h);
  3. dW from two such feature-lane tiles, k-step ks = registers 8 ks .. + 7 of both;
  4. the same 0/1-operand product applied to a forward WEIGHT tile yields the A operand of the dH chain (lane = input
     row, k-step s' = output neurons row_of(8 s' + j, hh)): transposed weight images need not be stored.
usage: python tools/mfma_lane_maps.py"""
import numpy as np


def row_of(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def mfma_32x32x16(A, B, C):
    Am, Bm = np.zeros((32, 16)), np.zeros((16, 32))
    for lane in range(64):
        m, hh = lane & 31, lane >> 5
        Am[m, 8 * hh:8 * hh + 8] = A[lane]
        Bm[8 * hh:8 * hh + 8, m] = B[lane]
    Dm = Am @ Bm
    D = C.copy()
    for lane in range(64):
        n, h = lane & 31, lane >> 5
        for r in range(16):
            D[lane, r] += Dm[row_of(r, h), n]
    return D


def d_layout(val):
    """val[feature][sample] (32 x 32) -> registers of a D tile whose rows are features and columns samples"""
    t = np.zeros((64, 16))
    for lane in range(64):
        n, h = lane & 31, lane >> 5
        for r in range(16):
            t[lane, r] = val[row_of(r, h), n]
    return t


def feature_lane(val):
    o = np.zeros((64, 16))
    for lane in range(64):
        f, h = lane & 31, lane >> 5
        for r in range(16):
            o[lane, r] = val[f, row_of(r, h)]
    return o


def main():
    rng = np.random.default_rng(0)
    # 1. chaining: Z^T = W H^T for one 32-wide tile pair
    W, H = rng.integers(-3, 4, (32, 32)).astype(float), rng.integers(-3, 4, (32, 32)).astype(float)   # W[out][in], H[in][sample]
    t = d_layout(H)
    acc = np.zeros((64, 16))
    for sp in range(2):
        A = np.zeros((64, 8))
        for lane in range(64):
            m, hh = lane & 31, lane >> 5
            for j in range(8):
                A[lane, j] = W[m, row_of(8 * sp + j, hh)]
        acc = mfma_32x32x16(A, t[:, 8 * sp:8 * sp + 8], acc)
    assert np.array_equal(acc, d_layout(W @ H)), "chaining"
    # 2. transpose by a 0/1 operand
    val = rng.standard_normal((32, 32))
    t = d_layout(val)
    o = np.zeros((64, 16))
    for sp in range(2):
        ident = np.zeros((64, 8))
        for lane in range(64):
            n, hh = lane & 31, lane >> 5
            for j in range(8):
                ident[lane, j] = 1.0 if n == row_of(8 * sp + j, hh) else 0.0
        o = mfma_32x32x16(t[:, 8 * sp:8 * sp + 8], ident, o)
    assert np.array_equal(o, feature_lane(val)), "transpose"
    # 3. dW[fo][fi] = sum_n dz[fo][n] h[fi][n] from feature-lane tiles
    dz, hh_ = rng.integers(-3, 4, (32, 32)).astype(float), rng.integers(-3, 4, (32, 32)).astype(float)
    A, B = feature_lane(dz), feature_lane(hh_)
    acc = np.zeros((64, 16))
    for ks in range(2):
        acc = mfma_32x32x16(A[:, 8 * ks:8 * ks + 8], B[:, 8 * ks:8 * ks + 8], acc)
    ref = dz @ hh_.T
    want = np.zeros((64, 16))
    for lane in range(64):
        fi, h = lane & 31, lane >> 5
        for r in range(16):
            want[lane, r] = ref[row_of(r, h), fi]
    assert np.array_equal(acc, want), "dW"
    # 4. transposed weight tile from the forward A operands: dH^T = W^T dZ^T with A taken from registers 8 s' .. of D = W_A I
    W, dZ = rng.integers(-3, 4, (32, 32)).astype(float), rng.integers(-3, 4, (32, 32)).astype(float)  # W[out][in], dZ[out][sample]
    wt = np.zeros((64, 16))
    for sp in range(2):
        A = np.zeros((64, 8))
        ident = np.zeros((64, 8))
        for lane in range(64):
            m, hh = lane & 31, lane >> 5
            for j in range(8):
                A[lane, j] = W[m, row_of(8 * sp + j, hh)]                      # forward image, k-step sp
                ident[lane, j] = 1.0 if m == row_of(8 * sp + j, hh) else 0.0
        wt = mfma_32x32x16(A, ident, wt)                                       # lane = input neuron, registers = outputs
    t = d_layout(dZ)
    acc = np.zeros((64, 16))
    for sp in range(2):
        acc = mfma_32x32x16(wt[:, 8 * sp:8 * sp + 8], t[:, 8 * sp:8 * sp + 8], acc)
    assert np.array_equal(acc, d_layout(W.T @ dZ)), "weight transpose"
    print("lane-map checks passed: chaining, transpose by 0/1 operand, dW from feature-lane tiles, transposed weights")


if __name__ == "__main__":
    main()
