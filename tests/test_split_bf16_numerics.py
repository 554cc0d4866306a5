"""Host-side restatement (numpy) of the operand splitting used by the split-bf16 MLP forward (csrc/mlp_device.h: split3 /
split8 / split_mac): every fp32 value is cut into three bf16 pieces by truncating the running remainder.  Checked here,
without a GPU: the pieces reconstruct the value EXACTLY (for |x| >= 2^-100; below, to < 2^-132 absolute), and a dot product that keeps the six
products of relative size >= 2^-16 (fp32 accumulation) is as accurate as an fp32 dot product, while three products are
not.  The GPU counterpart is tests/test_gpu_mlp.py::test_split_bf16_forward_keeps_fp32_accuracy."""
import numpy as np


def split3(x):
    """three float32 arrays, each with at most 8 significant bits (the value of a bf16), summing to x"""
    r = x.astype(np.float32)
    pieces = []
    for _ in range(3):
        top = (r.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
        pieces.append(top)
        r = (r - top).astype(np.float32)        # exact: top shares sign/exponent and leading bits with r
    return pieces, r


def test_pieces_are_bf16_and_reconstruct_exactly():
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2 ** 32, size=2_000_000, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    x = x[np.isfinite(x)]                          # every finite bit pattern class: normals, denormals, +-0, huge, tiny
    edge = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 1.0 + 2.0 ** -23, np.finfo(np.float32).max, np.finfo(np.float32).tiny,
                     np.float32(1e-45), 3.0e38, -2.5e-39, 65504.0, 1e4, 1e-20], dtype=np.float32)
    x = np.concatenate([x, edge])
    (a1, a2, a3), rest = split3(x)
    for p in (a1, a2, a3):
        assert not np.any(p.view(np.uint32) & np.uint32(0xFFFF))          # representable as bf16
    total = a1.astype(np.float64) + a2.astype(np.float64) + a3.astype(np.float64)
    big = np.abs(x) >= 2.0 ** -100                 # remainders stay normal numbers: nothing is left after three pieces
    assert np.array_equal(rest[big], np.zeros_like(rest[big]))
    assert np.array_equal(total[big], x[big].astype(np.float64))           # exact reconstruction
    # below that the second / third remainder is a denormal (7 instead of 8 bits survive the truncation):
    # what is lost is < 2^-132 in absolute terms
    assert np.abs(total[~big] - x[~big].astype(np.float64)).max() < 2.0 ** -132


def _dots(K, n, terms, rng):
    a = (rng.standard_normal((n, K)) * rng.choice([1e-3, 1.0, 30.0], size=(n, 1))).astype(np.float32)
    b = rng.standard_normal((n, K)).astype(np.float32)
    (a1, a2, a3), _ = split3(a)
    (b1, b2, b3), _ = split3(b)
    order = [(a3, b1), (a2, b2), (a1, b3), (a2, b1), (a1, b2), (a1, b1)]    # smallest first, as split_mac issues them
    if terms == 3:
        order = order[3:]
    acc = np.zeros(n, dtype=np.float32)
    for k0 in range(0, K, 16):                                              # one MFMA = 16-deep k-step, fp32 accumulate
        for pa, pb in order:
            prod = (pa[:, k0:k0 + 16].astype(np.float64) * pb[:, k0:k0 + 16].astype(np.float64)).sum(axis=1)
            acc = (acc.astype(np.float64) + prod).astype(np.float32)
    exact = (a.astype(np.float64) * b.astype(np.float64)).sum(axis=1)
    fp32 = np.zeros(n, dtype=np.float32)
    for k in range(K):
        fp32 = (fp32 + a[:, k] * b[:, k]).astype(np.float32)
    scale = (np.abs(a.astype(np.float64)) * np.abs(b.astype(np.float64))).sum(axis=1)
    return np.abs(acc - exact) / scale, np.abs(fp32 - exact) / scale


def test_six_products_match_fp32_three_do_not():
    rng = np.random.default_rng(1)
    e6, e32 = _dots(64, 20000, 6, rng)
    e3, _ = _dots(64, 20000, 3, rng)
    assert e6.max() <= 2.0 ** -21                      # truncated tails: < 3 * 2^-24 per product, plus accumulation
    assert e6.max() <= 4.0 * max(e32.max(), 2.0 ** -24)
    assert e3.max() > 2.0 ** -18                       # three products leave 2^-16-sized terms out
