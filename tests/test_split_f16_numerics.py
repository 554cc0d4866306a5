"""Host-side restatement (numpy) of the arithmetic of the two-piece fp16 MLP kernels (csrc/mlp_bwd_split_f16.hip, the f16 path of
csrc/mlp_device.h): every fp32 operand a = a0 + a1 with a0 = fp16(a) rounded TOWARD ZERO (v_cvt_pkrtz: the remainder a - a0 is
exact in fp32) and a1 = fp16(a - a0) rounded to nearest (v_fma_mixlo/hi_f16), products a0 b0 + a0 b1 + a1 b0 accumulated in fp32
(the parameter-gradient products keep a1 b1 too).  What the kernels rely on and what their guards are for is checked here without
a GPU; the GPU counterparts are tests/test_gpu_mlp.py::test_split_f16_backward_matches_float64 / ..._forward_against_float64 and
tests/test_gpu_hotpath_parity.py::test_cfg2_full_batch_dense_gradient_against_float64."""
import numpy as np


def f16_rtz(x):
    """fp32 -> fp16 rounded toward zero, subnormals kept, overflow saturating at the largest finite number (v_cvt_pkrtz)"""
    x = np.asarray(x, np.float32)
    with np.errstate(over="ignore"):
        h = x.astype(np.float16)
    too_big = np.abs(h.astype(np.float64)) > np.abs(x.astype(np.float64))
    h = np.where(too_big, np.nextafter(h, np.float16(0)), h)
    return h.astype(np.float16)


def split2(x):
    """(high, low) fp16 pieces of fp32 values, as the kernels form them"""
    x = np.asarray(x, np.float32)
    hi = f16_rtz(x)
    rem = x - hi.astype(np.float32)                 # exact in fp32 (checked below)
    return hi, rem.astype(np.float16)               # numpy's cast rounds to nearest even, like v_fma_mix


def prod3(a, b):
    """a0 b0 + a0 b1 + a1 b0 in float64 (every piece product is exact in fp32: 11 x 11 significant bits)"""
    a0, a1 = (p.astype(np.float64) for p in split2(a))
    b0, b1 = (p.astype(np.float64) for p in split2(b))
    return a1 * b0 + a0 * b1 + a0 * b0


def test_remainder_is_exact_and_pieces_reconstruct_to_22_bits():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200_000) * 10.0 ** rng.uniform(-4, 4, 200_000)).astype(np.float32)
    hi, lo = split2(x)
    rem64 = x.astype(np.float64) - hi.astype(np.float64)
    assert np.array_equal((x - hi.astype(np.float32)).astype(np.float64), rem64)        # the fp32 subtraction lost nothing
    assert np.all(np.abs(hi.astype(np.float64)) <= np.abs(x.astype(np.float64)))         # toward zero
    ok = np.abs(x) > 2.0 ** -3                      # low piece still a NORMAL fp16 number: 11 + 11 bits
    err = np.abs(hi.astype(np.float64) + lo.astype(np.float64) - x) / np.abs(x)
    assert err[ok].max() <= 2.0 ** -22
    # below 2^-3 the low piece is an fp16 SUBNORMAL: absolute precision 2^-25 (half the subnormal spacing), which is what makes
    # the scheme legitimate on gfx950 only because its matrix pipe honours subnormal inputs (attic/prototypes/mlp_fwd_split_f16.hip)
    small = np.abs(x) <= 2.0 ** -3
    assert np.abs(hi.astype(np.float64) + lo.astype(np.float64) - x)[small].max() <= 2.0 ** -25


def test_three_products_reach_fp32_level_two_do_not():
    rng = np.random.default_rng(1)
    a = rng.standard_normal(100_000).astype(np.float32)
    b = rng.standard_normal(100_000).astype(np.float32)
    exact = a.astype(np.float64) * b.astype(np.float64)
    e3 = np.abs(prod3(a, b) - exact) / np.abs(exact).max()
    a0, _ = split2(a)
    b0, b1 = split2(b)
    e2 = np.abs(a0.astype(np.float64) * (b0.astype(np.float64) + b1.astype(np.float64)) - exact) / np.abs(exact).max()
    assert e3.max() <= 2.0 ** -20 and e2.max() >= 2.0 ** -13          # dropping a1 b0 costs a thousand times more than a1 b1


def test_a_chain_on_the_mantissa_of_dy_is_as_accurate_for_small_gradients_as_for_large_ones():
    """dX[n] = W^T (dY[n] g): linear in dY[n].  On dY itself a sample whose gradient is 1e-6 of the batch's largest would have
    fp16-subnormal operands (2^-24 absolute precision: nothing left); on the MANTISSA of its dY, scaled into [2^4, 2^5) and
    multiplied back by 2^(e - 4) at the store (dy_parts in the kernel), every sample keeps 22 bits relative to itself."""
    rng = np.random.default_rng(2)
    w = rng.standard_normal(64).astype(np.float32) * 0.2
    g = rng.uniform(0.0, 1.1, 64).astype(np.float32)                   # gelu'
    for dy in (np.float32(1.0), np.float32(3e-7), np.float32(2e4)):
        exact = w.astype(np.float64) * (np.float64(dy) * g.astype(np.float64))
        naive = prod3(w, (dy * g).astype(np.float32))
        m, e = np.frexp(dy)                                            # dy = m 2^e, m in [0.5, 1)
        mant = np.float32(m * 32.0)                                    # [2^4, 2^5)
        scaled = prod3(w, (mant * g).astype(np.float32)) * 2.0 ** (int(e) - 5)
        rel = lambda v: (np.abs(v - exact) / np.abs(exact).max()).max()
        assert rel(scaled) <= 2.0 ** -20, (dy, rel(scaled))
        if dy < 1e-5:
            assert rel(naive) > 1e-3, (dy, rel(naive))                 # what the guard is for


def test_h_operand_needs_the_prescale_for_small_activations():
    """dW += dZ^T (H[n] 2^(e(n) - e_max)), e_max from the largest |dY| of the launch.  With encoding-like activations (1e-2), dY spread
    over six decades and ONE outlier that sets e_max, the H-side operand of most samples sits at or below fp16's subnormal spacing
    and their contributions lose their bits -- the truncation toward zero makes the loss systematic, not noise.  Round 4 measured
    1.2e-3 .. 1.7e-3 of dW on the GPU (test_cfg2_full_batch_dense_gradient_against_float64); this emulation gives 2.0e-3.
    Pre-scaled by 2^8 (H_PRESCALE_EXP, taken out again by the summing launch) the same sum is good to 1e-5 (here 7e-6, GPU
    <= 1.5e-5).  Same-signed terms, so that nothing cancels and the bias shows."""
    rng = np.random.default_rng(3)
    n = 200_000
    h = np.abs(rng.standard_normal(n) * 1e-2).astype(np.float32)
    dy = (rng.standard_normal(n) * 10.0 ** (-6.0 * rng.uniform(0, 1, n))).astype(np.float32)
    dy[0] = 1e3
    m, e = np.frexp(dy)                                                  # dy = m 2^e
    dz = (np.abs(m) * 32.0 * rng.uniform(0.05, 1.0, n)).astype(np.float32)   # |dZ| as the chain carries it: on mantissas in [16, 32)
    e2 = (e - e.max()).astype(np.int64)                                  # per-sample exponent relative to the launch maximum
    exact = np.sum(dz.astype(np.float64) * h.astype(np.float64) * 2.0 ** e2)

    def emulated(prescale):
        op = (h.astype(np.float64) * 2.0 ** (e2 + prescale)).astype(np.float32)      # exact power-of-two scaling
        a0, a1 = (p.astype(np.float64) for p in split2(dz))
        b0, b1 = (p.astype(np.float64) for p in split2(op))
        return np.sum(a0 * b0 + a0 * b1 + a1 * b0 + a1 * b1) * 2.0 ** -prescale      # the dW products keep all four

    assert abs(emulated(8) - exact) / exact <= 1e-5
    assert abs(emulated(0) - exact) / exact >= 1e-3
    # the limit the header states: an activation above 255 saturates its pre-scaled high piece
    assert float(f16_rtz(np.float32(300.0 * 2.0 ** 8))) == 65504.0
