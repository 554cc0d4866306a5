"""Fit and check a cheaper GELU for the MLP kernels (the forward is VALU-issue bound on GELU since it moved to the bf16
matrix pipe; attic/prototypes/mlp_fwd_split_bf16.hip).  Identity used:

    gelu(x) = x Phi(x) = max(x, 0) - |x| Phi(-|x|),          Phi(-t) = exp2(P(t)),  t = |x|

so one polynomial P (Horner, `deg` FMAs), one v_exp_f32, one v_max_f32 and one FMA give the activation: deg + 3
instructions against ~24 for 0.5 x (1 + erf(x / sqrt 2)) with a < 1-ulp erf.  What 0.5 x (1 + erf) evaluated in fp32
(the reference: torch.nn.GELU) can resolve is an ABSOLUTE error of about |x| * 6e-8, which is all the fit needs to match.
The script fits P in float64 (Lawson-weighted least squares on the absolute error of exp2(P)), rounds the
coefficients to fp32, then evaluates the fp32 instruction sequence (FMA emulated in float64 and rounded once; exp2 rounded
to fp32, i.e. the hardware's 1 ulp is not included) on a dense grid and reports the error against float64 GELU next to the
error of the fp32 erf formula.  CPU only.
usage: python tools/gelu_fit.py [degree ...]"""
import sys

import numpy as np
from scipy.special import erfc, log_ndtr, ndtr

T_MAX = 5.75      # Phi(-5.75) = 4.5e-9: beyond, |x| Phi(-|x|) < 2.6e-8 and the extrapolated polynomial only goes down


def target(t):
    return log_ndtr(-t) / np.log(2.0)


def fit(deg, n=20001, iters=60):
    t = 0.5 * T_MAX * (1 - np.cos(np.pi * (np.arange(n) + 0.5) / n))      # Chebyshev nodes on [0, T_MAX]
    f = target(t)
    e = ndtr(-t)
    V = np.polynomial.chebyshev.chebvander(2 * t / T_MAX - 1, deg)
    w = e.copy()                      # d exp2(P) = e ln2 dP: weight the residual of P by e to get absolute error of e
    lw = np.ones_like(t)
    for _ in range(iters):
        ww = w * lw
        c, *_ = np.linalg.lstsq(V * ww[:, None], f * ww, rcond=None)
        r = np.abs(V @ c - f) * w
        lw = lw * (r / r.max() + 1e-3) ** 0.5          # Lawson: push weight to where the error is largest
        lw /= lw.max()
    p = np.polynomial.chebyshev.Chebyshev(c, domain=[0, T_MAX]).convert(kind=np.polynomial.Polynomial)
    return p.coef                                      # ascending powers of t


def f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32)


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def phi_neg(t, coef32):
    """e = Phi(-t) = exp2(P(t)) in fp32, t >= 0 already clamped"""
    p = np.full_like(t, coef32[-1])
    for c in coef32[-2::-1]:
        p = fma32(p, t, np.full_like(t, c))
    return np.exp2(p.astype(np.float64)).astype(np.float32)


def gelu_new(x, coef32):
    """v_min (|x| modifier), deg x v_fma, v_exp, v_max, v_fma: t is clamped in the product too, so +-inf stay finite/inf"""
    t = np.minimum(np.abs(x), np.float32(T_MAX))
    e = phi_neg(t, coef32)
    return fma32(-t, e, np.maximum(x, np.float32(0)))


def gelu_grad_new(x, coef32):
    """gelu'(x) = Phi(x) + x phi(x) from the same e: Phi(x) = x < 0 ? e : 1 - e; phi(x) = exp2(-x^2 log2(e)/2)/sqrt(2 pi)"""
    t = np.minimum(np.abs(x), np.float32(T_MAX))
    e = phi_neg(t, coef32)
    cdf = np.where(x < 0, e, (np.float32(1) - e).astype(np.float32))
    q = (t * t).astype(np.float32)
    pdf = np.exp2((q * np.float32(-0.72134752044448170368)).astype(np.float32).astype(np.float64)).astype(np.float32)
    pdf = (pdf * np.float32(0.3989422804014327)).astype(np.float32)
    xc = np.clip(x, np.float32(-T_MAX), np.float32(T_MAX))
    return fma32(xc, pdf, cdf)


def gelu_erf32(x):
    """0.5 x (1 + erf(x / sqrt 2)) with every operation rounded to fp32 and a correctly rounded erf (best case)"""
    from scipy.special import erf
    a = (x * np.float32(0.70710678118654752440)).astype(np.float32)
    e = erf(a.astype(np.float64)).astype(np.float32)
    return ((np.float32(0.5) * x).astype(np.float32) * (np.float32(1) + e).astype(np.float32)).astype(np.float32)


def main():
    degs = [int(a) for a in sys.argv[1:]] or [6, 7, 8, 9]
    x = np.concatenate([np.linspace(-12, 12, 4_000_001), np.linspace(-1, 1, 2_000_001), np.linspace(-1e-3, 1e-3, 200_001)]).astype(np.float32)
    x64 = x.astype(np.float64)
    true = x64 * ndtr(x64)
    ref_err = np.abs(gelu_erf32(x).astype(np.float64) - true)
    scale = np.maximum(np.abs(x64), 1e-30)
    print("fp32 erf formula (reference arithmetic): max abs err %.3e, max err / |x| %.3e" % (ref_err.max(), (ref_err / scale).max()))
    tg = x64 * np.exp(-0.5 * x64 * x64) / np.sqrt(2 * np.pi) + ndtr(x64)
    for deg in degs:
        coef = fit(deg)
        c32 = f32(coef)
        err = np.abs(gelu_new(x, c32).astype(np.float64) - true)
        k = int(np.argmax(err / scale))
        gerr = np.abs(gelu_grad_new(x, c32).astype(np.float64) - tg)
        far = np.linspace(T_MAX, 60.0, 200001)
        mono = np.all(np.diff(np.polynomial.polynomial.polyval(far, c32.astype(np.float64))) < 0)
        print("degree %d (%d instructions): gelu max abs err %.3e, max err / |x| %.3e at x = %.4f;  gelu' max abs err %.3e;  "
              "P decreasing beyond T_MAX: %s" % (deg, deg + 4, err.max(), (err / scale).max(), x[k], gerr.max(), mono))
        print("   coefficients (ascending): " + ", ".join("%.9ef" % v for v in c32))


if __name__ == "__main__":
    main()
