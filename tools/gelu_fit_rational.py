"""GELU and GELU' from ONE exponential and ONE reciprocal (for the recomputing backward kernels, which need both):
    E = exp(-x^2/2),  t = 1/(1 + p|x|),  Phi(-|x|) ~= t P(t) E   (the Abramowitz-Stegun 7.1.26 form, refitted for fp32)
    gelu = x cdf,  gelu' = cdf + x E / sqrt(2 pi),  cdf = x < 0 ? Phi(-|x|) : 1 - Phi(-|x|)
Fits P (degree `deg`, Lawson-weighted least squares on the error of x Phi) for a grid of p, evaluates the fp32 instruction
sequence against float64.  CPU only.  usage: python tools/gelu_fit_rational.py [deg]"""
import sys
import numpy as np
from scipy.special import ndtr, log_ndtr

f = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f)


def fit(p, deg, n=40001, iters=80, xmax=9.0):
    x = 0.5 * xmax * (1 - np.cos(np.pi * (np.arange(n) + 0.5) / n))
    t = 1.0 / (1.0 + p * x)
    target = np.exp(log_ndtr(-x) + 0.5 * x * x)          # Phi(-x) / E, smooth, ~ 0.5 .. 1/(x sqrt(2 pi))
    V = np.stack([t ** (k + 1) for k in range(deg)], 1)  # q = t P(t), no constant term
    w0 = np.exp(-0.5 * x * x)                             # error of Phi = E dq  (gelu error = |x| times that)
    lw = np.ones_like(x)
    for _ in range(iters):
        w = w0 * lw
        c, *_ = np.linalg.lstsq(V * w[:, None], target * w, rcond=None)
        r = np.abs(V @ c - target) * w0
        lw = lw * (r / r.max() + 1e-3) ** 0.5
        lw /= lw.max()
    return c


def evaluate(p, c):
    x = np.concatenate([np.linspace(-12, 12, 2_000_001), np.linspace(-1.5, 1.5, 1_000_001)]).astype(f)
    x64 = x.astype(np.float64)
    true = x64 * ndtr(x64)
    tg = ndtr(x64) + x64 * np.exp(-0.5 * x64 * x64) / np.sqrt(2 * np.pi)
    ax = np.abs(x)
    E = np.exp2(((x * x).astype(f) * f(-0.72134752044448170368)).astype(f).astype(np.float64)).astype(f)
    t = (f(1) / fma(ax, np.full_like(x, f(p)), np.full_like(x, 1))).astype(f)
    c32 = c.astype(f)
    q = np.full_like(x, c32[-1])
    for k in c32[-2::-1]:
        q = fma(q, t, np.full_like(x, k))
    he = ((q * t).astype(f) * E).astype(f)
    cdf = np.where(x < 0, he, (f(1) - he).astype(f))
    h = (x * cdf).astype(f)
    gp = fma(x, (E * f(0.3989422804014327)).astype(f), cdf)
    sc = np.maximum(np.abs(x64), 1e-30)
    e = np.abs(h.astype(np.float64) - true)
    return e.max(), (e / sc).max(), np.abs(gp.astype(np.float64) - tg).max()


if __name__ == "__main__":
    deg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    best = None
    for p in np.linspace(0.15, 0.45, 31):
        c = fit(p, deg)
        r = evaluate(p, c)
        if best is None or r[1] < best[0][1]:
            best = (r, p, c)
        print("p %.3f: gelu max abs %.3e  err/|x| %.3e  gelu' abs %.3e" % (p, *r), flush=True)
    r, p, c = best
    print("BEST degree %d: p = %.9f  gelu err/|x| %.3e, abs %.3e, gelu' abs %.3e" % (deg, p, r[1], r[0], r[2]))
    print("coefficients of t^1..t^%d: " % deg + ", ".join("%.9ef" % v for v in c.astype(f)))
