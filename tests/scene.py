"""Shared seeded synthetic inputs for the sampling / compositing tests (numpy, CPU)."""
import numpy as np


def make_rays(R, seed=0, radius=1.5, jitter_target=0.35):
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(R, 3)).astype(np.float32)
    o = (o / np.linalg.norm(o, axis=1, keepdims=True) * radius).astype(np.float32)
    target = (rng.uniform(-jitter_target, jitter_target, size=(R, 3))).astype(np.float32)
    d = target - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return np.ascontiguousarray(o), np.ascontiguousarray(d)


def shell_occupancy(oracle, n, extent=1.0, tr=(0, 0, 0), r0=0.3, width=0.05, seed=1, drop=0.1):
    """occupancy = voxels whose centre lies in a spherical shell, with a few random holes; Morton order."""
    pts = oracle.grid_points(n, extent, tr)
    rad = np.linalg.norm(pts, axis=1)
    occ = np.abs(rad - r0) < width
    rng = np.random.default_rng(seed)
    occ &= rng.uniform(size=occ.shape) > drop
    return np.ascontiguousarray(occ)


def analytic_sdf(pos, r0=0.3):
    return (np.linalg.norm(pos, axis=1, keepdims=True) - r0).astype(np.float32)
