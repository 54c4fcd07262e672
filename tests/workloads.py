"""Seeded synthetic inputs of SURVEY.md section 8d (C1..C5 stand-ins).

The C3 axis order (x fastest, z slowest) is the one that reproduces the survey's
recorded 256^3 known answer (cnt = 1 236 692) exactly.
"""
import numpy as np


def c1():
    """2^20 uniform[0,1) doubles, default_rng(12345)."""
    return np.random.default_rng(12345).random(1 << 20)


def c2(seed=2024):
    """CESM-ATM stand-in: smooth 1800x3600 fp32 field + 1 % noise."""
    x = np.linspace(0, 1, 3600)
    y = np.linspace(0, 1, 1800)
    yy, xx = np.meshgrid(y, x, indexing="ij")
    f = (np.sin(6 * np.pi * xx) * np.cos(4 * np.pi * yy) + 0.3 * np.sin(40 * np.pi * xx * yy)
         + 0.01 * np.random.default_rng(seed).standard_normal((1800, 3600)))
    return f.astype(np.float32).ravel()


def c3(n=512, seed=512, dtype=np.float64):
    """Synthetic n^3 volume (C3 formula); seed 512+g gives C4's shard g."""
    l = np.linspace(0, 1, n)
    out = np.empty((n, n, n), np.float64)
    rng = np.random.default_rng(seed)
    y, x = np.meshgrid(l, l, indexing="ij")
    sx, cy = np.sin(4 * np.pi * x), np.cos(6 * np.pi * y)
    for k in range(n):  # z slowest; plane at a time keeps memory bounded
        z = l[k]
        out[k] = 37.5 * (sx * cy * np.sin(2 * np.pi * z) + 0.2 * np.sin(30 * np.pi * x * y * z)
                         + 1e-3 * rng.standard_normal((n, n)))
    return out.astype(dtype, copy=False).ravel()


# tests/list-msst19.txt:1-6 lengths (fp64, 1-D); data itself is not available offline
MSST19_LENGTHS = (31040, 32768, 12960, 12960, 16384, 37024)


def c5_fp64(length, seed):
    """Smooth + noise stand-in with an MSST19 length."""
    t = np.linspace(0, 1, length)
    rng = np.random.default_rng(seed)
    return (3.0 * np.sin(14 * np.pi * t) + np.cos(90 * np.pi * t * t) + 0.02 * rng.standard_normal(length)) * 41.0


def ragged(n, dtype, seed=7, scale=3.7):
    """Generic small case: smooth + noise, arbitrary length (remainder blocks)."""
    t = np.arange(n) / 97.0
    rng = np.random.default_rng(seed + n)
    v = scale * (np.sin(t) + 0.3 * np.cos(5.1 * t) + 0.05 * rng.standard_normal(n))
    return v.astype(dtype)
