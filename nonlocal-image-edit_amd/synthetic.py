"""Synthetic luminance planes for benchmarks and full-size property tests (host, numpy).

SURVEY.md section 8d / BASELINE.md section 3: integer-valued 0..255 like the L channel of 8-bit Lab
(what `getLuminanceChannel`, reference src/filter.cpp:460-469, hands to trainFilter):
    L(r,c) = clip(round(128 + 70 s(r/H, c/W) + 40 (u - 1/2))),
    s(a,b) = 1/2 sin(2 pi (1.5a + 0.5b)) + 1/2 cos(2 pi (0.7a - 2.2b)),
    u      = splitmix64((r*W + c) xor seed) mapped to [0, 1).
Data generation only -- no part of the filter is computed here.
"""
from __future__ import annotations

import numpy as np


def _splitmix64(x: np.ndarray) -> np.ndarray:
    z = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def synthetic_luminance(H: int, W: int, seed: int = 1234, rows=None) -> np.ndarray:
    """H x W float64 plane; `rows=(r0, r1)` generates only that row slab (same values)."""
    r0, r1 = (0, H) if rows is None else rows
    r = np.arange(r0, r1, dtype=np.float64)[:, None] / H
    c = np.arange(W, dtype=np.float64)[None, :] / W
    s = 0.5 * np.sin(2 * np.pi * (1.5 * r + 0.5 * c)) + 0.5 * np.cos(2 * np.pi * (0.7 * r - 2.2 * c))
    idx = (np.arange(r0, r1, dtype=np.uint64)[:, None] * np.uint64(W)
           + np.arange(W, dtype=np.uint64)[None, :])
    with np.errstate(over="ignore"):
        h = _splitmix64(idx ^ np.uint64(seed))
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return np.clip(np.rint(128.0 + 70.0 * s + 40.0 * (u - 0.5)), 0, 255)


# BASELINE.json configs 2-5 (config 1 is the flower image, CPU plumbing only)
CONFIGS = {
    "cfg2": dict(H=512, W=512, n_row=10, n_col=20, hx=512 / 4, hy=30.0, T=10, K=10, L=4),
    "cfg3": dict(H=2048, W=2048, n_row=20, n_col=20, hx=2048 / 4, hy=30.0, T=50, K=50, L=4),
    "cfg4": dict(H=4096, W=4096, n_row=20, n_col=10, hx=4096 / 4, hy=30.0, T=10, K=50, L=4),
    "cfg5": dict(H=8192, W=8192, n_row=30, n_col=30, hx=8192 / 8, hy=30.0, T=10, K=100, L=6),
    # cfg5's sample grid, bandwidth ratio (hx = W / 8), K and L at the largest size whose N x 900 fp64 matrix a 62 GB
    # host holds (30 GB): the shape the CPU oracle can still pin (tests/golden/fullsize_cfg5_2k.npz)
    "cfg5_2k": dict(H=2048, W=2048, n_row=30, n_col=30, hx=2048 / 8, hy=30.0, T=10, K=100, L=6),
}
