"""Deterministic synthetic rectified stereo pairs (NumPy only).

Used by the parity tests and by bench.py: the Middlebury images the reference
is normally run on are not available offline, so the harness builds pairs of
the same shape.  A random 8-bit texture is blurred with an integer 5x5 box,
perturbed with small integer noise and cut into horizontal bands of constant
disparity, so that ``right[y, x - d(y)] == left[y, x]`` wherever the shifted
pixel exists.  Gray values are ``uint8 / 255`` in float32, which mimics the
reference's ``to_gray(normalize=True)`` (stereo_io.py:47-75) including the many
exact ties 8-bit data produces under the census transform's strict ``>``.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def synth_texture_u8(height: int, width: int, rng: np.random.Generator) -> np.ndarray:
    """Random texture, 5x5 integer box blur, +-8 integer noise, clipped to uint8."""
    base = rng.integers(0, 256, size=(height + 4, width + 4), dtype=np.int64)
    csum = np.zeros((height + 5, width + 5), dtype=np.int64)
    csum[1:, 1:] = np.cumsum(np.cumsum(base, axis=0), axis=1)
    box = csum[5:, 5:] - csum[:-5, 5:] - csum[5:, :-5] + csum[:-5, :-5]
    blurred = box // 25
    noise = rng.integers(-8, 9, size=(height, width), dtype=np.int64)
    return np.clip(blurred + noise, 0, 255).astype(np.uint8)


def synth_pair_u8(
    height: int, width: int, dmax: int, seed: int, bands: int = 8
) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Return (left_u8, right_u8, true_disparity_per_row)."""
    rng = np.random.default_rng(seed)
    tex = synth_texture_u8(height, width + dmax, rng)
    lo = max(dmax // 10, 0)
    hi = max(9 * dmax // 10, lo + 1)
    band_d = rng.integers(lo, hi, size=bands)
    rows = np.arange(height)
    band_of_row = np.minimum(rows * bands // max(height, 1), bands - 1)
    d_row = band_d[band_of_row].astype(np.int64)
    left = tex[:, :width]
    cols = np.arange(width)[None, :] + d_row[:, None]
    right = np.take_along_axis(tex, cols, axis=1)
    return np.ascontiguousarray(left), np.ascontiguousarray(right), d_row


def synth_pair_gray(
    height: int, width: int, dmax: int, seed: int, bands: int = 8
) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Float32 gray pair in [0, 1] (uint8 / 255), plus the per-row true disparity."""
    left, right, d_row = synth_pair_u8(height, width, dmax, seed, bands)
    scale = np.float32(255.0)
    return left.astype(np.float32) / scale, right.astype(np.float32) / scale, d_row
