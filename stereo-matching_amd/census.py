"""Weighted census transform (WCT) cost volume -- drop-in for the reference's census.py.

Same public functions, arguments, defaults, dtypes and error behaviour as
/root/reference/census.py; the arithmetic runs on the GPU through libstereo_hip.so.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

import _native


def generate_offsets(radius: int = 4) -> List[Tuple[int, int, int]]:
    """8 directions x distances 1..radius as (dy, dx, r)  (reference census.py:20-45)."""
    if radius <= 0:
        raise ValueError("radius must be a positive integer")
    directions: Sequence[Tuple[int, int]] = (
        (1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (1, -1), (-1, 1), (-1, -1),
    )
    return [(dy * r, dx * r, r) for dy, dx in directions for r in range(1, radius + 1)]


def compute_weights(offsets: Sequence[Tuple[int, int, int]], base_weight: float = 8.0) -> np.ndarray:
    """float32(base_weight / 2**(r-1)) per offset  (reference census.py:48-62)."""
    return np.array([base_weight / (2 ** (r - 1)) for _, _, r in offsets], dtype=np.float32)


def compute_wct_cost_volume(
    left: np.ndarray,
    right: np.ndarray,
    dmax: int,
    radius: int = 4,
    base_weight: float = 8.0,
    progress_callback: Optional[Callable[[int, int, str], None]] = None,
    parallel_chunk: int = 8,
) -> np.ndarray:
    """(H, W, dmax) float32 cost volume, disparity innermost  (reference census.py:205-269).

    Validation order and ValueError conditions follow the reference.  `parallel_chunk` only
    sets the granularity of the progress callbacks here: the whole volume is one kernel launch.
    """
    left = np.asarray(left)
    right = np.asarray(right)
    if left.ndim != 2 or right.ndim != 2:
        raise ValueError("left/right must be 2-D gray images")
    if left.shape != right.shape:
        raise ValueError("left/right image sizes differ")
    if dmax <= 0:
        raise ValueError("dmax must be a positive integer")
    if parallel_chunk <= 0:
        raise ValueError("parallel_chunk must be a positive integer")
    if radius <= 0:
        raise ValueError("radius must be a positive integer")
    if left.dtype != right.dtype:
        common = np.result_type(left.dtype, right.dtype)
        left, right = left.astype(common), right.astype(common)
    l, dt = _native.image_for_census(left)
    r, _ = _native.image_for_census(right)
    height, width = l.shape
    dsi = np.empty((height, width, int(dmax)), dtype=np.float32)
    if height > 0 and width > 0:
        _native.require_gpu()
        _native.check(_native.lib().stereo_wct_cost_volume_host(
            _native.ptr(l), _native.ptr(r), dt, height, width, int(dmax), int(radius), float(base_weight),
            _native.ptr(dsi)))
    if progress_callback is not None:
        for d_start in range(0, dmax, parallel_chunk):
            progress_callback(min(d_start + parallel_chunk, dmax), dmax, "WCT cost volume")
    return dsi
