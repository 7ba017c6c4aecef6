"""Aggregation + winner-take-all and the end-to-end pipeline -- drop-in for the hot-path
functions of the reference's stereo.py (aggregate_and_wta, compute_disparity).

Signatures, defaults, dtypes (int32 disparity, float32 min_cost), validation and the progress
callback contract follow /root/reference/stereo.py:94-245.  compute_disparity runs the fused
GPU path: the (H, W, D) cost volume is never materialised.
"""
from __future__ import annotations

import sys
from typing import Callable, Optional, Tuple

import numpy as np

import _native
from census import compute_wct_cost_volume  # noqa: F401  (re-exported like the reference)
from filters import bilateral_filter, gaussian_filter, median_filter  # noqa: F401
from guided_filter import GuidedFilterPrecomputed, guided_filter_with_precompute, prepare_guided_filter  # noqa: F401
from stereo_io import ensure_same_shape

DEFAULT_WCT_RADIUS: int = 4
DEFAULT_BASE_WEIGHT: float = 8.0
DEFAULT_GUIDED_RADIUS: int = 3
DEFAULT_GUIDED_EPS: float = 0.01
DEFAULT_FILTER_TYPE: str = "guided"
DEFAULT_MEDIAN_RADIUS: int = 3
DEFAULT_MEDIAN_METHOD: str = "opencv"
DEFAULT_MEDIAN_BLOCK_ROWS: int = 128
DEFAULT_GAUSSIAN_SIGMA: float = 1.0
DEFAULT_BILATERAL_SIGMA: float = 1.0
DEFAULT_BAD_THRESHOLD: float = 2.0

_SUPPORTED = ("guided", "median", "gaussian", "bilateral")
_LABELS = {"guided": "Guided Filter", "median": "Median Filter", "gaussian": "Gaussian Filter",
           "bilateral": "Bilateral Filter"}


def _print_progress(current: int, total: int, label: str) -> None:
    """Reference stereo.py:33-53."""
    if total <= 0:
        raise ValueError("total must be a positive integer")
    clamped = min(max(current, 0), total)
    percent = (clamped / float(total)) * 100.0
    message = f"{label}: {clamped}/{total} ({percent:5.1f}%)"
    sys.stdout.write(f"\r{message}\n" if clamped >= total else f"\r{message}")
    sys.stdout.flush()


def _filter_params(filter_key: str, wct_radius: int, base_weight: float, guided_radius: int, guided_eps: float,
                   median_radius: int, median_method: str, median_block_rows: int, gaussian_sigma: float,
                   bilateral_sigma: float) -> _native.StereoParams:
    """Validate the active filter's arguments as the reference does at its first layer."""
    kind = _native.FILTER_GUIDED
    if filter_key == "guided":
        if guided_radius <= 0:
            raise ValueError("radius must be a positive integer")
        if guided_eps <= 0:
            raise ValueError("eps must be positive")
    elif filter_key == "median":
        if median_radius <= 0:
            raise ValueError("radius must be a positive integer")
        method_key = median_method.strip().lower()
        if method_key not in ("opencv", "scipy", "vectorized", "naive"):
            raise ValueError("method must be 'opencv', 'scipy', 'vectorized' or 'naive'")
        if method_key == "vectorized" and median_block_rows <= 0:
            raise ValueError("block_rows must be a positive integer")
        kind = _native.FILTER_MEDIAN_U8Q if method_key == "opencv" else _native.FILTER_MEDIAN
    elif filter_key == "gaussian":
        if gaussian_sigma <= 0:
            raise ValueError("sigma must be positive")
        kind = _native.FILTER_GAUSSIAN
    else:
        if bilateral_sigma <= 0:
            raise ValueError("sigma must be positive")
        kind = _native.FILTER_BILATERAL
    return _native.StereoParams(
        wct_radius=int(wct_radius), base_weight=float(base_weight), filter_kind=kind,
        guided_radius=int(guided_radius) if filter_key == "guided" else 1,
        guided_eps=float(guided_eps) if filter_key == "guided" else 1.0,
        median_radius=int(median_radius) if filter_key == "median" else 1,
        gaussian_sigma=float(gaussian_sigma) if filter_key == "gaussian" else 1.0,
        bilateral_sigma=float(bilateral_sigma) if filter_key == "bilateral" else 1.0,
    )


def aggregate_and_wta(
    dsi: np.ndarray,
    guide: np.ndarray,
    guided_radius: int,
    guided_eps: float,
    filter_type: str = DEFAULT_FILTER_TYPE,
    median_radius: int = DEFAULT_MEDIAN_RADIUS,
    median_method: str = DEFAULT_MEDIAN_METHOD,
    median_block_rows: int = DEFAULT_MEDIAN_BLOCK_ROWS,
    gaussian_sigma: float = DEFAULT_GAUSSIAN_SIGMA,
    bilateral_sigma: float = DEFAULT_BILATERAL_SIGMA,
    progress_callback: Optional[Callable[[int, int, str], None]] = None,
) -> Tuple[np.ndarray, np.ndarray]:
    """Filter every disparity layer and keep the running arg-min  (reference stereo.py:94-174).

    Returns (disparity int32 (H, W), min_cost float32 (H, W)).  Ties keep the lower disparity,
    NaN never wins, pixels no layer improves stay (0, +inf).
    """
    dsi = np.asarray(dsi)
    guide = np.asarray(guide)
    if dsi.ndim != 3:
        raise ValueError("dsi must be 3-D (H, W, D)")
    if guide.ndim != 2:
        raise ValueError("guide must be a 2-D gray image")
    if dsi.shape[0] != guide.shape[0] or dsi.shape[1] != guide.shape[1]:
        raise ValueError("dsi and guide sizes differ")
    height, width, dmax = dsi.shape
    filter_key = filter_type.strip().lower()
    if filter_key not in _SUPPORTED:
        raise ValueError(f"filter_type must be one of {_SUPPORTED}")
    min_cost = np.full((height, width), np.inf, dtype=np.float32)
    disparity = np.zeros((height, width), dtype=np.int32)
    if filter_key == "guided" or dmax > 0:
        params = _filter_params(filter_key, 1, 8.0, guided_radius, guided_eps, median_radius, median_method,
                                median_block_rows, gaussian_sigma, bilateral_sigma)
    if dmax == 0 or height == 0 or width == 0:
        return disparity, min_cost
    dsi_f = np.ascontiguousarray(dsi, dtype=np.float32)
    if guide.dtype == np.float32:
        g, gdt = np.ascontiguousarray(guide), _native.DT_F32
    else:
        g, gdt = np.ascontiguousarray(guide.astype(np.float64)), _native.DT_F64
    _native.require_gpu()
    _native.check(_native.lib().stereo_aggregate_wta_host(
        _native.ptr(dsi_f), _native.ptr(g), gdt, height, width, dmax, params, 0, dmax,
        _native.ptr(disparity), _native.ptr(min_cost)))
    if progress_callback is not None:
        for d in range(dmax):
            progress_callback(d + 1, dmax, _LABELS[filter_key])
    return disparity, min_cost


def compute_disparity(
    left_gray: np.ndarray,
    right_gray: np.ndarray,
    dmax: int,
    wct_radius: int = DEFAULT_WCT_RADIUS,
    base_weight: float = DEFAULT_BASE_WEIGHT,
    guided_radius: int = DEFAULT_GUIDED_RADIUS,
    guided_eps: float = DEFAULT_GUIDED_EPS,
    filter_type: str = DEFAULT_FILTER_TYPE,
    median_radius: int = DEFAULT_MEDIAN_RADIUS,
    median_method: str = DEFAULT_MEDIAN_METHOD,
    median_block_rows: int = DEFAULT_MEDIAN_BLOCK_ROWS,
    gaussian_sigma: float = DEFAULT_GAUSSIAN_SIGMA,
    bilateral_sigma: float = DEFAULT_BILATERAL_SIGMA,
    show_progress: bool = True,
    progress_callback: Optional[Callable[[int, int, str], None]] = None,
    d_begin: int = 0,
    d_end: Optional[int] = None,
) -> Tuple[np.ndarray, np.ndarray]:
    """WCT cost -> filter -> WTA in one fused GPU pass  (reference stereo.py:177-245).

    `d_begin` / `d_end` (not in the reference) restrict the search to a disparity shard; the
    default is the whole range [0, dmax).
    """
    left_gray = np.asarray(left_gray)
    right_gray = np.asarray(right_gray)
    height, width = ensure_same_shape(left_gray, right_gray)
    if progress_callback is not None:
        progress = progress_callback
    elif show_progress:
        progress = _print_progress
    else:
        progress = None
    if dmax <= 0:
        raise ValueError("dmax must be a positive integer")
    if wct_radius <= 0:
        raise ValueError("radius must be a positive integer")
    if progress is not None:  # the cost stage reports first, as compute_wct_cost_volume does
        for d_start in range(0, dmax, 8):
            progress(min(d_start + 8, dmax), dmax, "WCT cost volume")
    filter_key = filter_type.strip().lower()
    if filter_key not in _SUPPORTED:
        raise ValueError(f"filter_type must be one of {_SUPPORTED}")
    params = _filter_params(filter_key, wct_radius, base_weight, guided_radius, guided_eps, median_radius,
                            median_method, median_block_rows, gaussian_sigma, bilateral_sigma)
    if d_end is None:
        d_end = dmax
    if left_gray.dtype != right_gray.dtype:
        common = np.result_type(left_gray.dtype, right_gray.dtype)
        left_gray, right_gray = left_gray.astype(common), right_gray.astype(common)
    l, dt = _native.image_for_census(left_gray)
    r, _ = _native.image_for_census(right_gray)
    min_cost = np.full((height, width), np.inf, dtype=np.float32)
    disparity = np.zeros((height, width), dtype=np.int32)
    if height > 0 and width > 0:
        _native.require_gpu()
        _native.check(_native.lib().stereo_compute_disparity_host(
            _native.ptr(l), _native.ptr(r), dt, height, width, int(dmax), params, int(d_begin), int(d_end),
            _native.ptr(disparity), _native.ptr(min_cost)))
    if progress is not None:
        for d in range(dmax):
            progress(d + 1, dmax, _LABELS[filter_key])
    return disparity, min_cost
