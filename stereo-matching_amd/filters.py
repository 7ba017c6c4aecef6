"""Median / Gaussian / bilateral aggregation filters -- drop-in for the reference's filters.py.

Same public functions, arguments and error behaviour as /root/reference/filters.py.  The
`opencv` methods do not call OpenCV: the median reproduces the reference's own uint8
quantisation (filters.py:139-166) followed by an exact uint8 median with replicated borders and
the reference's rescale (filters.py:177-180); the Gaussian evaluates the reference's 2-D
float32 kernel with reflect-101 borders (filters.py:191-204).  Both are checked against the
reference's own formulas; against OpenCV itself (cv2.medianBlur, cv2.GaussianBlur with its
separable kernel) they are not, because OpenCV is not available where this was built and the
reference holds no test or fixture for those calls.
"""
from __future__ import annotations

import math

import numpy as np

import _native


def _validate_2d(image: np.ndarray, name: str) -> None:
    if image.ndim != 2:
        raise ValueError(f"{name} must be 2-D")


def _infer_radius_from_sigma(sigma: float) -> int:
    """Reference filters.py:42-47."""
    if sigma <= 0:
        raise ValueError("sigma must be positive")
    return max(int(math.ceil(3.0 * sigma)), 1)


def _f32_image(image: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(image, dtype=np.float32)


def median_filter(image: np.ndarray, radius: int, method: str = "opencv", block_rows: int = 128) -> np.ndarray:
    """Reference filters.py:227-256.  Returns float32 (H, W)."""
    image = np.asarray(image)
    _validate_2d(image, "image")
    if radius <= 0:
        raise ValueError("radius must be a positive integer")
    method_key = method.strip().lower()
    if method_key not in ("opencv", "scipy", "vectorized", "naive"):
        raise ValueError("method must be 'opencv', 'scipy', 'vectorized' or 'naive'")
    if method_key == "vectorized" and block_rows <= 0:
        raise ValueError("block_rows must be a positive integer")
    h, w = image.shape
    out = np.empty((h, w), dtype=np.float32)
    if h == 0 or w == 0:
        return out
    _native.require_gpu()
    lib = _native.lib()
    if method_key == "opencv":
        if image.dtype == np.uint8:
            img8 = np.ascontiguousarray(image)
            _native.check(lib.stereo_median_filter_u8_host(_native.ptr(img8), h, w, int(radius), _native.ptr(out)))
        else:
            img = _f32_image(image)
            _native.check(lib.stereo_median_filter_host(_native.ptr(img), h, w, int(radius),
                                                        _native.FILTER_MEDIAN_U8Q, _native.ptr(out)))
        return out
    # scipy / vectorized / naive: exact median of the window with edge padding.  Casting to
    # float32 first is exact here: the median of an odd window is one of its elements and the
    # cast is monotonic, so cast(median(x)) == median(cast(x)).
    img = _f32_image(image)
    _native.check(lib.stereo_median_filter_host(_native.ptr(img), h, w, int(radius), _native.FILTER_MEDIAN,
                                                _native.ptr(out)))
    return out


def gaussian_filter(image: np.ndarray, sigma: float, method: str = "opencv") -> np.ndarray:
    """Reference filters.py:259-276.  Returns float32 (H, W)."""
    image = np.asarray(image)
    _validate_2d(image, "image")
    method_key = method.strip().lower()
    if method_key not in ("opencv", "naive"):
        raise ValueError("method must be 'opencv' or 'naive'")
    _infer_radius_from_sigma(sigma)
    img = _f32_image(image)
    h, w = img.shape
    out = np.empty((h, w), dtype=np.float32)
    if h == 0 or w == 0:
        return out
    _native.require_gpu()
    _native.check(_native.lib().stereo_gaussian_filter_host(_native.ptr(img), h, w, float(sigma), _native.ptr(out)))
    return out


def bilateral_filter(image: np.ndarray, sigma: float) -> np.ndarray:
    """Reference filters.py:279-310.  Returns float32 (H, W)."""
    image = np.asarray(image)
    _validate_2d(image, "image")
    _infer_radius_from_sigma(sigma)
    img = _f32_image(image)
    h, w = img.shape
    out = np.empty((h, w), dtype=np.float32)
    if h == 0 or w == 0:
        return out
    _native.require_gpu()
    _native.check(_native.lib().stereo_bilateral_filter_host(_native.ptr(img), h, w, float(sigma), _native.ptr(out)))
    return out
