"""Guided image filter / box filter / integral image -- drop-in for the reference's guided_filter.py.

Same public names, arguments, dtypes and error behaviour as /root/reference/guided_filter.py,
plus `integral_image`, the public name the reference's README documents (README.md:212) for
its private `_integral_image_numba`.  All arithmetic runs on the GPU.
"""
from __future__ import annotations

import ctypes
import weakref
from dataclasses import dataclass

import numpy as np

import _native


def integral_image(image: np.ndarray) -> np.ndarray:
    """Zero-bordered (H+1, W+1) float64 summed-area table  (reference guided_filter.py:21-47)."""
    image = np.asarray(image)
    if image.ndim != 2:
        raise ValueError("image must be 2-D")
    img = np.ascontiguousarray(image, dtype=np.float64)
    h, w = img.shape
    out = np.zeros((h + 1, w + 1), dtype=np.float64)
    if h > 0 and w > 0:
        _native.require_gpu()
        _native.check(_native.lib().stereo_integral_image_host(_native.ptr(img), h, w, _native.ptr(out)))
    return out


_integral_image_numba = integral_image


@dataclass(frozen=True)
class BoxFilterIndex:
    """Window bounds and areas of a clipped box filter (reference guided_filter.py:50-60)."""

    y0: np.ndarray
    y1: np.ndarray
    x0: np.ndarray
    x1: np.ndarray
    area: np.ndarray


def _prepare_box_filter_index(height: int, width: int, radius: int) -> BoxFilterIndex:
    """Reference guided_filter.py:63-86."""
    if height <= 0 or width <= 0:
        raise ValueError("height and width must be positive integers")
    if radius < 0:
        raise ValueError("radius must be a non-negative integer")
    ys = np.arange(height)
    xs = np.arange(width)
    y0 = np.clip(ys - radius, 0, height - 1)
    y1 = np.clip(ys + radius, 0, height - 1)
    x0 = np.clip(xs - radius, 0, width - 1)
    x1 = np.clip(xs + radius, 0, width - 1)
    area = (y1 - y0 + 1)[:, None] * (x1 - x0 + 1)[None, :]
    return BoxFilterIndex(y0=y0, y1=y1, x0=x0, x1=x1, area=area.astype(np.float64))


@dataclass(frozen=True)
class GuidedFilterPrecomputed:
    """Per-scene guide statistics (reference guided_filter.py:89-99)."""

    guide_f: np.ndarray
    indices: BoxFilterIndex
    mean_guide: np.ndarray
    mean_gg: np.ndarray
    var_g: np.ndarray
    radius: int
    eps: float


def box_filter_mean(image: np.ndarray, radius: int) -> np.ndarray:
    """Clipped-window mean, float64  (reference guided_filter.py:141-152)."""
    image = np.asarray(image)
    indices = _prepare_box_filter_index(image.shape[0], image.shape[1], radius)
    return box_filter_mean_with_indices(image, indices)


def box_filter_mean_with_indices(image: np.ndarray, indices: BoxFilterIndex) -> np.ndarray:
    """Reference guided_filter.py:155-179: integral image, four-corner box sum, / area."""
    image = np.asarray(image)
    if image.ndim != 2:
        raise ValueError("image must be 2-D")
    img = np.ascontiguousarray(image, dtype=np.float64)
    h, w = img.shape
    y0 = np.ascontiguousarray(indices.y0, dtype=np.int32)
    y1 = np.ascontiguousarray(indices.y1, dtype=np.int32)
    x0 = np.ascontiguousarray(indices.x0, dtype=np.int32)
    x1 = np.ascontiguousarray(indices.x1, dtype=np.int32)
    area = np.ascontiguousarray(indices.area, dtype=np.float64)
    if y0.shape != (h,) or y1.shape != (h,) or x0.shape != (w,) or x1.shape != (w,) or area.shape != (h, w):
        raise ValueError("indices do not match the image shape")
    out = np.empty((h, w), dtype=np.float64)
    _native.require_gpu()
    _native.check(_native.lib().stereo_box_filter_mean_indexed_host(
        _native.ptr(img), h, w, _native.ptr(y0), _native.ptr(y1), _native.ptr(x0), _native.ptr(x1),
        _native.ptr(area), _native.ptr(out)))
    return out


def prepare_guided_filter(guide: np.ndarray, radius: int, eps: float) -> GuidedFilterPrecomputed:
    """Reference guided_filter.py:182-217."""
    guide = np.asarray(guide)
    if guide.ndim != 2:
        raise ValueError("guide must be 2-D")
    if radius <= 0:
        raise ValueError("radius must be a positive integer")
    if eps <= 0:
        raise ValueError("eps must be positive")
    guide_f = np.ascontiguousarray(guide.astype(np.float64))
    h, w = guide_f.shape
    indices = _prepare_box_filter_index(h, w, radius)
    mean_guide = np.empty((h, w), dtype=np.float64)
    mean_gg = np.empty((h, w), dtype=np.float64)
    var_g = np.empty((h, w), dtype=np.float64)
    _native.require_gpu()
    _native.check(_native.lib().stereo_guided_prepare_host(
        _native.ptr(guide_f), h, w, int(radius), _native.ptr(mean_guide), _native.ptr(mean_gg), _native.ptr(var_g)))
    return GuidedFilterPrecomputed(guide_f=guide_f, indices=indices, mean_guide=mean_guide, mean_gg=mean_gg,
                                   var_g=var_g, radius=radius, eps=float(eps))


# Device-side plans of live GuidedFilterPrecomputed objects, keyed by id(); a plan is destroyed when its object
# is collected (the object is a frozen dataclass, so the plan cannot live on it).
_plans: dict = {}


def _drop_plan(key: int) -> None:
    handle = _plans.pop(key, None)
    if handle is not None:
        try:
            _native.lib().stereo_guided_plan_destroy(handle)
        except Exception:   # noqa: BLE001 -- interpreter shutdown
            pass


def _plan_of(precomputed: GuidedFilterPrecomputed) -> ctypes.c_void_p:
    """The guide of `precomputed`, its box mean and 1 / (var + eps) on the device: uploaded and computed at the first
    call for this object and kept until the object dies, so that the reference's per-layer loop (stereo.py:143-150)
    moves one layer in and one out per call."""
    key = id(precomputed)
    handle = _plans.get(key)
    if handle is None:
        guide_f = np.ascontiguousarray(precomputed.guide_f, dtype=np.float64)
        h, w = guide_f.shape
        handle = ctypes.c_void_p()
        _native.require_gpu()
        _native.check(_native.lib().stereo_guided_plan_create(_native.ptr(guide_f), h, w, int(precomputed.radius),
                                                              float(precomputed.eps), ctypes.byref(handle)))
        _plans[key] = handle
        weakref.finalize(precomputed, _drop_plan, key)
    return handle


def guided_filter_with_precompute(precomputed: GuidedFilterPrecomputed, src: np.ndarray) -> np.ndarray:
    """Reference guided_filter.py:220-253; returns float32 (H, W)."""
    src = np.asarray(src)
    if src.ndim != 2:
        raise ValueError("src must be 2-D")
    if src.shape != precomputed.guide_f.shape:
        raise ValueError("src and guide sizes differ")
    src_f = np.ascontiguousarray(src, dtype=np.float32)
    h, w = src_f.shape
    out = np.empty((h, w), dtype=np.float32)
    if h == 0 or w == 0:
        return out
    _native.check(_native.lib().stereo_guided_plan_apply(_plan_of(precomputed), _native.ptr(src_f), _native.ptr(out)))
    return out


def guided_filter(guide: np.ndarray, src: np.ndarray, radius: int, eps: float) -> np.ndarray:
    """Reference guided_filter.py:256-275."""
    guide = np.asarray(guide)
    src = np.asarray(src)
    if guide.shape != src.shape:
        raise ValueError("guide and src sizes differ")
    if guide.ndim != 2:
        raise ValueError("guide must be 2-D")
    if radius <= 0:
        raise ValueError("radius must be a positive integer")
    if eps <= 0:
        raise ValueError("eps must be positive")
    guide_f = np.ascontiguousarray(guide.astype(np.float64))
    src_f = np.ascontiguousarray(src.astype(np.float32))
    h, w = src_f.shape
    out = np.empty((h, w), dtype=np.float32)
    _native.require_gpu()
    _native.check(_native.lib().stereo_guided_filter_host(
        _native.ptr(guide_f), _native.ptr(src_f), h, w, int(radius), float(eps), _native.ptr(out)))
    return out
