"""ctypes front-end for the CPU oracle (oracle/stereo_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py, as the checker / the timed CPU baseline.
The product modules never import it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libstereo_oracle.so")
_lib: Optional[ctypes.CDLL] = None

F32, F64, U8 = 0, 1, 2
FILTER_GUIDED, FILTER_MEDIAN, FILTER_MEDIAN_U8Q, FILTER_GAUSSIAN, FILTER_BILATERAL = 0, 1, 2, 3, 4


def build(force: bool = False) -> str:
    """Compile the oracle with gcc if the shared object is missing or stale."""
    src = os.path.join(_HERE, "stereo_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libstereo_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _check(rc: int, what: str) -> None:
    if rc == -1:
        raise ValueError(f"oracle: invalid argument in {what}")
    if rc != 0:
        raise MemoryError(f"oracle: {what} failed with {rc}")


def max_threads() -> int:
    return int(lib().orc_max_threads())


def generate_offsets(radius: int) -> np.ndarray:
    if radius <= 0:
        raise ValueError("radius must be positive")
    out = np.zeros((8 * radius, 3), dtype=np.int32)
    n = lib().orc_generate_offsets(ctypes.c_int(radius), _ptr(out))
    assert n == 8 * radius
    return out


def compute_weights(offsets: np.ndarray, base_weight: float = 8.0) -> np.ndarray:
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    out = np.zeros(offsets.shape[0], dtype=np.float32)
    lib().orc_compute_weights(_ptr(offsets), ctypes.c_int(offsets.shape[0]), ctypes.c_double(base_weight), _ptr(out))
    return out


def large_value(weights: np.ndarray) -> float:
    f = lib().orc_large_value
    f.restype = ctypes.c_double
    weights = np.ascontiguousarray(weights, dtype=np.float32)
    return float(f(_ptr(weights), ctypes.c_int(weights.shape[0])))


def _image_dtype(a: np.ndarray) -> Tuple[np.ndarray, int]:
    if a.dtype == np.float32:
        return np.ascontiguousarray(a), F32
    if a.dtype == np.float64:
        return np.ascontiguousarray(a), F64
    if a.dtype == np.uint8:
        return np.ascontiguousarray(a), U8
    raise TypeError(f"oracle census supports float32/float64/uint8, got {a.dtype}")


def wct_cost_volume(left: np.ndarray, right: np.ndarray, dmax: int, radius: int = 4, base_weight: float = 8.0,
                    parallel_chunk: int = 8, threads: int = 8) -> np.ndarray:
    if left.ndim != 2 or right.ndim != 2 or left.shape != right.shape:
        raise ValueError("left/right must be 2-D and of equal shape")
    if left.dtype != right.dtype:
        raise TypeError("left/right dtype mismatch")
    l, dt = _image_dtype(left)
    r, _ = _image_dtype(right)
    h, w = l.shape
    if dmax <= 0:
        raise ValueError("dmax must be positive")
    dsi = np.empty((h, w, dmax), dtype=np.float32)
    rc = lib().orc_wct_cost_volume(_ptr(l), _ptr(r), ctypes.c_int(dt), ctypes.c_int(h), ctypes.c_int(w),
                                   ctypes.c_int(dmax), ctypes.c_int(radius), ctypes.c_double(base_weight),
                                   _ptr(dsi), ctypes.c_int(parallel_chunk), ctypes.c_int(threads))
    _check(rc, "wct_cost_volume")
    return dsi


def integral_image(image: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(image, dtype=np.float64)
    h, w = img.shape
    out = np.empty((h + 1, w + 1), dtype=np.float64)
    _check(lib().orc_integral_image(_ptr(img), ctypes.c_int(h), ctypes.c_int(w), _ptr(out)), "integral_image")
    return out


def box_filter_mean(image: np.ndarray, radius: int, threads: int = 8) -> np.ndarray:
    img = np.ascontiguousarray(image, dtype=np.float64)
    h, w = img.shape
    out = np.empty((h, w), dtype=np.float64)
    _check(lib().orc_box_filter_mean(_ptr(img), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(radius), _ptr(out),
                                     ctypes.c_int(threads)), "box_filter_mean")
    return out


def guided_prepare(guide: np.ndarray, radius: int, threads: int = 8):
    g = np.ascontiguousarray(guide, dtype=np.float64)
    h, w = g.shape
    mean_guide = np.empty((h, w), dtype=np.float64)
    var_g = np.empty((h, w), dtype=np.float64)
    _check(lib().orc_guided_prepare(_ptr(g), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(radius),
                                    _ptr(mean_guide), _ptr(var_g), ctypes.c_int(threads)), "guided_prepare")
    return g, mean_guide, var_g


def guided_filter(guide: np.ndarray, src: np.ndarray, radius: int, eps: float, threads: int = 8) -> np.ndarray:
    g, mean_guide, var_g = guided_prepare(guide, radius, threads)
    s = np.ascontiguousarray(src, dtype=np.float32)
    h, w = g.shape
    out = np.empty((h, w), dtype=np.float32)
    _check(lib().orc_guided_apply(_ptr(g), _ptr(mean_guide), _ptr(var_g), _ptr(s), ctypes.c_int(h), ctypes.c_int(w),
                                  ctypes.c_int(radius), ctypes.c_double(eps), _ptr(out), ctypes.c_int(threads)),
           "guided_apply")
    return out


def _filter2d(fn_name: str, image: np.ndarray, param, threads: int) -> np.ndarray:
    img = np.ascontiguousarray(image, dtype=np.float32)
    h, w = img.shape
    out = np.empty((h, w), dtype=np.float32)
    rc = getattr(lib(), fn_name)(_ptr(img), ctypes.c_int(h), ctypes.c_int(w), param, _ptr(out), ctypes.c_int(threads))
    _check(rc, fn_name)
    return out


def median_filter(image: np.ndarray, radius: int, threads: int = 8) -> np.ndarray:
    return _filter2d("orc_median_filter", image, ctypes.c_int(radius), threads)


def median_filter_u8q(image: np.ndarray, radius: int, threads: int = 8) -> np.ndarray:
    return _filter2d("orc_median_filter_u8q", image, ctypes.c_int(radius), threads)


def gaussian_filter(image: np.ndarray, sigma: float, threads: int = 8) -> np.ndarray:
    return _filter2d("orc_gaussian_filter", image, ctypes.c_double(sigma), threads)


def bilateral_filter(image: np.ndarray, sigma: float, threads: int = 8) -> np.ndarray:
    return _filter2d("orc_bilateral_filter", image, ctypes.c_double(sigma), threads)


def gaussian_kernel(radius: int, sigma: float) -> np.ndarray:
    k = np.empty((2 * radius + 1, 2 * radius + 1), dtype=np.float32)
    _check(lib().orc_gaussian_kernel(ctypes.c_int(radius), ctypes.c_double(sigma), _ptr(k)), "gaussian_kernel")
    return k


def aggregate_and_wta(dsi: np.ndarray, guide: np.ndarray, filter_kind: int, radius: int, eps_or_sigma: float,
                      want_layers: bool = False, threads: int = 8):
    d = np.ascontiguousarray(dsi, dtype=np.float32)
    h, w, dmax = d.shape
    g = np.ascontiguousarray(guide, dtype=np.float64)
    disp = np.empty((h, w), dtype=np.int32)
    cost = np.empty((h, w), dtype=np.float32)
    layers = np.empty((dmax, h, w), dtype=np.float32) if want_layers else None
    rc = lib().orc_aggregate_and_wta(_ptr(d), _ptr(g), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(dmax),
                                     ctypes.c_int(filter_kind), ctypes.c_int(radius), ctypes.c_double(eps_or_sigma),
                                     _ptr(disp), _ptr(cost), _ptr(layers) if want_layers else None,
                                     ctypes.c_int(threads))
    _check(rc, "aggregate_and_wta")
    return (disp, cost, layers) if want_layers else (disp, cost)


def compute_disparity(left: np.ndarray, right: np.ndarray, dmax: int, wct_radius: int = 4, base_weight: float = 8.0,
                      filter_kind: int = FILTER_GUIDED, radius: int = 3, eps_or_sigma: float = 0.0154,
                      threads: int = 8):
    l = np.ascontiguousarray(left, dtype=np.float32)
    r = np.ascontiguousarray(right, dtype=np.float32)
    h, w = l.shape
    disp = np.empty((h, w), dtype=np.int32)
    cost = np.empty((h, w), dtype=np.float32)
    rc = lib().orc_compute_disparity(_ptr(l), _ptr(r), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(dmax),
                                     ctypes.c_int(wct_radius), ctypes.c_double(base_weight), ctypes.c_int(filter_kind),
                                     ctypes.c_int(radius), ctypes.c_double(eps_or_sigma), _ptr(disp), _ptr(cost),
                                     ctypes.c_int(threads))
    _check(rc, "compute_disparity")
    return disp, cost


def compute_disparity_allcores(left: np.ndarray, right: np.ndarray, dmax: int, wct_radius: int = 4,
                               base_weight: float = 8.0, radius: int = 3, eps: float = 0.0154, threads: int = 0):
    """compute_disparity (guided filter) with the layers dealt to `threads` workers (0 = all the runtime offers):
    same arithmetic per layer, bit-identical result, not the reference's threading."""
    l = np.ascontiguousarray(left, dtype=np.float32)
    r = np.ascontiguousarray(right, dtype=np.float32)
    h, w = l.shape
    disp = np.empty((h, w), dtype=np.int32)
    cost = np.empty((h, w), dtype=np.float32)
    rc = lib().orc_compute_disparity_allcores(_ptr(l), _ptr(r), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(dmax),
                                              ctypes.c_int(wct_radius), ctypes.c_double(base_weight), ctypes.c_int(radius),
                                              ctypes.c_double(eps), _ptr(disp), _ptr(cost),
                                              ctypes.c_int(threads or max_threads()))
    _check(rc, "compute_disparity_allcores")
    return disp, cost


def compute_disparity_query(left: np.ndarray, right: np.ndarray, dmax: int, query_disp: np.ndarray, wct_radius: int = 4,
                            base_weight: float = 8.0, radius: int = 3, eps: float = 0.0154, threads: int = 0):
    """compute_disparity_allcores plus the oracle's own filtered cost at a queried disparity per pixel:
    returns (disparity, min_cost, F[y, x, query_disp[y, x]]).  For gap-aware comparisons at sizes where the
    dmax filtered layers cannot be kept."""
    l = np.ascontiguousarray(left, dtype=np.float32)
    r = np.ascontiguousarray(right, dtype=np.float32)
    h, w = l.shape
    q = np.ascontiguousarray(query_disp, dtype=np.int32)
    if q.shape != (h, w):
        raise ValueError("query_disp must have the image's shape")
    disp = np.empty((h, w), dtype=np.int32)
    cost = np.empty((h, w), dtype=np.float32)
    qcost = np.empty((h, w), dtype=np.float32)
    rc = lib().orc_compute_disparity_query(_ptr(l), _ptr(r), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(dmax),
                                           ctypes.c_int(wct_radius), ctypes.c_double(base_weight), ctypes.c_int(radius),
                                           ctypes.c_double(eps), _ptr(disp), _ptr(cost), _ptr(q), _ptr(qcost),
                                           ctypes.c_int(threads or max_threads()))
    _check(rc, "compute_disparity_query")
    return disp, cost, qcost


# ---- steps either side of the hot path (numpy restatements; test infrastructure only) ----------
def to_gray(image: np.ndarray, normalize: bool = False, max_value=None) -> np.ndarray:
    """Reference stereo_io.py:47-75 with stereo_io.py:22-44 for the scale: float32 products and
    sums in source order, division by float32(scale), clip to [0, 1]."""
    image = np.asarray(image)
    if image.ndim == 2:
        gray = image.astype(np.float32)
    elif image.ndim == 3 and image.shape[2] >= 3:
        rgb = image[..., :3].astype(np.float32)
        gray = (np.float32(0.299) * rgb[..., 0] + np.float32(0.587) * rgb[..., 1]) + np.float32(0.114) * rgb[..., 2]
    else:
        raise ValueError("unsupported image shape")
    gray = gray.astype(np.float32)
    if not normalize:
        return gray
    if max_value is not None:
        if max_value <= 0:
            raise ValueError("max_value must be positive")
        scale = float(max_value)
    elif np.issubdtype(image.dtype, np.integer):
        scale = float(np.iinfo(image.dtype).max)
    else:
        lo, hi = float(np.nanmin(image)), float(np.nanmax(image))
        if lo >= 0.0 and hi <= 1.0:
            scale = 1.0
        elif hi <= 0.0:
            raise ValueError("the image maximum must be positive")
        else:
            scale = hi
    return np.clip(gray / np.float32(scale), 0.0, 1.0).astype(np.float32)


def pbm_rms(disparity: np.ndarray, ground_truth: np.ndarray, valid_mask: np.ndarray, bad_threshold: float):
    """Reference stereo.py:560-577: (pbm percent, rms, valid pixel count)."""
    ok = np.isfinite(ground_truth) & valid_mask
    n = int(np.sum(ok))
    if n == 0:
        raise ValueError("no valid pixel")
    diff = np.abs(disparity.astype(np.float32) - ground_truth.astype(np.float32))[ok]
    return float(np.mean(diff > bad_threshold) * 100.0), float(np.sqrt(np.mean(diff ** 2))), n
