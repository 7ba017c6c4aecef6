"""Image, ground-truth and result files around the hot path -- drop-in for the reference's
stereo_io.py (same function names, arguments, return dtypes and ValueError conditions).

File parsing is host work; the per-pixel arithmetic (gray conversion, normalisation, the
nanmin/nanmax that picks the normalisation scale of float images) runs on the GPU through the
C ABI and reproduces the reference's float32 rounding sequence bit for bit.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Tuple

import numpy as np

import _native

_DEVICE_DTYPES = {np.dtype(np.uint8): _native.DT_U8, np.dtype(np.uint16): _native.DT_U16,
                  np.dtype(np.float32): _native.DT_F32}


def read_image(path: str) -> np.ndarray:
    """Reference stereo_io.py:9-19: decode with Pillow, keep channels and dtype."""
    from PIL import Image

    with Image.open(path) as image:
        return np.array(image)


def _nan_minmax(values: np.ndarray) -> Tuple[float, float]:
    v = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
    lo, hi = ctypes.c_float(0.0), ctypes.c_float(0.0)
    _native.require_gpu()
    _native.check(_native.lib().stereo_nan_minmax_host(_native.ptr(v), v.size, ctypes.byref(lo), ctypes.byref(hi)))
    if lo.value > hi.value:  # empty or all-NaN: numpy's nanmin answers NaN (with a warning)
        return float("nan"), float("nan")
    return float(lo.value), float(hi.value)


def _resolve_normalization_scale(image: np.ndarray, max_value: Optional[float]) -> float:
    """Reference stereo_io.py:22-44: explicit max_value, else the integer type's maximum, else
    1.0 for float images already inside [0, 1], else the image maximum (must be positive)."""
    if max_value is not None:
        if max_value <= 0:
            raise ValueError("max_value must be positive")
        return float(max_value)
    if np.issubdtype(image.dtype, np.integer):
        return float(np.iinfo(image.dtype).max)
    image_min, image_max = _nan_minmax(image)
    if image_min >= 0.0 and image_max <= 1.0:
        return 1.0
    if image_max <= 0.0:
        raise ValueError("the image maximum must be positive to normalise to [0, 1]")
    return image_max


def to_gray(image: np.ndarray, normalize: bool = False, max_value: Optional[float] = None) -> np.ndarray:
    """Gray float32 image from (H, W) or (H, W, >=3) pixels  (reference stereo_io.py:47-75).

    gray = 0.299 R + 0.587 G + 0.114 B in float32, left to right, each product and sum rounded
    separately; `normalize` divides by float32(scale) and clips to [0, 1].  The result is
    bit-identical to the reference's NumPy expression (the census transform downstream compares
    gray values with a strict '>', so a last-bit difference would change cost bits).
    """
    image = np.asarray(image)
    if image.ndim == 2:
        channels = 1
    elif image.ndim == 3 and image.shape[2] >= 3:
        channels = int(image.shape[2])
    else:
        raise ValueError("unsupported image shape")
    scale = _resolve_normalization_scale(image, max_value) if normalize else 1.0
    height, width = int(image.shape[0]), int(image.shape[1])
    dt = _DEVICE_DTYPES.get(image.dtype)
    if dt is None:  # other pixel types enter through the same float32 cast the reference applies first
        image, dt = image.astype(np.float32), _native.DT_F32
    pixels = np.ascontiguousarray(image)
    gray = np.empty((height, width), dtype=np.float32)
    if gray.size:
        _native.require_gpu()
        _native.check(_native.lib().stereo_to_gray_host(_native.ptr(pixels), dt, height, width, channels,
                                                        1 if normalize else 0, float(scale), _native.ptr(gray)))
    return gray


def ensure_same_shape(left: np.ndarray, right: np.ndarray) -> Tuple[int, int]:
    """Reference stereo_io.py:79-95: shapes must match and be 2-D; returns (height, width)."""
    if left.shape != right.shape:
        raise ValueError("left/right image sizes differ")
    if left.ndim != 2:
        raise ValueError("gray images must be 2-D")
    return int(left.shape[0]), int(left.shape[1])


def _next_token_line(handler) -> str:
    while True:
        raw = handler.readline()
        if not raw:
            raise ValueError("truncated PFM file")
        text = raw.decode("ascii").strip()
        if text and not text.startswith("#"):
            return text


def read_pfm(path: str) -> np.ndarray:
    """PFM -> float32 (H, W) or (H, W, 3), top row first  (reference stereo_io.py:98-139).

    Header "Pf" (gray) / "PF" (colour), then "width height", then a scale whose sign gives the
    byte order (negative = little endian); rows are stored bottom-up."""
    with open(path, "rb") as handler:
        magic = handler.readline().decode("ascii").strip()
        if magic not in ("PF", "Pf"):
            raise ValueError("PFM header must be PF or Pf")
        channels = 3 if magic == "PF" else 1
        width_s, height_s = _next_token_line(handler).split()
        width, height = int(width_s), int(height_s)
        byte_order = "<" if float(_next_token_line(handler)) < 0 else ">"
        count = width * height * channels
        data = np.fromfile(handler, dtype=f"{byte_order}f4", count=count)
    if data.size != count:
        raise ValueError("PFM payload has the wrong size")
    shape = (height, width, 3) if channels == 3 else (height, width)
    return np.ascontiguousarray(data.reshape(shape)[::-1]).astype(np.float32)


def read_pgm_mask(path: str) -> np.ndarray:
    """PGM mask -> bool array, True where the pixel is non-zero  (reference stereo_io.py:142-156)."""
    image = read_image(path)
    if image.ndim == 3:
        image = image[..., 0]
    if image.ndim != 2:
        raise ValueError("the PGM mask must be 2-D")
    return image > 0


def write_pfm(path: str, image: np.ndarray, scale: float = 1.0) -> None:
    """float32 (H, W) or (H, W, 3) -> PFM in native byte order  (reference stereo_io.py:159-187)."""
    image = np.asarray(image)
    if image.ndim not in (2, 3):
        raise ValueError("a PFM image must be 2-D or 3-D")
    if image.ndim == 3 and image.shape[2] != 3:
        raise ValueError("a colour PFM needs 3 channels")
    payload = np.ascontiguousarray(image[::-1]).astype(np.float32)
    order = payload.dtype.byteorder
    little = order == "<" or (order == "=" and np.little_endian)
    signed_scale = -abs(scale) if little else abs(scale)
    with open(path, "wb") as handler:
        handler.write(("PF\n" if image.ndim == 3 else "Pf\n").encode("ascii"))
        handler.write(f"{int(image.shape[1])} {int(image.shape[0])}\n".encode("ascii"))
        handler.write(f"{signed_scale}\n".encode("ascii"))
        payload.tofile(handler)


def save_npz(path: str, arrays: Dict[str, np.ndarray]) -> None:
    """Reference stereo_io.py:190-202."""
    if not arrays:
        raise ValueError("arrays must not be empty")
    np.savez_compressed(path, **arrays)


def load_npz(path: str) -> Dict[str, np.ndarray]:
    """Reference stereo_io.py:205-215."""
    with np.load(path) as data:
        return {key: data[key] for key in data.files}


def save_disparity_npz(path: str, disparity: np.ndarray, min_cost: np.ndarray) -> None:
    """Reference stereo_io.py:218-231."""
    if disparity.shape != min_cost.shape:
        raise ValueError("disparity and min_cost sizes differ")
    save_npz(path, {"disparity": disparity, "min_cost": min_cost})


def load_disparity_npz(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """Reference stereo_io.py:234-246."""
    data = load_npz(path)
    if "disparity" not in data or "min_cost" not in data:
        raise ValueError("the NPZ must hold disparity and min_cost")
    return data["disparity"], data["min_cost"]


def convert_npz_to_pfm(npz_path: str, pfm_path: str, key: str = "disparity") -> None:
    """Reference stereo_io.py:249-263."""
    data = load_npz(npz_path)
    if key not in data:
        raise ValueError(f"the NPZ has no key: {key}")
    write_pfm(pfm_path, data[key])


def convert_pfm_to_npz(pfm_path: str, npz_path: str, key: str = "disparity") -> None:
    """Reference stereo_io.py:266-278."""
    save_npz(npz_path, {key: read_pfm(pfm_path)})
