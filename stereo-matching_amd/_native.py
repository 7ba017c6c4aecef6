"""ctypes binding of libstereo_hip.so (the C ABI declared in include/stereo_hip.h).

There is no CPU fallback: if the shared library is missing, or no HIP device is
visible when a compute entry point is called, the call fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# STEREO_HIP_LIB selects another build of the same library (A/B timing of compiler flags)
LIB_PATH = os.environ.get("STEREO_HIP_LIB") or os.path.join(_HERE, "libstereo_hip.so")

OK, EINVAL, EUNSUPPORTED, EHIP, ENOMEM, ENONFINITE = 0, -1, -2, -3, -4, -5
DT_F32, DT_F64, DT_U8, DT_U16 = 0, 1, 2, 3
FILTER_GUIDED, FILTER_MEDIAN, FILTER_MEDIAN_U8Q, FILTER_GAUSSIAN, FILTER_BILATERAL = 0, 1, 2, 3, 4


class StereoParams(ctypes.Structure):
    _fields_ = [
        ("wct_radius", ctypes.c_int32),
        ("base_weight", ctypes.c_double),
        ("filter_kind", ctypes.c_int32),
        ("guided_radius", ctypes.c_int32),
        ("guided_eps", ctypes.c_double),
        ("median_radius", ctypes.c_int32),
        ("gaussian_sigma", ctypes.c_double),
        ("bilateral_sigma", ctypes.c_double),
    ]


_lib: Optional[ctypes.CDLL] = None
_device_checked = False

_VP, _I, _D, _SZ = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
_PP = ctypes.POINTER(StereoParams)

_SIGNATURES = {
    "stereo_hip_init": [_I],
    "stereo_hip_device_count": [ctypes.POINTER(ctypes.c_int)],
    "stereo_generate_offsets": [_I, _VP],
    "stereo_compute_weights": [_VP, _I, _D, _VP],
    "stereo_large_value": [_I, _D, ctypes.POINTER(ctypes.c_float)],
    "stereo_wct_cost_volume_host": [_VP, _VP, _I, _I, _I, _I, _I, _D, _VP],
    "stereo_integral_image_host": [_VP, _I, _I, _VP],
    "stereo_box_filter_mean_host": [_VP, _I, _I, _I, _VP],
    "stereo_box_filter_mean_indexed_host": [_VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP],
    "stereo_guided_prepare_host": [_VP, _I, _I, _I, _VP, _VP, _VP],
    "stereo_guided_filter_host": [_VP, _VP, _I, _I, _I, _D, _VP],
    "stereo_guided_plan_create": [_VP, _I, _I, _I, _D, ctypes.POINTER(ctypes.c_void_p)],
    "stereo_guided_plan_apply": [_VP, _VP, _VP],
    "stereo_guided_plan_destroy": [_VP],
    "stereo_median_filter_host": [_VP, _I, _I, _I, _I, _VP],
    "stereo_median_filter_u8_host": [_VP, _I, _I, _I, _VP],
    "stereo_gaussian_filter_host": [_VP, _I, _I, _D, _VP],
    "stereo_bilateral_filter_host": [_VP, _I, _I, _D, _VP],
    "stereo_aggregate_wta_host": [_VP, _VP, _I, _I, _I, _I, _PP, _I, _I, _VP, _VP],
    "stereo_compute_disparity_host": [_VP, _VP, _I, _I, _I, _I, _PP, _I, _I, _VP, _VP],
    "stereo_fused_workspace_bytes": [_I, _I, _PP, ctypes.POINTER(ctypes.c_size_t)],
    "stereo_compute_disparity_noalloc_dev": [_VP, _VP, _I, _I, _I, _PP, _I, _I, _VP, _VP],
    "stereo_wct_cost_volume_dev": [_VP, _VP, _I, _I, _I, _I, _D, _VP, _SZ, _VP, _VP],
    "stereo_aggregate_wta_dev": [_VP, _VP, _I, _I, _I, _PP, _I, _I, _VP, _SZ, _VP, _VP],
    "stereo_compute_disparity_dev": [_VP, _VP, _I, _I, _I, _PP, _I, _I, _VP, _SZ, _VP, _VP],
    "stereo_aggregate_workspace_bytes": [_I, _I, _I, _PP, ctypes.POINTER(ctypes.c_size_t)],
    "stereo_cost_codes_bytes": [_I, _I, _I, ctypes.POINTER(ctypes.c_size_t)],
    "stereo_encode_cost_volume_dev": [_VP, _I, _I, _I, _I, _I, _VP, _SZ, _VP],
    "stereo_aggregate_wta_codes_dev": [_VP, _VP, _VP, _I, _I, _I, _PP, _I, _I, _VP, _SZ, _VP, _VP],
    "stereo_progress_plan": [_I, _I, _PP, _I, _VP, ctypes.POINTER(ctypes.c_int)],
    "stereo_compute_disparity_progress_dev": [_VP, _VP, _I, _I, _I, _PP, _I, _I, _VP, _SZ, _VP, _I, _VP, ctypes.c_uint32,
                                              _VP],
    "stereo_wait_progress_dev": [_VP, _I, ctypes.c_uint32, _VP],
    "stereo_streams_overlap": [_VP, _VP, ctypes.POINTER(ctypes.c_int)],
    "stereo_merge_keys_dev": [_VP, _VP, _SZ, _VP],
    "stereo_merge_key_slices_dev": [_VP, _SZ, _I, _VP],
    "stereo_keys_to_i64_dev": [_VP, _VP, _SZ, _VP],
    "stereo_keys_from_i64_dev": [_VP, _VP, _SZ, _VP],
    "stereo_unpack_keys_dev": [_VP, _SZ, _VP, _VP, _VP],
    "stereo_to_gray_host": [_VP, _I, _I, _I, _I, _I, _D, _VP],
    "stereo_nan_minmax_host": [_VP, _SZ, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)],
    "stereo_pbm_rms_host": [_VP, _VP, _VP, _I, _I, _D, _VP],
    "stereo_debug_force_general_kernel": [_I],
    "stereo_hip_release_cache": [],
    "stereo_profile_begin": [_I],
    "stereo_profile_end": [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)],
}


def exported_symbols():
    """Every function include/stereo_hip.h declares (used by the CPU-side ABI test)."""
    return sorted(list(_SIGNATURES) + ["stereo_hip_last_error", "stereo_hip_version"])


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64 with the system
    library's SONAME.  When torch is imported first, this library binds to torch's copy by SONAME;
    when this library were loaded first it would bring in the system copy, torch would later add its
    own, and the second runtime to initialise does not see the GPU ("No HIP GPUs are available").
    Loading torch's copy here first (torch itself is NOT imported) makes both orders equal.
    STEREO_HIP_SYSTEM_RUNTIME=1 keeps the system runtime for processes that never import torch."""
    import importlib.util
    import sys

    if "torch" in sys.modules or os.environ.get("STEREO_HIP_SYSTEM_RUNTIME") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with stereo-matching_amd/build.sh "
                "(there is no CPU fallback for the stereo hot path)"
            )
        _share_hip_runtime_with_torch()
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        handle.stereo_hip_last_error.restype = ctypes.c_char_p
        handle.stereo_hip_version.restype = ctypes.c_char_p
        _lib = handle
    return _lib


def last_error() -> str:
    return lib().stereo_hip_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc == OK:
        return
    msg = last_error()
    if rc in (EINVAL, ENONFINITE):
        raise ValueError(msg)
    if rc == EUNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def require_gpu() -> None:
    """Fail loudly when no HIP device is visible."""
    global _device_checked
    if _device_checked:
        return
    n = ctypes.c_int(0)
    lib().stereo_hip_device_count(ctypes.byref(n))
    if n.value <= 0:
        raise RuntimeError("no HIP device visible: the stereo hot path runs on an MI355X only (no CPU fallback)")
    _device_checked = True


def ptr(a: np.ndarray) -> ctypes.c_void_p:
    return a.ctypes.data_as(ctypes.c_void_p)


def image_for_census(a: np.ndarray):
    """Return (contiguous array, dtype code) with the census comparison order of `a` preserved.

    float32 / float64 / uint8 are compared natively on the device.  Other integer and float types
    are widened to a type that represents every value exactly (so `>` gives the same bits).
    """
    if a.dtype == np.float32:
        return np.ascontiguousarray(a), DT_F32
    if a.dtype == np.float64:
        return np.ascontiguousarray(a), DT_F64
    if a.dtype == np.uint8:
        return np.ascontiguousarray(a), DT_U8
    if a.dtype == np.bool_:
        return np.ascontiguousarray(a.astype(np.uint8)), DT_U8
    if a.dtype in (np.int8, np.int16, np.uint16, np.float16):
        return np.ascontiguousarray(a.astype(np.float32)), DT_F32
    if a.dtype in (np.int32, np.uint32):
        return np.ascontiguousarray(a.astype(np.float64)), DT_F64
    if a.dtype in (np.int64, np.uint64):
        if a.size and (np.abs(a.astype(np.float64)).max() >= 2.0 ** 53):
            raise NotImplementedError("64-bit integer images beyond 2**53 are not supported on the GPU path")
        return np.ascontiguousarray(a.astype(np.float64)), DT_F64
    raise TypeError(f"unsupported image dtype {a.dtype}")
