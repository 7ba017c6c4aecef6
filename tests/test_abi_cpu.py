"""CPU-side checks of the C ABI: the in-tree library loads, exports every symbol the header
declares, validates arguments without a device, and fails loudly (no CPU fallback) when no GPU
is visible."""
from __future__ import annotations

import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO_ROOT, has_gpu


def test_library_exports_every_declared_symbol():
    import _native
    lib = _native.lib()
    header = open(os.path.join(REPO_ROOT, "include", "stereo_hip.h")).read()
    declared = set(re.findall(r"\b(stereo_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found in include/stereo_hip.h"
    for name in sorted(declared):
        assert hasattr(lib, name), f"libstereo_hip.so does not export {name}"
    assert declared == set(_native.exported_symbols())
    assert b"gfx950" in lib.stereo_hip_version()


def test_host_only_entry_points():
    import _native
    lib = _native.lib()
    off = np.zeros((32, 3), dtype=np.int32)
    assert lib.stereo_generate_offsets(4, _native.ptr(off)) == 0
    assert off[:5].tolist() == [[1, 0, 1], [2, 0, 2], [3, 0, 3], [4, 0, 4], [-1, 0, 1]]
    w = np.zeros(32, dtype=np.float32)
    assert lib.stereo_compute_weights(_native.ptr(off), 32, 8.0, _native.ptr(w)) == 0
    assert w[:4].tolist() == [8.0, 4.0, 2.0, 1.0] and float(w.sum()) == 120.0
    large = ctypes.c_float(0)
    assert lib.stereo_large_value(4, 8.0, ctypes.byref(large)) == 0 and large.value == 1200.0
    assert lib.stereo_large_value(8, 8.0, ctypes.byref(large)) == 0 and large.value == 1275.0
    assert lib.stereo_generate_offsets(0, _native.ptr(off)) == _native.EINVAL
    assert lib.stereo_large_value(9, 8.0, ctypes.byref(large)) == _native.EUNSUPPORTED
    assert "radius" in _native.last_error()


def test_python_validation_runs_before_any_device_work():
    import census
    import filters
    import guided_filter
    import stereo
    img = np.zeros((8, 8), dtype=np.float32)
    with pytest.raises(ValueError):
        census.compute_wct_cost_volume(img, np.zeros((8, 9), np.float32), 4)
    with pytest.raises(ValueError):
        census.compute_wct_cost_volume(img, img, -1)
    with pytest.raises(ValueError):
        stereo.compute_disparity(img, img, 0, show_progress=False)
    with pytest.raises(ValueError):
        stereo.aggregate_and_wta(np.zeros((8, 8, 2), np.float32), img, 3, 0.01, filter_type="x")
    with pytest.raises(ValueError):
        filters.median_filter(img, 0)
    with pytest.raises(ValueError):
        filters.gaussian_filter(img, -1.0)
    with pytest.raises(ValueError):
        guided_filter.prepare_guided_filter(img, 3, 0.0)
    assert census.generate_offsets(1) == [(1, 0, 1), (-1, 0, 1), (0, 1, 1), (0, -1, 1), (1, 1, 1), (1, -1, 1),
                                          (-1, 1, 1), (-1, -1, 1)]
    idx = guided_filter._prepare_box_filter_index(5, 6, 2)
    assert idx.area[0, 0] == 9.0 and idx.area[2, 3] == 25.0 and idx.y0.tolist() == [0, 0, 0, 1, 2]


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a machine without a GPU")
def test_no_cpu_fallback_without_a_gpu():
    import census
    img = np.zeros((16, 16), dtype=np.float32)
    with pytest.raises(RuntimeError, match="no HIP device"):
        census.compute_wct_cost_volume(img, img, 4)


def test_null_pointers_are_rejected_before_any_device_work():
    """Every pointer-taking entry point answers a NULL data pointer with STEREO_EINVAL (no crash, no kernel launched
    on a bad address); checked without a GPU, so the rejection demonstrably precedes all device calls."""
    import ctypes

    import _native

    lib = _native.lib()
    p = _native.StereoParams(wct_radius=4, base_weight=8.0, filter_kind=_native.FILTER_GUIDED, guided_radius=3,
                             guided_eps=0.0154, median_radius=3, gaussian_sigma=1.0, bilateral_sigma=1.0)
    E = _native.EINVAL
    N = None
    assert lib.stereo_compute_disparity_dev(N, N, 8, 8, 4, p, 0, 4, N, 0, N, N) == E
    assert lib.stereo_compute_disparity_dev(N, N, 8, 8, 4, None, 0, 4, N, 0, N, N) == E
    assert lib.stereo_compute_disparity_progress_dev(N, N, 8, 8, 4, p, 0, 4, N, 0, N, 1, N, 1, N) == E
    assert lib.stereo_compute_disparity_noalloc_dev(N, N, 8, 8, 4, p, 0, 4, N, N) == E
    assert lib.stereo_wct_cost_volume_dev(N, N, 8, 8, 4, 4, 8.0, N, 0, N, N) == E
    assert lib.stereo_aggregate_wta_dev(N, N, 8, 8, 4, p, 0, 4, N, 0, N, N) == E
    assert lib.stereo_wait_progress_dev(N, 0, 1, N) == E
    assert lib.stereo_progress_plan(8, 8, None, 1, N, None) == E
    assert lib.stereo_merge_keys_dev(N, N, 4, N) == E
    assert lib.stereo_merge_key_slices_dev(N, 4, 2, N) == E
    assert lib.stereo_keys_to_i64_dev(N, N, 4, N) == E
    assert lib.stereo_keys_from_i64_dev(N, N, 4, N) == E
    assert lib.stereo_unpack_keys_dev(N, 4, N, N, N) == E
    assert lib.stereo_fused_workspace_bytes(8, 8, p, None) == E
    assert lib.stereo_compute_disparity_host(N, N, _native.DT_F32, 8, 8, 4, p, 0, 4, N, N) == E
    assert lib.stereo_aggregate_wta_host(N, N, _native.DT_F32, 8, 8, 4, p, 0, 4, N, N) == E
    assert lib.stereo_wct_cost_volume_host(N, N, _native.DT_F32, 8, 8, 4, 4, 8.0, N) == E
    assert lib.stereo_integral_image_host(N, 8, 8, N) == E
    assert lib.stereo_box_filter_mean_host(N, 8, 8, 2, N) == E
    assert lib.stereo_box_filter_mean_indexed_host(N, 8, 8, N, N, N, N, N, N) == E
    assert lib.stereo_guided_prepare_host(N, 8, 8, 2, N, N, N) == E
    assert lib.stereo_guided_filter_host(N, N, 8, 8, 2, 0.01, N) == E
    assert lib.stereo_median_filter_host(N, 8, 8, 1, 0, N) == E
    assert lib.stereo_median_filter_u8_host(N, 8, 8, 1, N) == E
    assert lib.stereo_gaussian_filter_host(N, 8, 8, 1.0, N) == E
    assert lib.stereo_bilateral_filter_host(N, 8, 8, 1.0, N) == E
    assert lib.stereo_to_gray_host(N, _native.DT_U8, 8, 8, 3, 1, 255.0, N) == E
    assert lib.stereo_nan_minmax_host(N, 4, None, None) == E
    assert lib.stereo_pbm_rms_host(N, N, N, 8, 8, 1.0, N) == E
    assert lib.stereo_generate_offsets(4, N) == E
    assert lib.stereo_compute_weights(N, 32, 8.0, N) == E
    assert lib.stereo_large_value(4, 8.0, None) == E
    assert lib.stereo_hip_device_count(None) == E
    assert "null" in _native.last_error().lower()


def test_bad_arguments_are_rejected_before_any_device_work():
    """Argument checks of the device entry points run before the first HIP call: without a GPU they still answer
    STEREO_EINVAL (a call that reached the device layer would answer STEREO_EHIP here)."""
    import ctypes

    import _native

    lib = _native.lib()
    p = _native.StereoParams(wct_radius=4, base_weight=8.0, filter_kind=_native.FILTER_GUIDED, guided_radius=3,
                             guided_eps=0.0154, median_radius=3, gaussian_sigma=1.0, bilateral_sigma=1.0)
    fake = ctypes.create_string_buffer(64)   # never dereferenced by a call that is rejected
    ptr = ctypes.cast(fake, ctypes.c_void_p)
    nbytes = ctypes.c_size_t(0)
    _native.check(lib.stereo_fused_workspace_bytes(40, 60, p, ctypes.byref(nbytes)))

    def call(h=40, w=60, dmax=8, params=p, d0=0, d1=8, ws_bytes=None):
        return lib.stereo_compute_disparity_dev(ptr, ptr, h, w, dmax, params, d0, d1, ptr,
                                                nbytes.value if ws_bytes is None else ws_bytes, ptr, None)

    E = _native.EINVAL
    assert call(ws_bytes=nbytes.value - 1) == E
    assert call(d0=5, d1=4) == E and call(d1=9) == E and call(d0=-1) == E
    assert call(dmax=0, d1=0) == E
    assert call(h=0) == E and call(w=-3) == E
    bad_r = _native.StereoParams(wct_radius=9, base_weight=8.0, filter_kind=_native.FILTER_GUIDED, guided_radius=3,
                                 guided_eps=0.0154, median_radius=3, gaussian_sigma=1.0, bilateral_sigma=1.0)
    assert call(params=bad_r) == _native.EUNSUPPORTED   # census radius above 8: no packed word layout
    bad_eps = _native.StereoParams(wct_radius=4, base_weight=8.0, filter_kind=_native.FILTER_GUIDED, guided_radius=3,
                                   guided_eps=0.0, median_radius=3, gaussian_sigma=1.0, bilateral_sigma=1.0)
    assert call(params=bad_eps) == E
    # the entry point that allocates nothing: the same checks, and frames beyond its static scratch
    def call_noalloc(h=40, w=60, dmax=8, params=p, d0=0, d1=8):
        return lib.stereo_compute_disparity_noalloc_dev(ptr, ptr, h, w, dmax, params, d0, d1, ptr, None)
    assert call_noalloc(d0=5, d1=4) == E and call_noalloc(d1=9) == E and call_noalloc(dmax=0, d1=0) == E
    assert call_noalloc(h=0) == E and call_noalloc(params=bad_eps) == E
    assert call_noalloc(h=1025, w=1024) == E and "exceeds" in _native.last_error()
    assert call_noalloc(params=bad_r) == _native.EUNSUPPORTED
    # the two-call device entry points
    assert lib.stereo_wct_cost_volume_dev(ptr, ptr, 40, 60, 0, 4, 8.0, ptr, nbytes.value, ptr, None) == E
    assert lib.stereo_wct_cost_volume_dev(ptr, ptr, 40, 60, 8, 4, 8.0, ptr, nbytes.value - 1, ptr, None) == E
    assert lib.stereo_aggregate_wta_dev(ptr, ptr, 40, 60, 8, p, 3, 2, ptr, nbytes.value, ptr, None) == E
    assert lib.stereo_aggregate_wta_dev(ptr, ptr, 40, 60, 8, p, 0, 8, ptr, nbytes.value - 1, ptr, None) == E
    row_end = (ctypes.c_int32 * 16)()
    n = ctypes.c_int(0)
    assert lib.stereo_progress_plan(40, 60, p, 0, row_end, ctypes.byref(n)) == E
    assert lib.stereo_progress_plan(40, 60, p, 17, row_end, ctypes.byref(n)) == E
    assert lib.stereo_progress_plan(40, 60, p, 4, row_end, ctypes.byref(n)) == _native.OK and n.value >= 1
    assert lib.stereo_compute_disparity_progress_dev(ptr, ptr, 40, 60, 8, p, 0, 8, ptr, nbytes.value, ptr, 0, ptr, 1, None) == E
    assert lib.stereo_compute_disparity_progress_dev(ptr, ptr, 40, 60, 8, p, 0, 8, ptr, nbytes.value, ptr, 3, ptr, 1, None) == E
    assert lib.stereo_wait_progress_dev(ptr, 16, 1, None) == E
