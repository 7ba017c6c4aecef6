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
