"""The entry point that allocates nothing (stereo_compute_disparity_noalloc_dev): same kernels as
stereo_compute_disparity_dev, with the library's static device array as workspace.  Through the C ABI: two calls
in a row on one GPU, the second on a smaller frame (the array keeps the first call's contents), each compared
bit for bit with the existing entry point; then a frame beyond the stated maximum is refused."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(native, filt, wct_radius):
    kind = {"guided": native.FILTER_GUIDED, "median": native.FILTER_MEDIAN_U8Q, "gaussian": native.FILTER_GAUSSIAN,
            "bilateral": native.FILTER_BILATERAL}[filt]
    return native.StereoParams(wct_radius=wct_radius, base_weight=8.0, filter_kind=kind, guided_radius=3, guided_eps=0.0154,
                               median_radius=3, gaussian_sigma=1.0, bilateral_sigma=1.0)


def _both(lib, native, torch, dev, h, w, dmax, params, d_begin, d_end, seed):
    from synth import synth_pair_gray

    left_np, right_np, _ = synth_pair_gray(h, w, dmax, seed=seed)
    left, right = torch.from_numpy(left_np).to(dev), torch.from_numpy(right_np).to(dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws_bytes = ctypes.c_size_t(0)
    native.check(lib.stereo_fused_workspace_bytes(h, w, params, ctypes.byref(ws_bytes)))
    ws = torch.empty(ws_bytes.value, dtype=torch.uint8, device=dev)
    ref = torch.full((h * w,), -1, dtype=torch.int64, device=dev)
    native.check(lib.stereo_compute_disparity_dev(left.data_ptr(), right.data_ptr(), h, w, dmax, params, d_begin, d_end,
                                                  ws.data_ptr(), ws_bytes.value, ref.data_ptr(), stream))
    got = torch.full((h * w,), -2, dtype=torch.int64, device=dev)
    native.check(lib.stereo_compute_disparity_noalloc_dev(left.data_ptr(), right.data_ptr(), h, w, dmax, params, d_begin,
                                                          d_end, got.data_ptr(), stream))
    torch.cuda.synchronize()
    return ref.cpu().numpy(), got.cpu().numpy()


@pytest.mark.parametrize("filt,wct_radius", [("guided", 4), ("median", 8), ("gaussian", 4), ("bilateral", 8)])
def test_two_calls_in_a_row_match_the_workspace_entry_point(filt, wct_radius):
    import torch

    import _native

    lib = _native.lib()
    _native.check(lib.stereo_hip_init(0))
    dev = torch.device("cuda", 0)
    params = _params(_native, filt, wct_radius)
    # first a frame close to the maximum, then a smaller one with a disparity sub-range
    ref, got = _both(lib, _native, torch, dev, 700, 1400, 40, params, 0, 40, seed=21)
    assert np.array_equal(ref, got)
    ref, got = _both(lib, _native, torch, dev, 211, 333, 24, params, 5, 19, seed=22)
    assert np.array_equal(ref, got)


def test_frames_beyond_the_static_scratch_are_refused():
    import torch

    import _native

    lib = _native.lib()
    dev = torch.device("cuda", 0)
    h, w = 1025, 1024   # one row more than the 2^20 pixels the static array is sized for
    img = torch.zeros(h * w, dtype=torch.float32, device=dev)
    keys = torch.empty(h * w, dtype=torch.int64, device=dev)
    params = _params(_native, "guided", 4)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.stereo_compute_disparity_noalloc_dev(img.data_ptr(), img.data_ptr(), h, w, 8, params, 0, 8, keys.data_ptr(), stream)
    assert rc == _native.EINVAL
    assert b"exceeds" in lib.stereo_hip_last_error()
    # the largest frame it takes
    h = 1024
    rc = lib.stereo_compute_disparity_noalloc_dev(img.data_ptr(), img.data_ptr(), h, w, 8, params, 0, 8, keys.data_ptr(), stream)
    torch.cuda.synchronize()
    assert rc == _native.OK
