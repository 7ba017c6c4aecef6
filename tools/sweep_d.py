#!/usr/bin/env python3
"""Kernel time of the fused guided path against the number of disparities, on a Motorcycle-shaped
synthetic pair: the intercept of the line is the per-tile fixed cost (staging, barriers, epilogue)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import _native  # noqa: E402
from synth import synth_pair_gray  # noqa: E402


def main() -> None:
    H, W = 2000, 2964
    lib = _native.lib()
    _native.check(lib.stereo_hip_init(0))
    dev = torch.device("cuda", 0)
    left_np, right_np, _ = synth_pair_gray(H, W, 270, 2014)
    left, right = torch.from_numpy(left_np).to(dev), torch.from_numpy(right_np).to(dev)
    params = _native.StereoParams(wct_radius=4, base_weight=8.0, filter_kind=_native.FILTER_GUIDED, guided_radius=3,
                                  guided_eps=0.0154, median_radius=3, gaussian_sigma=1.0, bilateral_sigma=1.0)
    ws_bytes = ctypes.c_size_t(0)
    _native.check(lib.stereo_fused_workspace_bytes(H, W, params, ctypes.byref(ws_bytes)))
    ws = torch.empty(ws_bytes.value, dtype=torch.uint8, device=dev)
    keys = torch.empty(H * W, dtype=torch.int64, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # the first range lies right of every column: all tiles skip all disparities (staging and epilogue only)
    for d0, d1 in [(3000, 3008), (100, 108), (100, 116), (100, 132), (100, 164), (100, 228), (0, 270)]:
        def run():
            _native.check(lib.stereo_compute_disparity_dev(left.data_ptr(), right.data_ptr(), H, W, max(270, d1), params, d0, d1,
                                                           ws.data_ptr(), ws_bytes.value, keys.data_ptr(), stream))
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        steps = 10
        _native.check(lib.stereo_profile_begin(steps))
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
        _native.check(lib.stereo_profile_end(ctypes.byref(ms), ctypes.byref(n)))
        print(f"disparities {d1 - d0:4d}: kernel {ms.value / max(n.value, 1):.4f} ms", flush=True)


if __name__ == "__main__":
    main()
