#!/usr/bin/env python3
"""Kernel time of the fused guided kernels (two columns per lane / one column per lane / general)
for every guided radius and both census word sizes, on a synthetic 1482x1000 pair, dmax 128."""
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
    H, W, D = 1000, 1482, 128
    lib = _native.lib()
    _native.check(lib.stereo_hip_init(0))
    dev = torch.device("cuda", 0)
    left_np, right_np, _ = synth_pair_gray(H, W, D, 7)
    left, right = torch.from_numpy(left_np).to(dev), torch.from_numpy(right_np).to(dev)
    keys = torch.empty(H * W, dtype=torch.int64, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for wct in (4, 8):
        for rg in (1, 2, 3, 4):
            params = _native.StereoParams(wct_radius=wct, base_weight=8.0, filter_kind=_native.FILTER_GUIDED, guided_radius=rg,
                                          guided_eps=0.0154, median_radius=3, gaussian_sigma=1.0, bilateral_sigma=1.0)
            ws_bytes = ctypes.c_size_t(0)
            _native.check(lib.stereo_fused_workspace_bytes(H, W, params, ctypes.byref(ws_bytes)))
            ws = torch.empty(ws_bytes.value, dtype=torch.uint8, device=dev)
            out = []
            for mode in (0, 2, 1):
                _native.check(lib.stereo_debug_force_general_kernel(mode))

                def run():
                    _native.check(lib.stereo_compute_disparity_dev(left.data_ptr(), right.data_ptr(), H, W, D, params, 0, D,
                                                                   ws.data_ptr(), ws_bytes.value, keys.data_ptr(), stream))
                for _ in range(2):
                    run()
                torch.cuda.synchronize()
                steps = 5
                _native.check(lib.stereo_profile_begin(steps))
                for _ in range(steps):
                    run()
                torch.cuda.synchronize()
                ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
                _native.check(lib.stereo_profile_end(ctypes.byref(ms), ctypes.byref(n)))
                out.append(ms.value / max(n.value, 1))
            _native.check(lib.stereo_debug_force_general_kernel(0))
            print(f"census radius {wct} guided radius {rg}: default {out[0]:.3f} ms, one column per lane {out[1]:.3f} ms, "
                  f"general {out[2]:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
