#!/usr/bin/env python3
"""Compute-only time of every rank's disparity shard on ONE GPU (census packing, guide statistics, fused
kernel; no exchange): the floor under the N-GPU strong-scaling curve of bench.py.
    python tools/shard_compute.py [motorcycle|4k]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import _native  # noqa: E402
import sharding  # noqa: E402
from synth import synth_pair_gray  # noqa: E402


def main() -> None:
    H, W, D = (2160, 3840, 512) if (len(sys.argv) > 1 and sys.argv[1] == "4k") else (2000, 2964, 270)
    lib = _native.lib()
    _native.check(lib.stereo_hip_init(0))
    dev = torch.device("cuda", 0)
    left_np, right_np, _ = synth_pair_gray(H, W, D, 2014)
    left, right = torch.from_numpy(left_np).to(dev), torch.from_numpy(right_np).to(dev)
    params = _native.StereoParams(wct_radius=4, base_weight=8.0, filter_kind=_native.FILTER_GUIDED, guided_radius=3,
                                  guided_eps=0.0154, median_radius=3, gaussian_sigma=1.0, bilateral_sigma=1.0)
    ws_bytes = ctypes.c_size_t(0)
    _native.check(lib.stereo_fused_workspace_bytes(H, W, params, ctypes.byref(ws_bytes)))
    ws = torch.empty(ws_bytes.value, dtype=torch.uint8, device=dev)
    keys = torch.empty(H * W, dtype=torch.int64, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for world in (1, 2, 4, 8):
        worst_wall, worst_kernel = 0.0, 0.0
        for rank in range(world):
            d0, d1 = sharding.shard_range(D, world, rank)

            def run():
                _native.check(lib.stereo_compute_disparity_dev(left.data_ptr(), right.data_ptr(), H, W, D, params, d0, d1,
                                                               ws.data_ptr(), ws_bytes.value, keys.data_ptr(), stream))
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            steps = 10
            _native.check(lib.stereo_profile_begin(steps))
            t0 = time.perf_counter()
            for _ in range(steps):
                run()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / steps * 1e3
            ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
            _native.check(lib.stereo_profile_end(ctypes.byref(ms), ctypes.byref(n)))
            worst_wall, worst_kernel = max(worst_wall, wall), max(worst_kernel, ms.value / max(n.value, 1))
        print(f"N={world}: slowest shard {worst_wall:.3f} ms per step (fused kernel {worst_kernel:.3f} ms)", flush=True)

    # rows x disparities: a rank searches a horizontal part of the frame (cut on tile rows, one tile row of halo on
    # each inner side, so that its tiles coincide with the whole frame's) for a contiguous disparity range
    th = 30
    tile_rows = -(-H // th)
    for ry, rd in ((2, 4), (4, 2), (8, 1), (2, 2), (2, 1)):
        worst = 0.0
        for iy in range(ry):
            t0r, t1r = iy * tile_rows // ry, (iy + 1) * tile_rows // ry
            row0, row1 = max(0, (t0r - 1) * th), min(H, (t1r + 1) * th)
            for idd in range(rd):
                d0, d1 = sharding.shard_range(D, rd, idd)

                def run():
                    _native.check(lib.stereo_compute_disparity_dev(left.data_ptr() + row0 * W * 4, right.data_ptr() + row0 * W * 4,
                                                                   row1 - row0, W, D, params, d0, d1, ws.data_ptr(),
                                                                   ws_bytes.value, keys.data_ptr(), stream))
                for _ in range(2):
                    run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    run()
                torch.cuda.synchronize()
                worst = max(worst, (time.perf_counter() - t0) / 10 * 1e3)
        print(f"N={ry * rd} as {ry} row parts x {rd} disparity ranges: slowest shard {worst:.3f} ms per step", flush=True)


if __name__ == "__main__":
    main()
