#!/usr/bin/env python3
"""Random large frames through stereo_compute_disparity_progress_dev with and without the tail split
(STEREO_NO_TAIL_SPLIT): keys and band flags must be identical.  Frames are large enough (more tiles than
compute units) for the split to be planned; disparity ranges, bands, census and guided radii vary.
    python tests/fuzz_tail_split.py [cases] [seed]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def run(cases: int, seed: int) -> int:
    import torch

    import _native

    lib = _native.lib()
    _native.check(lib.stereo_hip_init(0))
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    main = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    failures = 0
    for case in range(cases):
        h, w = int(rng.integers(600, 2400)), int(rng.integers(900, 4000))
        dmax = int(rng.integers(8, 200))
        d0 = int(rng.integers(0, dmax // 2))
        d1 = int(rng.integers(d0 + 1, dmax + 1))
        bands = int(rng.integers(1, 9))
        params = _native.StereoParams(wct_radius=int(rng.choice([3, 4, 6, 8])), base_weight=8.0, filter_kind=_native.FILTER_GUIDED,
                                      guided_radius=int(rng.integers(2, 5)), guided_eps=0.0154, median_radius=3,
                                      gaussian_sigma=1.0, bilateral_sigma=1.0)
        left = torch.rand(h * w, device=dev)
        left = (left * 255).round() / 255
        right = torch.roll(left.view(h, w), -int(rng.integers(0, dmax)), dims=1).contiguous().view(-1)
        nbytes = ctypes.c_size_t(0)
        _native.check(lib.stereo_fused_workspace_bytes(h, w, params, ctypes.byref(nbytes)))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        row_end = (ctypes.c_int32 * 16)()
        n_bands = ctypes.c_int(0)
        _native.check(lib.stereo_progress_plan(h, w, params, bands, row_end, ctypes.byref(n_bands)))
        out = []
        ok = True
        for no_split in ("1", "0"):
            os.environ["STEREO_NO_TAIL_SPLIT"] = no_split
            keys = torch.zeros(h * w, dtype=torch.int64, device=dev)
            progress = torch.zeros(48, dtype=torch.int32, device=dev)
            _native.check(lib.stereo_compute_disparity_progress_dev(
                left.data_ptr(), right.data_ptr(), h, w, dmax, params, d0, d1, ws.data_ptr(), nbytes.value, keys.data_ptr(),
                n_bands.value, progress.data_ptr(), 5, main))
            for b in range(n_bands.value):
                _native.check(lib.stereo_wait_progress_dev(progress.data_ptr(), b, 5, main))
            torch.cuda.synchronize()
            words = progress.cpu().numpy()
            ok = ok and list(words[16:16 + n_bands.value]) == [5] * n_bands.value and not words[32:48].any()
            out.append(keys)
        ok = ok and torch.equal(out[0], out[1])
        if not ok:
            failures += 1
            print(f"FAIL case {case}: {w}x{h} dmax {dmax} d[{d0},{d1}) bands {n_bands.value} r_c {params.wct_radius} r_g {params.guided_radius}",
                  flush=True)
        if (case + 1) % 10 == 0:
            print(f"... {case + 1} cases, {failures} failures", flush=True)
    os.environ.pop("STEREO_NO_TAIL_SPLIT", None)
    print(f"{cases} cases, {failures} failures")
    return failures


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sys.exit(1 if run(n, sd) else 0)
