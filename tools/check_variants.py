#!/usr/bin/env python3
"""Agreement of the three fused guided kernels on large synthetic frames (not part of the test suite:
the general kernel takes tens of milliseconds per frame at these sizes, the host copies seconds)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import _native  # noqa: E402
import stereo  # noqa: E402
from synth import synth_pair_gray  # noqa: E402


def main() -> None:
    lib = _native.lib()
    for (h, w, dmax, wct, rg, seed) in ((2160, 3840, 512, 4, 3, 3840), (4320, 7680, 256, 8, 3, 7680), (1500, 2000, 300, 8, 4, 5),
                                        (1500, 2000, 300, 4, 2, 6), (1100, 1900, 200, 4, 1, 7)):
        left, right, _ = synth_pair_gray(h, w, dmax, seed)
        res = {}
        for mode in (1, 0, 2):
            _native.check(lib.stereo_debug_force_general_kernel(mode))
            res[mode] = stereo.compute_disparity(left, right, dmax, wct_radius=wct, guided_radius=rg, guided_eps=0.0154,
                                                 show_progress=False)
        _native.check(lib.stereo_debug_force_general_kernel(0))
        g = res[1]
        for mode in (0, 2):
            d, c = res[mode]
            rel = np.abs(c.astype(np.float64) - g[1]) / np.maximum(1.0, np.abs(g[1].astype(np.float64)))
            print(f"{w}x{h} dmax {dmax} census r {wct} guided r {rg}: kernel {mode} vs general: max rel {rel.max():.2e}, "
                  f"disparity agreement {np.mean(d == g[0]):.6f}", flush=True)


if __name__ == "__main__":
    main()
