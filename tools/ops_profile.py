#!/usr/bin/env python3
"""Runs the stand-alone operators at the Motorcycle frame size (2000 x 2964) a few times each, for
`rocprofv3 --kernel-trace --stats -- python3 tools/ops_profile.py`: integral image, box mean, guided filter
(fused kernel, radius 3; layer-at-a-time chain, radius 6), exact and uint8 median, Gaussian, bilateral.
Prints the wall time of each host call (which includes the copies to and from the device)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import filters  # noqa: E402
import guided_filter as gf  # noqa: E402


def main() -> None:
    h, w, reps = 2000, 2964, 5
    rng = np.random.default_rng(1)
    img64 = rng.random((h, w)) * 255.0
    guide = rng.random((h, w)).astype(np.float32)
    src = (rng.random((h, w)) * 1200).astype(np.float32)
    calls = [
        ("integral_image", lambda: gf.integral_image(img64)),
        ("box_filter_mean r=3", lambda: gf.box_filter_mean(img64, 3)),
        ("guided_filter r=3 (fused)", lambda: gf.guided_filter(guide, src, 3, 0.0154)),
        ("guided_filter r=6 (layer chain)", lambda: gf.guided_filter(guide, src, 6, 0.0154)),
        ("median_filter r=3 opencv", lambda: filters.median_filter(src, 3)),
        ("median_filter r=3 scipy", lambda: filters.median_filter(src, 3, method="scipy")),
        ("gaussian_filter sigma=1", lambda: filters.gaussian_filter(src, 1.0)),
        ("bilateral_filter sigma=1", lambda: filters.bilateral_filter(src, 1.0)),
    ]
    for name, fn in calls:
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        print(f"{name}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per host call", flush=True)


if __name__ == "__main__":
    main()
