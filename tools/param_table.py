#!/usr/bin/env python3
"""Which kernel serves which parameter set, and what a Motorcycle-sized compute_disparity call costs with it
(2964 x 2000, dmax 270; host call, copies to and from the device included): one line per parameter set."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
sys.path.insert(0, ROOT)

import stereo  # noqa: E402
from synth import synth_pair_gray  # noqa: E402

CASES = [
    ("guided r=3 eps=0.0154 (default)", dict(guided_eps=0.0154)),
    ("guided r=3 eps=0.005 (two-column kernel, precise variant)", dict(guided_eps=0.005)),
    ("guided r=3 eps=0.0001 (two-column kernel, precise variant)", dict(guided_eps=0.0001)),
    ("guided r=1 eps=0.005 (precise variant)", dict(guided_radius=1, guided_eps=0.005)),
    ("guided r=1", dict(guided_radius=1, guided_eps=0.0154)),
    ("guided r=4", dict(guided_radius=4, guided_eps=0.0154)),
    ("guided r=5", dict(guided_radius=5, guided_eps=0.0154)),
    ("guided r=6 (one-column fused kernel)", dict(guided_radius=6, guided_eps=0.0154)),
    ("guided r=8 (one-column fused kernel)", dict(guided_radius=8, guided_eps=0.0154)),
    ("guided r=6 eps=0.003 (float64 fused kernel)", dict(guided_radius=6, guided_eps=0.003)),
    ("guided r=9 (no fused kernel)", dict(guided_radius=9, guided_eps=0.0154)),
    ("census radius 8, guided r=3", dict(wct_radius=8, guided_eps=0.0154)),
    ("base_weight 3.3 (weights not exactly summable), guided r=3", dict(base_weight=3.3, guided_eps=0.0154)),
    ("median r=3 opencv", dict(filter_type="median")),
    ("median r=3 scipy (exact float median)", dict(filter_type="median", median_method="scipy")),
    ("median r=5 opencv", dict(filter_type="median", median_radius=5)),
    ("gaussian sigma=1", dict(filter_type="gaussian")),
    ("gaussian sigma=2 (radius 6)", dict(filter_type="gaussian", gaussian_sigma=2.0)),
    ("bilateral sigma=1", dict(filter_type="bilateral")),
    ("bilateral sigma=2 (radius 6)", dict(filter_type="bilateral", bilateral_sigma=2.0)),
    ("gaussian sigma=3 (radius 9)", dict(filter_type="gaussian", gaussian_sigma=3.0)),
    ("bilateral sigma=3 (radius 9)", dict(filter_type="bilateral", bilateral_sigma=3.0)),
    ("median r=9 opencv", dict(filter_type="median", median_radius=9)),
    ("median r=13 opencv (no fused kernel)", dict(filter_type="median", median_radius=13)),
]


def main() -> None:
    only = sys.argv[1:]
    h, w, d = 2000, 2964, 270
    left, right, _ = synth_pair_gray(h, w, d, 2014)
    for name, kw in CASES:
        if only and not any(o in name for o in only):
            continue
        stereo.compute_disparity(left, right, d, show_progress=False, **kw)
        t0 = time.perf_counter()
        stereo.compute_disparity(left, right, d, show_progress=False, **kw)
        print(f"{name}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)


if __name__ == "__main__":
    main()
