#!/usr/bin/env python3
"""Wall time of the public compute_disparity call (host arrays in, host arrays out: H2D of two gray
images, every kernel, D2H of disparity and min cost) on the Motorcycle-shaped synthetic pair."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
sys.path.insert(0, ROOT)

import stereo  # noqa: E402
from synth import synth_pair_gray  # noqa: E402


def main() -> None:
    H, W, D = 2000, 2964, 270
    left, right, _ = synth_pair_gray(H, W, D, 2014)
    for rep in range(4):
        t0 = time.perf_counter()
        stereo.compute_disparity(left, right, D, guided_eps=0.0154, show_progress=False)
        dt = time.perf_counter() - t0
        print(f"run {rep}: {dt * 1e3:8.2f} ms  {H * W * D / dt / 1e6:10.1f} MDE/s", flush=True)


if __name__ == "__main__":
    main()
