#!/usr/bin/env python3
"""The 8K benchmark configuration (7680x4320, dmax 256, census radius 8; median / gaussian / bilateral) through the
public compute_disparity call: wall time of the whole call, host copies included."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
sys.path.insert(0, ROOT)

import stereo  # noqa: E402
from synth import synth_pair_gray  # noqa: E402


def main() -> None:
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    H, W, D = int(4320 * scale), int(7680 * scale), 256
    t0 = time.perf_counter()
    left, right, _ = synth_pair_gray(H, W, D, 7680)
    print(f"synthetic {W}x{H} pair in {time.perf_counter() - t0:.1f} s", flush=True)
    for filt in ("gaussian", "median", "bilateral", "guided"):
        for rep in range(2):
            t0 = time.perf_counter()
            stereo.compute_disparity(left, right, D, wct_radius=8, filter_type=filt, guided_eps=0.0154, show_progress=False)
            dt = time.perf_counter() - t0
            print(f"{filt:10s} run {rep}: {dt * 1e3:9.1f} ms  {H * W * D / dt / 1e6:10.1f} MDE/s", flush=True)


if __name__ == "__main__":
    main()
