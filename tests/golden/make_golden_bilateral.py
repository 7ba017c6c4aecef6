#!/usr/bin/env python3
"""A larger golden vector of the bilateral aggregation from the reference itself: a frame whose interior dominates
its borders, so that the share of pixels on which two layers tie EXACTLY (constant `large` windows near the left
border) is small and the agreement of another implementation with the reference can be read off it.

Run in the build container only (the reference tree is not present on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_bilateral.py

The reference modules are imported unmodified from /root/reference under the identity stand-in for numba, exactly
as tests/golden/make_golden.py does (see there).  Only inputs and outputs are stored.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("STEREO_REFERENCE_DIR", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_numba_standin"))
sys.path.insert(0, REFERENCE)

import numpy as np  # noqa: E402

import census as ref_census  # noqa: E402
import stereo as ref_stereo  # noqa: E402

assert ref_census.__file__.startswith(REFERENCE) and ref_stereo.__file__.startswith(REFERENCE)

_spec = importlib.util.spec_from_file_location("amd_synth", os.path.join(REPO, "stereo-matching_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)


def main() -> None:
    h, w, dmax = 120, 200, 20
    left, right, _ = synth.synth_pair_gray(h, w, dmax, seed=4242)
    t0 = time.perf_counter()
    dsi = ref_census.compute_wct_cost_volume(left, right, dmax)
    disp, cost = ref_stereo.aggregate_and_wta(dsi, left, 3, 0.0154, filter_type="bilateral", bilateral_sigma=1.0)
    print(f"reference bilateral pipeline on {h}x{w}x{dmax}: {time.perf_counter() - t0:.1f} s")
    np.savez_compressed(os.path.join(HERE, "bilateral_120x200x20.npz"), left=left, right=right, dmax=np.int32(dmax),
                        disp_bilateral=disp, cost_bilateral=cost)


if __name__ == "__main__":
    main()
