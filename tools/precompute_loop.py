#!/usr/bin/env python3
"""Time of the reference's per-layer loop -- prepare_guided_filter once, guided_filter_with_precompute per layer
(reference stereo.py:143-150) -- on a Motorcycle-sized frame, through the drop-in module.
    python tools/precompute_loop.py [layers]
Layers are float32 (H, W) arrays, contiguous (the copy of a strided dsi[:, :, d] view is the caller's in either
implementation)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
import guided_filter as gf  # noqa: E402
from synth import synth_pair_gray  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 270
h, w = 2000, 2964
left, _, _ = synth_pair_gray(h, w, 16, 2014)
rng = np.random.default_rng(0)
pool = [rng.integers(0, 121, size=(h, w)).astype(np.float32) for _ in range(4)]
t0 = time.perf_counter()
pre = gf.prepare_guided_filter(left, 3, 0.0154)
t_prepare = time.perf_counter() - t0
gf.guided_filter_with_precompute(pre, pool[0])      # first call: uploads the guide, computes its statistics
t0 = time.perf_counter()
for d in range(layers):
    gf.guided_filter_with_precompute(pre, pool[d % 4])
dt = time.perf_counter() - t0
print(f"prepare_guided_filter {t_prepare * 1e3:.1f} ms; {layers} layers of {w}x{h}: {dt:.3f} s = {dt / layers * 1e3:.2f} ms per layer")
