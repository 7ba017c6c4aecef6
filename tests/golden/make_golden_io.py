#!/usr/bin/env python3
"""Golden vectors for the steps either side of the hot path: gray conversion, PFM files, the
Jet colour map and the PBM / RMS metrics.

Run in the build container only (it imports the reference unmodified from /root/reference,
with the identity `numba` stand-in next to this file, exactly like make_golden.py):

    python tests/golden/make_golden_io.py

Writes tests/golden/io_metrics.npz.  Inputs are seeded random arrays; expected outputs come
from the reference's stereo_io.to_gray / write_pfm / read_pfm and stereo._jet_colormap /
_compute_pbm_rms.
"""
from __future__ import annotations

import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("STEREO_REFERENCE_DIR", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_numba_standin"))
sys.path.insert(0, REFERENCE)

import numpy as np  # noqa: E402

import stereo as ref_stereo  # noqa: E402
import stereo_io as ref_io  # noqa: E402

assert ref_io.__file__.startswith(REFERENCE), ref_io.__file__
assert ref_stereo.__file__.startswith(REFERENCE), ref_stereo.__file__


def main() -> None:
    rng = np.random.default_rng(20240607)
    out = {}
    # gray conversion: every pixel type the device kernel takes, plus types that go through the float32 cast
    rgb_u8 = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    rgba_u8 = rng.integers(0, 256, size=(19, 31, 4), dtype=np.uint8)
    rgb_u16 = rng.integers(0, 65536, size=(23, 41, 3), dtype=np.uint16)
    gray_u8 = rng.integers(0, 256, size=(17, 29), dtype=np.uint8)
    gray_u16 = rng.integers(0, 65536, size=(17, 29), dtype=np.uint16)
    rgb_f32_unit = rng.random((21, 33, 3), dtype=np.float32)
    rgb_f32_wide = (rng.random((21, 33, 3), dtype=np.float32) * 300.0 - 20.0).astype(np.float32)
    rgb_f32_wide[3, 4, 1] = np.nan
    rgb_f64 = rng.random((11, 13, 3)) * 1000.0
    gray_i32 = rng.integers(0, 100000, size=(9, 14), dtype=np.int32)
    cases = {
        "rgb_u8": rgb_u8, "rgba_u8": rgba_u8, "rgb_u16": rgb_u16, "gray_u8": gray_u8, "gray_u16": gray_u16,
        "rgb_f32_unit": rgb_f32_unit, "rgb_f32_wide": rgb_f32_wide, "rgb_f64": rgb_f64, "gray_i32": gray_i32,
    }
    for name, img in cases.items():
        out[f"gray_in_{name}"] = img
        out[f"gray_raw_{name}"] = ref_io.to_gray(img, normalize=False)
        out[f"gray_norm_{name}"] = ref_io.to_gray(img, normalize=True)
    out["gray_norm_rgb_u8_max200"] = ref_io.to_gray(rgb_u8, normalize=True, max_value=200.0)

    # PFM: the reference's writer output as bytes, and its reader's view of a big-endian colour file
    gt = (rng.random((13, 17), dtype=np.float32) * 64.0).astype(np.float32)
    gt[2, 3] = np.inf
    color = rng.random((7, 9, 3), dtype=np.float32)
    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, "a.pfm")
        ref_io.write_pfm(p, gt)
        out["pfm_gray_image"] = gt
        out["pfm_gray_bytes"] = np.fromfile(p, dtype=np.uint8)
        ref_io.write_pfm(p, color, scale=2.5)
        out["pfm_color_image"] = color
        out["pfm_color_bytes"] = np.fromfile(p, dtype=np.uint8)
        big = b"PF\n# comment line\n9 7\n\n1.0\n" + np.flipud(color).astype(">f4").tobytes()
        with open(p, "wb") as fh:
            fh.write(big)
        out["pfm_big_endian_bytes"] = np.frombuffer(big, dtype=np.uint8)
        out["pfm_big_endian_read"] = ref_io.read_pfm(p)

    # Jet colour map
    vals = np.linspace(-0.2, 1.2, 15 * 11, dtype=np.float32).reshape(15, 11)
    out["jet_in"] = vals
    out["jet_out"] = ref_stereo._jet_colormap(vals)

    # metrics
    h, w = 61, 83
    disp = rng.integers(0, 64, size=(h, w), dtype=np.int32)
    truth = (disp.astype(np.float32) + rng.normal(0.0, 1.5, size=(h, w)).astype(np.float32)).astype(np.float32)
    truth[rng.random((h, w)) < 0.05] = np.inf
    mask = rng.random((h, w)) < 0.8
    out["metrics_disp"], out["metrics_gt"], out["metrics_mask"] = disp, truth, mask
    for thr in (0.5, 1.0, 2.0):
        m = ref_stereo._compute_pbm_rms(disp, truth, mask, thr)
        out[f"metrics_thr{thr}"] = np.array([m["pbm"], m["rms"], m["bad_threshold"], m["valid_pixel_count"]])

    path = os.path.join(HERE, "io_metrics.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
