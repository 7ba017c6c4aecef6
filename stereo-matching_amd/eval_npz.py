"""Evaluate a saved disparity.npz against ground truth with a chosen bad-pixel threshold -- same
flags and output as the reference's eval_npz.py (eval_npz.py:13-115); the reduction runs on the GPU."""
from __future__ import annotations

import argparse
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

from stereo import _compute_pbm_rms, _print_metrics, _write_metrics
from stereo_io import load_npz, read_pfm, read_pgm_mask


def _parse_args(argv: Optional[Sequence[str]] = None) -> argparse.Namespace:
    parser = argparse.ArgumentParser(description="Evaluate disparity NPZ with custom bad threshold")
    parser.add_argument("--npz", required=True, type=str, help="input NPZ")
    parser.add_argument("--gt", required=True, type=str, help="ground-truth PFM")
    parser.add_argument("--gt_mask", required=True, type=str, help="ground-truth mask PGM")
    parser.add_argument("--key", type=str, default="disparity", help="key of the disparity inside the NPZ")
    parser.add_argument("--bad_threshold", type=float, default=1.0, help="bad pixel threshold")
    parser.add_argument("--output_json", type=str, default="", help="where to write the metrics JSON")
    return parser.parse_args(argv)


def _load_disparity_from_npz(path: str, key: str) -> np.ndarray:
    data = load_npz(path)
    if key not in data:
        raise ValueError(f"the NPZ has no key: {key}")
    if data[key].ndim != 2:
        raise ValueError("disparity must be 2-D")
    return data[key]


def main(argv: Optional[Sequence[str]] = None) -> None:
    args = _parse_args(argv)
    disparity = _load_disparity_from_npz(args.npz, args.key)
    metrics = _compute_pbm_rms(disparity, read_pfm(args.gt), read_pgm_mask(args.gt_mask), args.bad_threshold)
    _print_metrics(metrics)
    if args.output_json:
        _write_metrics(Path(args.output_json), metrics)


if __name__ == "__main__":
    main()
