"""NPZ <-> PFM conversion tool -- same flags as the reference's convert.py (convert.py:8-30)."""
from __future__ import annotations

import argparse
from typing import Optional, Sequence

from stereo_io import convert_npz_to_pfm, convert_pfm_to_npz


def _parse_args(argv: Optional[Sequence[str]] = None) -> argparse.Namespace:
    parser = argparse.ArgumentParser(description="NPZ / PFM converter")
    parser.add_argument("--input", required=True, type=str, help="input file")
    parser.add_argument("--output", required=True, type=str, help="output file")
    parser.add_argument("--mode", required=True, type=str, choices=["npz2pfm", "pfm2npz"], help="direction")
    parser.add_argument("--key", type=str, default="disparity", help="array key inside the NPZ")
    return parser.parse_args(argv)


def main(argv: Optional[Sequence[str]] = None) -> None:
    args = _parse_args(argv)
    convert = convert_npz_to_pfm if args.mode == "npz2pfm" else convert_pfm_to_npz
    convert(args.input, args.output, key=args.key)


if __name__ == "__main__":
    main()
