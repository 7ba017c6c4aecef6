"""Aggregation + winner-take-all, the end-to-end pipeline and the command line -- drop-in for
the reference's stereo.py.

Signatures, defaults, dtypes (int32 disparity, float32 min_cost), validation and the progress
callback contract follow /root/reference/stereo.py:94-245.  compute_disparity runs the fused
GPU path: the (H, W, D) cost volume is never materialised.  The command line (same flags, run
directory layout, params.json / metrics.json / metrics_summary.json schemas as
/root/reference/stereo.py:304-812) additionally deals the scenes of --all-datasets to the ranks
of a torch.distributed launch, one GPU each.
"""
from __future__ import annotations

import argparse
import csv
import json
import os
import sys
from dataclasses import dataclass
from datetime import datetime
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

import _native
from census import compute_wct_cost_volume  # noqa: F401  (re-exported like the reference)
from filters import bilateral_filter, gaussian_filter, median_filter  # noqa: F401
from guided_filter import GuidedFilterPrecomputed, guided_filter_with_precompute, prepare_guided_filter  # noqa: F401
from stereo_io import ensure_same_shape, read_image, read_pfm, read_pgm_mask, save_disparity_npz, to_gray

DEFAULT_WCT_RADIUS: int = 4
DEFAULT_BASE_WEIGHT: float = 8.0
DEFAULT_GUIDED_RADIUS: int = 3
DEFAULT_GUIDED_EPS: float = 0.01
DEFAULT_FILTER_TYPE: str = "guided"
DEFAULT_MEDIAN_RADIUS: int = 3
DEFAULT_MEDIAN_METHOD: str = "opencv"
DEFAULT_MEDIAN_BLOCK_ROWS: int = 128
DEFAULT_GAUSSIAN_SIGMA: float = 1.0
DEFAULT_BILATERAL_SIGMA: float = 1.0
DEFAULT_BAD_THRESHOLD: float = 2.0

_SUPPORTED = ("guided", "median", "gaussian", "bilateral")
_LABELS = {"guided": "Guided Filter", "median": "Median Filter", "gaussian": "Gaussian Filter",
           "bilateral": "Bilateral Filter"}


def _print_progress(current: int, total: int, label: str) -> None:
    """Reference stereo.py:33-53."""
    if total <= 0:
        raise ValueError("total must be a positive integer")
    clamped = min(max(current, 0), total)
    percent = (clamped / float(total)) * 100.0
    message = f"{label}: {clamped}/{total} ({percent:5.1f}%)"
    sys.stdout.write(f"\r{message}\n" if clamped >= total else f"\r{message}")
    sys.stdout.flush()


def _filter_params(filter_key: str, wct_radius: int, base_weight: float, guided_radius: int, guided_eps: float,
                   median_radius: int, median_method: str, median_block_rows: int, gaussian_sigma: float,
                   bilateral_sigma: float) -> _native.StereoParams:
    """Validate the active filter's arguments as the reference does at its first layer."""
    kind = _native.FILTER_GUIDED
    if filter_key == "guided":
        if guided_radius <= 0:
            raise ValueError("radius must be a positive integer")
        if guided_eps <= 0:
            raise ValueError("eps must be positive")
    elif filter_key == "median":
        if median_radius <= 0:
            raise ValueError("radius must be a positive integer")
        method_key = median_method.strip().lower()
        if method_key not in ("opencv", "scipy", "vectorized", "naive"):
            raise ValueError("method must be 'opencv', 'scipy', 'vectorized' or 'naive'")
        if method_key == "vectorized" and median_block_rows <= 0:
            raise ValueError("block_rows must be a positive integer")
        kind = _native.FILTER_MEDIAN_U8Q if method_key == "opencv" else _native.FILTER_MEDIAN
    elif filter_key == "gaussian":
        if gaussian_sigma <= 0:
            raise ValueError("sigma must be positive")
        kind = _native.FILTER_GAUSSIAN
    else:
        if bilateral_sigma <= 0:
            raise ValueError("sigma must be positive")
        kind = _native.FILTER_BILATERAL
    return _native.StereoParams(
        wct_radius=int(wct_radius), base_weight=float(base_weight), filter_kind=kind,
        guided_radius=int(guided_radius) if filter_key == "guided" else 1,
        guided_eps=float(guided_eps) if filter_key == "guided" else 1.0,
        median_radius=int(median_radius) if filter_key == "median" else 1,
        gaussian_sigma=float(gaussian_sigma) if filter_key == "gaussian" else 1.0,
        bilateral_sigma=float(bilateral_sigma) if filter_key == "bilateral" else 1.0,
    )


def aggregate_and_wta(
    dsi: np.ndarray,
    guide: np.ndarray,
    guided_radius: int,
    guided_eps: float,
    filter_type: str = DEFAULT_FILTER_TYPE,
    median_radius: int = DEFAULT_MEDIAN_RADIUS,
    median_method: str = DEFAULT_MEDIAN_METHOD,
    median_block_rows: int = DEFAULT_MEDIAN_BLOCK_ROWS,
    gaussian_sigma: float = DEFAULT_GAUSSIAN_SIGMA,
    bilateral_sigma: float = DEFAULT_BILATERAL_SIGMA,
    progress_callback: Optional[Callable[[int, int, str], None]] = None,
) -> Tuple[np.ndarray, np.ndarray]:
    """Filter every disparity layer and keep the running arg-min  (reference stereo.py:94-174).

    Returns (disparity int32 (H, W), min_cost float32 (H, W)).  Ties keep the lower disparity,
    NaN never wins, pixels no layer improves stay (0, +inf).
    """
    dsi = np.asarray(dsi)
    guide = np.asarray(guide)
    if dsi.ndim != 3:
        raise ValueError("dsi must be 3-D (H, W, D)")
    if guide.ndim != 2:
        raise ValueError("guide must be a 2-D gray image")
    if dsi.shape[0] != guide.shape[0] or dsi.shape[1] != guide.shape[1]:
        raise ValueError("dsi and guide sizes differ")
    height, width, dmax = dsi.shape
    filter_key = filter_type.strip().lower()
    if filter_key not in _SUPPORTED:
        raise ValueError(f"filter_type must be one of {_SUPPORTED}")
    min_cost = np.full((height, width), np.inf, dtype=np.float32)
    disparity = np.zeros((height, width), dtype=np.int32)
    if filter_key == "guided" or dmax > 0:
        params = _filter_params(filter_key, 1, 8.0, guided_radius, guided_eps, median_radius, median_method,
                                median_block_rows, gaussian_sigma, bilateral_sigma)
    if dmax == 0 or height == 0 or width == 0:
        return disparity, min_cost
    if dsi.dtype == np.uint8 and params.filter_kind == _native.FILTER_MEDIAN_U8Q:
        # The reference hands uint8 layers to the byte median as they are, without its min/max
        # quantisation (filters.py:158-161), and returns the bytes as float32: that is the exact
        # median of the window with replicated borders.
        params.filter_kind = _native.FILTER_MEDIAN
    dsi_f = np.ascontiguousarray(dsi, dtype=np.float32)
    if guide.dtype == np.float32:
        g, gdt = np.ascontiguousarray(guide), _native.DT_F32
    else:
        g, gdt = np.ascontiguousarray(guide.astype(np.float64)), _native.DT_F64
    _native.require_gpu()
    _native.check(_native.lib().stereo_aggregate_wta_host(
        _native.ptr(dsi_f), _native.ptr(g), gdt, height, width, dmax, params, 0, dmax,
        _native.ptr(disparity), _native.ptr(min_cost)))
    if progress_callback is not None:
        for d in range(dmax):
            progress_callback(d + 1, dmax, _LABELS[filter_key])
    return disparity, min_cost


def compute_disparity(
    left_gray: np.ndarray,
    right_gray: np.ndarray,
    dmax: int,
    wct_radius: int = DEFAULT_WCT_RADIUS,
    base_weight: float = DEFAULT_BASE_WEIGHT,
    guided_radius: int = DEFAULT_GUIDED_RADIUS,
    guided_eps: float = DEFAULT_GUIDED_EPS,
    filter_type: str = DEFAULT_FILTER_TYPE,
    median_radius: int = DEFAULT_MEDIAN_RADIUS,
    median_method: str = DEFAULT_MEDIAN_METHOD,
    median_block_rows: int = DEFAULT_MEDIAN_BLOCK_ROWS,
    gaussian_sigma: float = DEFAULT_GAUSSIAN_SIGMA,
    bilateral_sigma: float = DEFAULT_BILATERAL_SIGMA,
    show_progress: bool = True,
    progress_callback: Optional[Callable[[int, int, str], None]] = None,
    d_begin: int = 0,
    d_end: Optional[int] = None,
) -> Tuple[np.ndarray, np.ndarray]:
    """WCT cost -> filter -> WTA in one fused GPU pass  (reference stereo.py:177-245).

    `d_begin` / `d_end` (not in the reference) restrict the search to a disparity shard; the
    default is the whole range [0, dmax).
    """
    left_gray = np.asarray(left_gray)
    right_gray = np.asarray(right_gray)
    height, width = ensure_same_shape(left_gray, right_gray)
    if progress_callback is not None:
        progress = progress_callback
    elif show_progress:
        progress = _print_progress
    else:
        progress = None
    if dmax <= 0:
        raise ValueError("dmax must be a positive integer")
    if wct_radius <= 0:
        raise ValueError("radius must be a positive integer")
    if progress is not None:  # the cost stage reports first, as compute_wct_cost_volume does
        for d_start in range(0, dmax, 8):
            progress(min(d_start + 8, dmax), dmax, "WCT cost volume")
    filter_key = filter_type.strip().lower()
    if filter_key not in _SUPPORTED:
        raise ValueError(f"filter_type must be one of {_SUPPORTED}")
    params = _filter_params(filter_key, wct_radius, base_weight, guided_radius, guided_eps, median_radius,
                            median_method, median_block_rows, gaussian_sigma, bilateral_sigma)
    if d_end is None:
        d_end = dmax
    if left_gray.dtype != right_gray.dtype:
        common = np.result_type(left_gray.dtype, right_gray.dtype)
        left_gray, right_gray = left_gray.astype(common), right_gray.astype(common)
    l, dt = _native.image_for_census(left_gray)
    r, _ = _native.image_for_census(right_gray)
    min_cost = np.full((height, width), np.inf, dtype=np.float32)
    disparity = np.zeros((height, width), dtype=np.int32)
    if height > 0 and width > 0:
        _native.require_gpu()
        _native.check(_native.lib().stereo_compute_disparity_host(
            _native.ptr(l), _native.ptr(r), dt, height, width, int(dmax), params, int(d_begin), int(d_end),
            _native.ptr(disparity), _native.ptr(min_cost)))
    if progress is not None:
        for d in range(dmax):
            progress(d + 1, dmax, _LABELS[filter_key])
    return disparity, min_cost


def compute_disparity_sharded(
    left_gray: np.ndarray,
    right_gray: np.ndarray,
    dmax: int,
    world: int,
    rank: int,
    local_rank: int = 0,
    use_dist: Optional[bool] = None,
    **kwargs,
) -> Tuple[Optional[np.ndarray], Optional[np.ndarray]]:
    """compute_disparity (reference stereo.py:177-245) with the disparity range split over the `world`
    ranks of the initialised torch.distributed process group, one GPU per rank.  Every rank passes the
    same images and keyword arguments (those of compute_disparity); rank 0 returns
    (disparity, min_cost), the other ranks (None, None).  The merged result is the unsharded one bit
    for bit (per-pixel minimum of (cost, disparity) keys).

    Raises NotImplementedError for inputs the device-resident path does not take -- float64 images,
    non-finite guide pixels, parameters without a fused kernel -- so that the caller can run
    compute_disparity on one rank instead.
    """
    import torch

    import sharding

    left_gray = np.asarray(left_gray)
    right_gray = np.asarray(right_gray)
    height, width = ensure_same_shape(left_gray, right_gray)
    if dmax <= 0:
        raise ValueError("dmax must be a positive integer")
    filter_key = str(kwargs.get("filter_type", DEFAULT_FILTER_TYPE)).strip().lower()
    if filter_key not in _SUPPORTED:
        raise ValueError(f"filter_type must be one of {_SUPPORTED}")
    if int(kwargs.get("wct_radius", DEFAULT_WCT_RADIUS)) <= 0:
        raise ValueError("radius must be a positive integer")
    params = _filter_params(
        filter_key, kwargs.get("wct_radius", DEFAULT_WCT_RADIUS), kwargs.get("base_weight", DEFAULT_BASE_WEIGHT),
        kwargs.get("guided_radius", DEFAULT_GUIDED_RADIUS), kwargs.get("guided_eps", DEFAULT_GUIDED_EPS),
        kwargs.get("median_radius", DEFAULT_MEDIAN_RADIUS), kwargs.get("median_method", DEFAULT_MEDIAN_METHOD),
        kwargs.get("median_block_rows", DEFAULT_MEDIAN_BLOCK_ROWS), kwargs.get("gaussian_sigma", DEFAULT_GAUSSIAN_SIGMA),
        kwargs.get("bilateral_sigma", DEFAULT_BILATERAL_SIGMA))
    for img in (left_gray, right_gray):
        # float32 holds uint8 / uint16 / float32 images exactly; wider types would lose comparison order
        if img.dtype not in (np.float32, np.uint8, np.uint16):
            raise NotImplementedError(f"the sharded path takes float32, uint8 or uint16 images, got {img.dtype}")
    if height == 0 or width == 0:
        raise NotImplementedError("empty image")
    left32 = np.ascontiguousarray(left_gray, dtype=np.float32)
    right32 = np.ascontiguousarray(right_gray, dtype=np.float32)
    if not np.isfinite(left32).all():
        raise NotImplementedError("non-finite guide pixels: use compute_disparity, which reproduces the reference's NaN region")
    _native.require_gpu()
    device = torch.device("cuda", local_rank)
    search = sharding.ShardedFrameSearch(height, width, int(dmax), params, world, rank, device, use_dist=use_dist)
    left_dev = torch.from_numpy(left32).to(device).view(-1)
    right_dev = torch.from_numpy(right32).to(device).view(-1)
    disparity, min_cost = search.run(left_dev, right_dev)
    torch.cuda.synchronize(device)
    search.check()   # collective: a timed-out band wait on any rank raises on every rank
    if rank != 0:
        return None, None
    return disparity.view(height, width).cpu().numpy(), min_cost.view(height, width).cpu().numpy()


# --------------------------------------------------------------------------------------------
# Result images
# --------------------------------------------------------------------------------------------
def _save_disparity_image(disparity: np.ndarray, dmax: int, path: str) -> None:
    """8-bit gray PNG of d / (dmax - 1) * 255  (reference stereo.py:248-263)."""
    from PIL import Image

    if dmax <= 0:
        raise ValueError("dmax must be a positive integer")
    with np.errstate(divide="ignore", invalid="ignore"):
        scaled = (disparity.astype(np.float32) / float(dmax - 1)) * 255.0
        Image.fromarray(scaled.astype(np.uint8)).save(path)


def _jet_colormap(values: np.ndarray) -> np.ndarray:
    """Piecewise-linear Jet map of values in [0, 1] -> RGB in [0, 1]  (reference stereo.py:266-282)."""
    if values.ndim < 2:
        raise ValueError("values must be at least 2-D")
    v4 = 4.0 * np.clip(values.astype(np.float32), 0.0, 1.0)
    ramps = [np.clip(np.minimum(v4 - rise, -v4 + fall), 0.0, 1.0) for rise, fall in ((1.5, 4.5), (0.5, 3.5), (-0.5, 2.5))]
    return np.stack(ramps, axis=-1)


def _save_disparity_color_image(disparity: np.ndarray, dmax: int, path: str) -> None:
    """Jet-coloured PNG  (reference stereo.py:285-301)."""
    from PIL import Image

    if dmax <= 0:
        raise ValueError("dmax must be a positive integer")
    with np.errstate(divide="ignore", invalid="ignore"):
        rgb = _jet_colormap(disparity.astype(np.float32) / float(dmax - 1)) * 255.0
        Image.fromarray(rgb.astype(np.uint8)).save(path)


# --------------------------------------------------------------------------------------------
# Dataset layout
# --------------------------------------------------------------------------------------------
def _read_ndisp_summary(path: Path) -> Dict[str, int]:
    """scene -> ndisp from ndisp_summary.csv; only "*-perfect" rows with an integer ndisp count
    (reference stereo.py:304-332)."""
    if not path.exists():
        raise FileNotFoundError(f"ndisp_summary.csv not found: {path}")
    table: Dict[str, int] = {}
    with path.open("r", encoding="utf-8") as handler:
        for row in csv.DictReader(handler):
            scene = (row.get("scene") or "").strip()
            raw = (row.get("ndisp") or "").strip()
            if not scene or not raw or not scene.endswith("-perfect"):
                continue
            try:
                table[scene] = int(raw)
            except ValueError:
                continue
    if not table:
        raise ValueError("ndisp_summary.csv is empty or lists no usable scene")
    return table


def _resolve_dataset_paths(dataset_root: Path, scene: str) -> Tuple[Path, Path, Path, Path]:
    """Middlebury 2014 file names of one scene  (reference stereo.py:335-350)."""
    folder = dataset_root / scene
    return folder / "im0.png", folder / "im1.png", folder / "disp0.pfm", folder / "disp0-n.pgm"


def _resolve_scene_inputs(dataset_root: Path, scene: str, ndisp_map: Dict[str, int],
                          dmax_override: int) -> Tuple[Path, Path, Path, Path, int]:
    """Paths and dmax of one scene; every file must exist  (reference stereo.py:353-383)."""
    paths = _resolve_dataset_paths(dataset_root, scene)
    for what, p in zip(("left image", "right image", "ground truth", "ground-truth mask"), paths):
        if not p.exists():
            raise FileNotFoundError(f"{what} not found: {p}")
    if dmax_override > 0:
        return (*paths, dmax_override)
    if scene not in ndisp_map:
        raise ValueError(f"ndisp_summary.csv has no dmax for scene {scene}; pass --dmax")
    return (*paths, int(ndisp_map[scene]))


# --------------------------------------------------------------------------------------------
# Command line
# --------------------------------------------------------------------------------------------
def _parse_args(argv: Optional[Sequence[str]] = None) -> argparse.Namespace:
    """Flags and defaults of the reference's command line  (reference stereo.py:386-418)."""
    parser = argparse.ArgumentParser(description="Stereo Matching (WCT + Guided Filter + WTA) on MI355X")
    parser.add_argument("--left", type=str, default="", help="left image path")
    parser.add_argument("--right", type=str, default="", help="right image path")
    parser.add_argument("--dmax", type=int, default=0, help="number of disparities")
    parser.add_argument("--dataset", type=str, default="", help="scene folder under dataset/")
    parser.add_argument("--all-datasets", action="store_true", help="run every scene under dataset/")
    parser.add_argument("--wct_radius", type=int, default=DEFAULT_WCT_RADIUS, help="WCT radius")
    parser.add_argument("--base_weight", type=float, default=DEFAULT_BASE_WEIGHT, help="WCT base weight")
    parser.add_argument("--guided_radius", type=int, default=DEFAULT_GUIDED_RADIUS, help="guided filter radius")
    parser.add_argument("--guided_eps", type=float, default=DEFAULT_GUIDED_EPS, help="guided filter epsilon")
    parser.add_argument("--filter", type=str, default=DEFAULT_FILTER_TYPE, choices=list(_SUPPORTED),
                        help="aggregation filter")
    parser.add_argument("--median_radius", type=int, default=DEFAULT_MEDIAN_RADIUS, help="median filter radius")
    parser.set_defaults(median_method=DEFAULT_MEDIAN_METHOD)
    parser.add_argument("--median_block_rows", type=int, default=DEFAULT_MEDIAN_BLOCK_ROWS,
                        help="median filter block rows")
    parser.add_argument("--gaussian_sigma", type=float, default=DEFAULT_GAUSSIAN_SIGMA, help="gaussian sigma")
    parser.add_argument("--bilateral_sigma", type=float, default=DEFAULT_BILATERAL_SIGMA, help="bilateral sigma")
    parser.add_argument("--gt", type=str, default="", help="ground-truth PFM path")
    parser.add_argument("--gt-mask", type=str, default="", help="ground-truth mask PGM path")
    parser.add_argument("--bad_threshold", type=float, default=DEFAULT_BAD_THRESHOLD, help="bad pixel threshold")
    return parser.parse_args(argv)


def _create_run_directory(base_dir: str, timestamp: str) -> Path:
    """<base_dir>/<YYYYMMDDHHMMSS>; on a clash <YYYYMMDD>_<NN>_<HHMMSS> so the name still ends with
    the time  (reference stereo.py:421-450)."""
    root = Path(base_dir)
    root.mkdir(parents=True, exist_ok=True)
    date_part, time_part = timestamp[:-6], timestamp[-6:]
    candidate = root / f"{date_part}{time_part}"
    serial = 0
    while candidate.exists():
        serial += 1
        prefix = f"{date_part}_" if date_part else ""
        candidate = root / f"{prefix}{serial:02d}_{time_part}"
    candidate.mkdir(parents=True, exist_ok=False)
    return candidate


_FILTER_METADATA = {
    "guided": ("guided_radius", "guided_eps"),
    "median": ("median_radius", "median_method", "median_block_rows"),
    "gaussian": ("gaussian_sigma",),
    "bilateral": ("bilateral_sigma",),
}


def _build_run_metadata(args: argparse.Namespace, run_dir: Path, output_gray: Path, output_color: Path,
                        output_npz: Path, output_metrics: Optional[Path], resolved_left: Path,
                        resolved_right: Path, resolved_gt: Path, resolved_gt_mask: Path, resolved_dmax: int,
                        dataset_name: str, all_datasets: bool) -> Dict[str, str]:
    """params.json content: all values as strings, only the active filter's settings
    (reference stereo.py:453-515)."""
    meta: Dict[str, str] = {
        "timestamp": run_dir.name, "run_dir": str(run_dir), "left": str(resolved_left),
        "right": str(resolved_right), "dmax": str(resolved_dmax), "wct_radius": str(args.wct_radius),
        "base_weight": str(args.base_weight), "filter": str(args.filter),
        "bad_threshold": str(args.bad_threshold), "eval": str(True), "gt": str(resolved_gt),
        "gt_mask": str(resolved_gt_mask), "dataset": dataset_name, "all_datasets": str(bool(all_datasets)),
        "progress": str(True), "output_disparity_png": str(output_gray),
        "output_disparity_color_png": str(output_color), "output_disparity_npz": str(output_npz),
        "output_metrics_json": "" if output_metrics is None else str(output_metrics),
    }
    for name in _FILTER_METADATA.get(str(args.filter).strip().lower(), ()):
        meta[name] = str(getattr(args, name))
    return meta


def _write_json(path: Path, data: Dict[str, object]) -> None:
    """Reference stereo.py:586-589 (also used for stereo.py:518-529 and 580-583)."""
    with path.open("w", encoding="utf-8") as handler:
        json.dump(data, handler, ensure_ascii=True, indent=2, sort_keys=True)


_write_run_metadata = _write_json
_write_metrics = _write_json


def _compute_pbm_rms(disparity: np.ndarray, ground_truth: np.ndarray, valid_mask: np.ndarray,
                     bad_threshold: float) -> Dict[str, float]:
    """PBM (percent of valid pixels with |d - gt| > threshold) and RMS over pixels whose ground
    truth is finite and whose mask is set  (reference stereo.py:532-577).  One GPU reduction;
    the counts are exact, the squared differences are float32 like the reference's and are
    summed in float64, where NumPy's mean of a float32 array sums pairwise in float32: the rms
    agrees with the reference's to about 1e-7 relative, not to the last bit."""
    disparity, ground_truth, valid_mask = np.asarray(disparity), np.asarray(ground_truth), np.asarray(valid_mask)
    if disparity.shape != ground_truth.shape:
        raise ValueError("disparity and ground_truth sizes differ")
    if disparity.ndim != 2 or ground_truth.ndim != 2:
        raise ValueError("disparity and ground_truth must be 2-D")
    if valid_mask.shape != ground_truth.shape:
        raise ValueError("valid_mask and ground_truth sizes differ")
    if valid_mask.ndim != 2:
        raise ValueError("valid_mask must be 2-D")
    if bad_threshold <= 0:
        raise ValueError("bad_threshold must be positive")
    height, width = disparity.shape
    if height == 0 or width == 0:
        raise ValueError("ground_truth has no valid pixel")
    d = np.ascontiguousarray(disparity, dtype=np.int32)
    g = np.ascontiguousarray(ground_truth, dtype=np.float32)
    m = np.ascontiguousarray(valid_mask.astype(bool)).view(np.uint8)
    out = np.zeros(3, dtype=np.float64)
    _native.require_gpu()
    _native.check(_native.lib().stereo_pbm_rms_host(_native.ptr(d), _native.ptr(g), _native.ptr(m), height, width,
                                                    float(bad_threshold), _native.ptr(out)))
    return {"pbm": float(out[0]), "rms": float(out[1]), "bad_threshold": float(bad_threshold),
            "valid_pixel_count": float(out[2])}


def _print_metrics(metrics: Dict[str, float]) -> None:
    """Reference stereo.py:592-605."""
    nan = float("nan")
    print("Evaluation:\n"
          f"  PBM: {metrics.get('pbm', nan):.4f}%\n"
          f"  RMS: {metrics.get('rms', nan):.4f}\n"
          f"  Bad Threshold: {metrics.get('bad_threshold', nan):.4f}\n"
          f"  Valid Pixel Count: {metrics.get('valid_pixel_count', nan):.0f}")


def _validate_args(args: argparse.Namespace, dataset_root: Path, ndisp_map: Dict[str, int]) -> None:
    """Mutually exclusive input modes  (reference stereo.py:608-639)."""
    explicit = bool(args.left or args.right or args.gt or args.gt_mask)
    if args.all_datasets and args.dataset:
        raise ValueError("--dataset and --all-datasets exclude each other")
    if args.all_datasets:
        if explicit:
            raise ValueError("--all-datasets excludes --left/--right/--gt/--gt-mask")
        if args.dmax > 0:
            raise ValueError("--all-datasets excludes --dmax")
        if not dataset_root.exists():
            raise FileNotFoundError(f"dataset root not found: {dataset_root}")
        if not ndisp_map:
            raise ValueError("ndisp_summary.csv lists no usable scene")
        return
    if args.dataset:
        if explicit:
            raise ValueError("--dataset excludes --left/--right/--gt/--gt-mask")
        if args.dmax < 0:
            raise ValueError("--dmax must be a positive integer")
        if not dataset_root.exists():
            raise FileNotFoundError(f"dataset root not found: {dataset_root}")
        if args.dmax == 0 and args.dataset not in ndisp_map:
            raise ValueError(f"ndisp_summary.csv has no dmax for scene {args.dataset}; pass --dmax")
        return
    if not args.left or not args.right:
        raise ValueError("--left and --right are required without --dataset")
    if args.dmax <= 0:
        raise ValueError("a valid --dmax is required without --dataset")
    if not args.gt:
        raise ValueError("--gt is required without --dataset")
    if not args.gt_mask:
        raise ValueError("--gt-mask is required without --dataset")


def _run_scene(args: argparse.Namespace, run_dir: Path, left_path: Path, right_path: Path, gt_path: Path,
               gt_mask_path: Path, dmax: int, output_metrics: Optional[Path], dataset_name: str,
               all_datasets: bool, print_metrics: bool,
               progress_callback: Optional[Callable[[int, int, str], None]] = None,
               shard: Optional[Tuple[int, int, int]] = None) -> Dict[str, float]:
    """One scene: read, gray, match, write disparity.png / disparity_color.png / disparity.npz,
    evaluate, write metrics.json and params.json  (reference stereo.py:642-709)."""
    run_dir.mkdir(parents=True, exist_ok=True)
    output_gray, output_color, output_npz = (run_dir / n for n in ("disparity.png", "disparity_color.png",
                                                                    "disparity.npz"))
    left_gray = to_gray(read_image(str(left_path)), normalize=True)
    right_gray = to_gray(read_image(str(right_path)), normalize=True)
    matcher_kwargs = dict(
        wct_radius=args.wct_radius, base_weight=args.base_weight, guided_radius=args.guided_radius,
        guided_eps=args.guided_eps, filter_type=args.filter, median_radius=args.median_radius,
        median_method=args.median_method, median_block_rows=args.median_block_rows,
        gaussian_sigma=args.gaussian_sigma, bilateral_sigma=args.bilateral_sigma)
    if shard is not None:
        # (world, rank, local_rank): every rank searches its share of the disparities, rank 0 carries on
        world, rank, local_rank = shard
        try:
            disparity, min_cost = compute_disparity_sharded(left_gray, right_gray, dmax, world, rank, local_rank,
                                                            use_dist=True, **matcher_kwargs)
        except NotImplementedError:
            disparity = min_cost = None
            if rank == 0:
                disparity, min_cost = compute_disparity(left_gray, right_gray, dmax=dmax, show_progress=False,
                                                        **matcher_kwargs)
        if rank != 0:
            return {}
        return _finish_scene(args, run_dir, disparity, min_cost, dmax, output_gray, output_color, output_npz, left_path,
                             right_path, gt_path, gt_mask_path, output_metrics, dataset_name, all_datasets, print_metrics)
    disparity, min_cost = compute_disparity(
        left_gray, right_gray, dmax=dmax, wct_radius=args.wct_radius, base_weight=args.base_weight,
        guided_radius=args.guided_radius, guided_eps=args.guided_eps, filter_type=args.filter,
        median_radius=args.median_radius, median_method=args.median_method,
        median_block_rows=args.median_block_rows, gaussian_sigma=args.gaussian_sigma,
        bilateral_sigma=args.bilateral_sigma, show_progress=progress_callback is None,
        progress_callback=progress_callback)
    return _finish_scene(args, run_dir, disparity, min_cost, dmax, output_gray, output_color, output_npz, left_path,
                         right_path, gt_path, gt_mask_path, output_metrics, dataset_name, all_datasets, print_metrics)


def _finish_scene(args: argparse.Namespace, run_dir: Path, disparity: np.ndarray, min_cost: np.ndarray, dmax: int,
                  output_gray: Path, output_color: Path, output_npz: Path, left_path: Path, right_path: Path,
                  gt_path: Path, gt_mask_path: Path, output_metrics: Optional[Path], dataset_name: str,
                  all_datasets: bool, print_metrics: bool) -> Dict[str, float]:
    """Outputs and evaluation of one matched scene  (reference stereo.py:668-709)."""
    _save_disparity_image(disparity, dmax, str(output_gray))
    _save_disparity_color_image(disparity, dmax, str(output_color))
    save_disparity_npz(str(output_npz), disparity, min_cost)
    metrics = _compute_pbm_rms(disparity, read_pfm(str(gt_path)), read_pgm_mask(str(gt_mask_path)), args.bad_threshold)
    if output_metrics is not None:
        _write_metrics(output_metrics, metrics)
    if print_metrics:
        _print_metrics(metrics)
    _write_run_metadata(run_dir / "params.json", _build_run_metadata(
        args, run_dir, output_gray, output_color, output_npz, output_metrics, left_path, right_path, gt_path,
        gt_mask_path, dmax, dataset_name, all_datasets))
    return metrics


@dataclass
class DatasetProgressState:
    """Whether the two-line batch progress display has been drawn yet  (reference stereo.py:56-60)."""

    rendered: bool = False


def _print_dataset_stage_progress(dataset_current: int, dataset_total: int, stage_current: int, stage_total: int,
                                  label: str, state: DatasetProgressState) -> None:
    """Two-line display: scenes done, and the current stage  (reference stereo.py:63-91)."""
    if dataset_total <= 0:
        raise ValueError("dataset_total must be a positive integer")
    if stage_total <= 0:
        raise ValueError("stage_total must be a positive integer")
    dc = min(max(dataset_current, 0), dataset_total)
    sc = min(max(stage_current, 0), stage_total)
    top = f"Dataset: {dc}/{dataset_total} ({dc / float(dataset_total) * 100.0:5.1f}%)"
    bottom = f"{label}: {sc}/{stage_total} ({sc / float(stage_total) * 100.0:5.1f}%)"
    if state.rendered:
        sys.stdout.write(f"\033[1A\r\033[2K{top}\n\r\033[2K{bottom}")
    else:
        sys.stdout.write(f"{top}\n{bottom}")
        state.rendered = True
    sys.stdout.flush()


def _scene_work(dataset_root: Path, scene: str, ndisp: int) -> int:
    """Pixel-disparity evaluations of a scene, from the PNG header only (used to balance ranks)."""
    from PIL import Image

    try:
        with Image.open(dataset_root / scene / "im0.png") as im:
            return int(im.size[0]) * int(im.size[1]) * int(ndisp)
    except OSError:
        return int(ndisp)


def _run_all_datasets(args: argparse.Namespace, dataset_root: Path, ndisp_map: Dict[str, int], run_dir: Path,
                      world: int, rank: int) -> Optional[Dict[str, object]]:
    """Every scene listed in ndisp_summary.csv that exists on disk; one metrics_summary.json
    (reference stereo.py:722-778).  With world > 1 the scenes are dealt to the ranks by work and
    only the per-scene metric dicts are gathered on rank 0."""
    import sharding

    scenes: List[str] = sorted(s for s in ndisp_map if (dataset_root / s).is_dir())
    if not scenes:
        raise ValueError("no usable scene under the dataset root")
    total = len(scenes)
    state = DatasetProgressState()
    position = {s: i + 1 for i, s in enumerate(scenes)}

    def one(scene: str) -> Dict[str, float]:
        scene_dir = run_dir / scene
        scene_dir.mkdir(parents=True, exist_ok=False)
        left, right, gt, gt_mask, dmax = _resolve_scene_inputs(dataset_root, scene, ndisp_map, 0)
        callback = None
        if world == 1:
            def callback(current: int, stage_total: int, label: str, idx: int = position[scene]) -> None:
                _print_dataset_stage_progress(idx, total, current, stage_total, label, state)
        else:
            def callback(current: int, stage_total: int, label: str) -> None:
                return None
        return _run_scene(args, scene_dir, left, right, gt, gt_mask, dmax, output_metrics=None,
                          dataset_name=scene, all_datasets=True, print_metrics=False, progress_callback=callback)

    sizes = [_scene_work(dataset_root, s, ndisp_map[s]) for s in scenes]
    per_scene = sharding.run_scenes_parallel(scenes, sizes, world, rank, one)
    if state.rendered:
        sys.stdout.write("\n")
        sys.stdout.flush()
    if per_scene is None:
        return None
    per_scene = {s: per_scene[s] for s in scenes}
    pbm = [m["pbm"] for m in per_scene.values()]
    rms = [m["rms"] for m in per_scene.values()]
    summary: Dict[str, object] = {
        "scene_count": len(per_scene), "bad_threshold": float(args.bad_threshold),
        "pbm_mean": float(np.mean(pbm)) if pbm else float("nan"),
        "rms_mean": float(np.mean(rms)) if rms else float("nan"), "scenes": per_scene,
    }
    _write_json(run_dir / "metrics_summary.json", summary)
    return summary


def main(argv: Optional[Sequence[str]] = None, dataset_root: str = "dataset", result_root: str = "result") -> None:
    """Command-line entry  (reference stereo.py:712-812): --left/--right/--gt/--gt-mask/--dmax,
    --dataset <scene>, or --all-datasets; results go to result/<timestamp>/.

    Under `python -m torch.distributed.run --nproc-per-node N stereo.py ...` every rank takes one GPU:
    --all-datasets deals the scenes to the ranks, a single scene is searched by all ranks together,
    each on its share of the disparity range."""
    args = _parse_args(argv)
    root = Path(dataset_root)
    ndisp_map: Dict[str, int] = {}
    if args.all_datasets or args.dataset:
        ndisp_map = _read_ndisp_summary(root / "ndisp_summary.csv")
    _validate_args(args, root, ndisp_map)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    _native.require_gpu()
    _native.check(_native.lib().stereo_hip_init(local_rank))
    # STEREO_FORCE_PROCESS_GROUP=1 takes the multi-rank route with a single rank (rehearsal on one GPU)
    if world > 1 or os.environ.get("STEREO_FORCE_PROCESS_GROUP") == "1":
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        names: List[Optional[str]] = [None]
        if rank == 0:
            names[0] = str(_create_run_directory(result_root, datetime.now().strftime("%Y%m%d%H%M%S")))
        dist.broadcast_object_list(names, src=0)
        run_dir = Path(str(names[0]))
        try:
            if args.all_datasets:
                _run_all_datasets(args, root, ndisp_map, run_dir, world, rank)
            else:
                _run_single(args, root, ndisp_map, run_dir, shard=(world, rank, local_rank))
            dist.barrier()
        finally:
            dist.destroy_process_group()
        return
    run_dir = _create_run_directory(result_root, datetime.now().strftime("%Y%m%d%H%M%S"))
    if args.all_datasets:
        _run_all_datasets(args, root, ndisp_map, run_dir, 1, 0)
        return
    _run_single(args, root, ndisp_map, run_dir)


def _run_single(args: argparse.Namespace, root: Path, ndisp_map: Dict[str, int], run_dir: Path,
                shard: Optional[Tuple[int, int, int]] = None) -> None:
    """--dataset <scene> or explicit --left/--right/--gt/--gt-mask  (reference stereo.py:779-812)."""
    if args.dataset:
        left, right, gt, gt_mask, dmax = _resolve_scene_inputs(root, args.dataset, ndisp_map, args.dmax)
        name = args.dataset
    else:
        left, right, gt, gt_mask, dmax, name = (Path(args.left), Path(args.right), Path(args.gt), Path(args.gt_mask),
                                                args.dmax, "")
    _run_scene(args, run_dir, left, right, gt, gt_mask, dmax, output_metrics=run_dir / "metrics.json",
               dataset_name=name, all_datasets=False, print_metrics=shard is None or shard[1] == 0, shard=shard)


if __name__ == "__main__":
    main()
