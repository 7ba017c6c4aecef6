#!/usr/bin/env python3
"""NaN and infinite pixels through the stand-alone operators against the oracle: the pattern of NaN / +inf / -inf
outputs must be identical (the reference's integral images spread a non-finite pixel over everything below and to
the right of it; np.median returns NaN for a window that holds one), finite outputs within 1e-5.  Test
infrastructure: `python tests/fuzz_nonfinite_vs_oracle.py N SEED`; tests/test_gpu_fuzz.py runs a batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import filters  # noqa: E402
import guided_filter as gf  # noqa: E402
import stereo  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def same(a, b, tol=0.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return False, "nan pattern"
    m = ~np.isnan(a)
    if not np.array_equal(np.isinf(a[m]), np.isinf(b[m])) or not np.array_equal(a[m][np.isinf(a[m])], b[m][np.isinf(b[m])]):
        return False, "inf pattern"
    f = m & np.isfinite(a) & np.isfinite(b)
    e = np.max(np.abs(a[f] - b[f]) / np.maximum(1, np.abs(b[f]))) if f.any() else 0.0
    return e <= tol, f"err {e:.2e}"


def run_cases(n_cases: int, seed: int) -> int:
    rng = np.random.default_rng(seed)
    fails = 0
    for case in range(n_cases):
        h, w = int(rng.integers(8, 60)), int(rng.integers(8, 90))
        img = (rng.random((h, w)) * 100).astype(np.float32)
        for _ in range(int(rng.integers(1, 4))):
            img[rng.integers(0, h), rng.integers(0, w)] = rng.choice([np.nan, np.inf, -np.inf])
        op = str(rng.choice(["integral", "box", "guided", "gaussian", "bilateral", "median", "agg"]))
        try:
            if op == "agg":
                # aggregate_and_wta on a volume with non-finite costs: the guided filter spreads them per layer
                nd = int(rng.integers(1, 7))
                dsi = (np.round(rng.random((h, w, nd)) * 400) / 4).astype(np.float32)
                for _ in range(int(rng.integers(1, 3))):
                    dsi[rng.integers(0, h), rng.integers(0, w), rng.integers(0, nd)] = rng.choice([np.nan, np.inf, -np.inf])
                guide = rng.random((h, w)).astype(np.float32)
                filt = str(rng.choice(["guided", "gaussian", "bilateral"]))
                kind = {"guided": orc.FILTER_GUIDED, "gaussian": orc.FILTER_GAUSSIAN, "bilateral": orc.FILTER_BILATERAL}[filt]
                dref, cref = orc.aggregate_and_wta(dsi, guide, kind, 3 if filt == "guided" else 0, 0.0154 if filt == "guided" else 1.0)[:2]
                d, c = stereo.aggregate_and_wta(dsi, guide, 3, 0.0154, filter_type=filt)
                ok, why = same(np.where(d == dref, c, cref), cref, 1e-5)
                if ok and np.mean(d == dref) < 0.98:
                    ok, why = False, f"disparity agreement {np.mean(d == dref):.3f}"
                why = f"{filt} {why}"
            elif op == "integral":
                ok, why = same(gf.integral_image(img), orc.integral_image(img), 1e-12)
            elif op == "box":
                r = int(rng.integers(0, 5)); ok, why = same(gf.box_filter_mean(img, r), orc.box_filter_mean(img, r), 1e-11)
            elif op == "guided":
                src = (rng.random((h, w)) * 50).astype(np.float32); r = int(rng.integers(1, 5))
                g = img / 100 if rng.random() < 0.5 else (rng.random((h, w))).astype(np.float32)
                s2 = src if rng.random() < 0.5 else img
                ok, why = same(gf.guided_filter(g, s2, r, 0.01), orc.guided_filter(g, s2, r, 0.01), 1e-5)
            elif op == "gaussian":
                ok, why = same(filters.gaussian_filter(img, 1.0), orc.gaussian_filter(img, 1.0), 1e-5)
            elif op == "bilateral":
                ok, why = same(filters.bilateral_filter(img, 1.0), orc.bilateral_filter(img, 1.0), 1e-5)
            else:
                r = int(rng.integers(1, 4)); ok, why = same(filters.median_filter(img, r, method="naive"), orc.median_filter(img, r))
        except Exception as e:
            ok, why = False, f"{type(e).__name__}: {e}"
        if not ok:
            fails += 1
            print("FAIL", case, op, f"{w}x{h}", why, flush=True)
    return fails


def main() -> None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    fails = run_cases(n, int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    print(f"{n} cases, {fails} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
