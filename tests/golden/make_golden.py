#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference itself.

Run in the build container only (the reference tree is not present on the GPU
box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference modules (census.py, guided_filter.py, filters.py, stereo.py) are
imported unmodified from /root/reference.  `numba` is not installed, so an
identity stand-in (tests/golden/_numba_standin/numba.py) supplies `njit` and
`prange`; both reference decorators use fastmath=False, so plain CPython
evaluation is IEEE-identical to the JIT.  `cv2` is not installed either: the
`opencv` median path is pinned as the reference's own uint8 quantisation
(filters.py:139-166) + an exact uint8 median with replicate border
(scipy.ndimage.median_filter, mode="nearest") + the reference's rescale
(filters.py:177-180); the `opencv` Gaussian path is pinned to the reference's
`naive` path (filters.py:191-204).  Those two pins are labelled *_pin in the
fixtures.

Only inputs and outputs are stored (.npz, no pickles).
"""
from __future__ import annotations

import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("STEREO_REFERENCE_DIR", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_numba_standin"))
sys.path.insert(0, REFERENCE)
sys.path.insert(0, os.path.join(REPO, "stereo-matching_amd"))

import numpy as np  # noqa: E402
import scipy.ndimage  # noqa: E402

import census as ref_census  # noqa: E402
import filters as ref_filters  # noqa: E402
import guided_filter as ref_gf  # noqa: E402
import stereo as ref_stereo  # noqa: E402

assert ref_census.__file__.startswith(REFERENCE), ref_census.__file__
assert ref_stereo.__file__.startswith(REFERENCE), ref_stereo.__file__

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location(
    "amd_synth", os.path.join(REPO, "stereo-matching_amd", "synth.py")
)
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)


def wta_loop(layers):
    """The reference's streaming WTA (stereo.py:134-135,167-170) over a list of layers."""
    h, w = layers[0].shape
    min_cost = np.full((h, w), np.inf, dtype=np.float32)
    disparity = np.zeros((h, w), dtype=np.int32)
    for d, filtered in enumerate(layers):
        better = filtered < min_cost
        if np.any(better):
            min_cost = np.where(better, filtered, min_cost)
            disparity[better] = d
    return disparity, min_cost


def median_opencv_pin(layer: np.ndarray, radius: int) -> np.ndarray:
    prepared, offset, inv_scale = ref_filters._prepare_opencv_median_input(layer)
    k = 2 * radius + 1
    filtered = scipy.ndimage.median_filter(prepared, size=k, mode="nearest")
    if inv_scale == 0.0:
        return np.full(layer.shape, offset, dtype=np.float32)
    return filtered.astype(np.float32) * inv_scale + offset


def save(name: str, **arrays) -> None:
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


def case_guided(name, h, w, dmax, seed, wct_radius=4, g_radius=3, eps=0.0154, keep_layers=True):
    t0 = time.time()
    left, right, d_row = synth.synth_pair_gray(h, w, dmax, seed)
    offsets = np.asarray(ref_census.generate_offsets(wct_radius), dtype=np.int32)
    weights = ref_census.compute_weights(ref_census.generate_offsets(wct_radius), 8.0)
    dsi = ref_census.compute_wct_cost_volume(left, right, dmax, radius=wct_radius)
    pre = ref_gf.prepare_guided_filter(left, g_radius, eps)
    layers = [ref_gf.guided_filter_with_precompute(pre, dsi[:, :, d]) for d in range(dmax)]
    disp, min_cost = ref_stereo.aggregate_and_wta(dsi, left, g_radius, eps)
    d2, m2 = wta_loop(layers)
    assert np.array_equal(disp, d2) and np.array_equal(min_cost, m2)
    out = dict(
        left=left, right=right, d_row=d_row.astype(np.int32), dmax=np.int32(dmax),
        wct_radius=np.int32(wct_radius), guided_radius=np.int32(g_radius), guided_eps=np.float64(eps),
        offsets=offsets, weights=weights, dsi=dsi, disparity=disp, min_cost=min_cost,
        mean_guide=pre.mean_guide, var_g=pre.var_g,
    )
    if keep_layers:
        out["guided_layers"] = np.stack(layers, axis=0)
    save(name, **out)
    print(f"  {name}: {time.time() - t0:.1f}s")
    return left, right, dsi


def main() -> None:
    print("reference:", REFERENCE)

    # ---- known-answer constants (census.py:20-62, :240)
    off4 = np.asarray(ref_census.generate_offsets(4), dtype=np.int32)
    off8 = np.asarray(ref_census.generate_offsets(8), dtype=np.int32)
    w4 = ref_census.compute_weights(ref_census.generate_offsets(4), 8.0)
    w8 = ref_census.compute_weights(ref_census.generate_offsets(8), 8.0)
    w_nd = ref_census.compute_weights(ref_census.generate_offsets(4), 3.3)
    save("offsets_weights", off4=off4, off8=off8, w4=w4, w8=w8, w_nd=w_nd,
         large4=np.float64(float(np.sum(w4)) * 10.0), large8=np.float64(float(np.sum(w8)) * 10.0))

    # ---- main guided-filter pipeline cases
    left, right, dsi = case_guided("guided_48x80x16", 48, 80, 16, seed=1)
    case_guided("guided_64x96x32", 64, 96, 32, seed=2, keep_layers=False)
    case_guided("guided_40x64x12_gr5", 40, 64, 12, seed=5, g_radius=5, eps=0.01)

    # ---- stand-alone operators on the 48x80 case
    layer = np.ascontiguousarray(dsi[:, :, 5])
    layer_nc = dsi[:, :, 5]  # strided view, as stereo.py:146 passes it
    integral = ref_gf._integral_image_numba(left.astype(np.float64))
    integral_layer = ref_gf._integral_image_numba(layer)
    box3 = ref_gf.box_filter_mean(left, 3)
    box0 = ref_gf.box_filter_mean(left, 0)
    box9 = ref_gf.box_filter_mean(layer, 9)
    gf_api = ref_gf.guided_filter(left, layer_nc, 3, 0.0154)
    rng = np.random.default_rng(11)
    rnd_src = (rng.random((48, 80)) * 50.0 - 10.0).astype(np.float32)
    rnd_guide = rng.random((48, 80))  # float64 guide
    gf_rnd = ref_gf.guided_filter(rnd_guide, rnd_src, 2, 1e-3)
    save("operators_48x80",
         left=left, layer=layer, integral_left=integral, integral_layer=integral_layer,
         box3_left=box3, box0_left=box0, box9_layer=box9, guided_layer=gf_api,
         rnd_src=rnd_src, rnd_guide=rnd_guide, guided_rnd_r2=gf_rnd)

    # ---- alternative filters, single layer (filters.py:227-310)
    t0 = time.time()
    med_scipy = ref_filters.median_filter(layer_nc, 3, method="scipy")
    med_vec = ref_filters.median_filter(layer_nc, 3, method="vectorized", block_rows=16)
    med_naive = ref_filters.median_filter(layer_nc, 3, method="naive")
    assert np.array_equal(med_scipy, med_vec) and np.array_equal(med_scipy, med_naive)
    med_r1 = ref_filters.median_filter(rnd_src, 1, method="scipy")
    med_pin = median_opencv_pin(layer_nc, 3)
    u8q, q_off, q_inv = ref_filters._prepare_opencv_median_input(layer_nc)
    gauss = ref_filters.gaussian_filter(layer_nc, 1.0, method="naive")
    gauss_rnd = ref_filters.gaussian_filter(rnd_src, 0.7, method="naive")
    bilat = ref_filters.bilateral_filter(layer_nc, 1.0)
    bilat_rnd = ref_filters.bilateral_filter(rnd_src, 1.5)
    gk1 = ref_filters._gaussian_kernel(3, 1.0)
    save("filters_48x80",
         layer=layer, rnd_src=rnd_src, median_r3=med_scipy, median_rnd_r1=med_r1,
         median_opencv_pin_r3=med_pin, u8_quantised=u8q, q_offset=np.float64(q_off),
         q_inv_scale=np.float64(q_inv), gaussian_s1=gauss, gaussian_rnd_s07=gauss_rnd,
         bilateral_s1=bilat, bilateral_rnd_s15=bilat_rnd, gaussian_kernel_r3_s1=gk1)
    print(f"  filters single layer: {time.time() - t0:.1f}s")

    # ---- alternative filters through the WTA loop (stereo.py:145-172), 48x80x16
    t0 = time.time()
    dmax = dsi.shape[2]
    d_med, m_med = ref_stereo.aggregate_and_wta(dsi, left, 3, 0.0154, filter_type="median",
                                                median_radius=3, median_method="scipy")
    d_medp, m_medp = wta_loop([median_opencv_pin(dsi[:, :, d], 3) for d in range(dmax)])
    d_gau, m_gau = wta_loop([ref_filters.gaussian_filter(dsi[:, :, d], 1.0, method="naive")
                             for d in range(dmax)])
    d_bil, m_bil = ref_stereo.aggregate_and_wta(dsi, left, 3, 0.0154, filter_type="bilateral",
                                                bilateral_sigma=1.0)
    save("wta_filters_48x80x16",
         left=left, right=right, dsi=dsi,
         disp_median=d_med, cost_median=m_med, disp_median_opencv_pin=d_medp,
         cost_median_opencv_pin=m_medp, disp_gaussian_pin=d_gau, cost_gaussian_pin=m_gau,
         disp_bilateral=d_bil, cost_bilateral=m_bil)
    print(f"  filters WTA: {time.time() - t0:.1f}s")

    # ---- census radius 8 (64 offsets), with the non-guided filters
    t0 = time.time()
    h, w, dmax = 40, 72, 12
    l8, r8, _ = synth.synth_pair_gray(h, w, dmax, seed=3)
    dsi8 = ref_census.compute_wct_cost_volume(l8, r8, dmax, radius=8)
    d8_g, m8_g = ref_stereo.aggregate_and_wta(dsi8, l8, 3, 0.0154)
    d8_medp, m8_medp = wta_loop([median_opencv_pin(dsi8[:, :, d], 3) for d in range(dmax)])
    d8_gau, m8_gau = wta_loop([ref_filters.gaussian_filter(dsi8[:, :, d], 1.0, method="naive")
                               for d in range(dmax)])
    d8_bil, m8_bil = ref_stereo.aggregate_and_wta(dsi8, l8, 3, 0.0154, filter_type="bilateral")
    save("radius8_40x72x12", left=l8, right=r8, dsi=dsi8, disp_guided=d8_g, cost_guided=m8_g,
         disp_median_opencv_pin=d8_medp, cost_median_opencv_pin=m8_medp,
         disp_gaussian_pin=d8_gau, cost_gaussian_pin=m8_gau,
         disp_bilateral=d8_bil, cost_bilateral=m8_bil)
    print(f"  radius 8: {time.time() - t0:.1f}s")

    # ---- edge cases
    # dmax > W (layers d >= W are all `large`)
    lw, rw, _ = synth.synth_pair_gray(24, 20, 8, seed=4)
    dsi_w = ref_census.compute_wct_cost_volume(lw, rw, 28)
    dw, mw = ref_stereo.aggregate_and_wta(dsi_w, lw, 3, 0.0154)
    # uint8 inputs straight into the pipeline
    lu, ru, _ = synth.synth_pair_u8(32, 48, 8, seed=6)
    dsi_u = ref_census.compute_wct_cost_volume(lu, ru, 8)
    du, mu = ref_stereo.compute_disparity(lu, ru, 8, guided_eps=0.0154, show_progress=False)
    # constant image
    lc = np.full((20, 24), 0.5, dtype=np.float32)
    dsi_c = ref_census.compute_wct_cost_volume(lc, lc, 4)
    dc, mc = ref_stereo.aggregate_and_wta(dsi_c, lc, 3, 0.0154)
    # non-dyadic base weight (sequential float32 adds matter)
    ln, rn, _ = synth.synth_pair_gray(32, 48, 8, seed=7)
    dsi_n = ref_census.compute_wct_cost_volume(ln, rn, 8, base_weight=3.3)
    dn, mn = ref_stereo.compute_disparity(ln, rn, 8, base_weight=3.3, guided_eps=0.0154,
                                          show_progress=False)
    # float64 inputs whose float32 roundings collide
    rng = np.random.default_rng(8)
    base = rng.integers(0, 4, size=(24, 36)).astype(np.float64)
    l64 = base + rng.random((24, 36)) * 1e-9
    r64 = np.roll(l64, -2, axis=1) + rng.random((24, 36)) * 1e-12
    dsi_64 = ref_census.compute_wct_cost_volume(l64, r64, 6)
    # image smaller than the census window: everything invalid
    lt = rng.random((7, 30)).astype(np.float32)
    dsi_t = ref_census.compute_wct_cost_volume(lt, lt, 3)
    # tiny image with radius-1 census and clipped guided windows everywhere
    ls = rng.random((5, 6)).astype(np.float32)
    rs = rng.random((5, 6)).astype(np.float32)
    dsi_s = ref_census.compute_wct_cost_volume(ls, rs, 3, radius=1)
    ds, ms = ref_stereo.aggregate_and_wta(dsi_s, ls, 3, 0.0154)
    save("edge_cases",
         wide_left=lw, wide_right=rw, wide_dsi=dsi_w, wide_disp=dw, wide_cost=mw,
         u8_left=lu, u8_right=ru, u8_dsi=dsi_u, u8_disp=du, u8_cost=mu,
         const_img=lc, const_dsi=dsi_c, const_disp=dc, const_cost=mc,
         nd_left=ln, nd_right=rn, nd_dsi=dsi_n, nd_disp=dn, nd_cost=mn,
         f64_left=l64, f64_right=r64, f64_dsi=dsi_64,
         thin_img=lt, thin_dsi=dsi_t,
         small_left=ls, small_right=rs, small_dsi=dsi_s, small_disp=ds, small_cost=ms)

    # ---- WTA semantics: ties, NaN, +inf layers (stereo.py:134-135,167-170)
    g = rng.random((6, 7)).astype(np.float32)
    tie = np.zeros((6, 7, 5), dtype=np.float32)
    tie[:, :, 0] = 3.0
    tie[:, :, 1] = 1.0
    tie[:, :, 2] = 1.0
    tie[:, :, 3] = np.nan
    tie[:, :, 4] = np.inf
    tie[0, 0, :] = np.inf
    tie[1, 1, :] = np.nan
    d_t, m_t = wta_loop([tie[:, :, d] for d in range(5)])
    save("wta_semantics", layers=tie, disparity=d_t, min_cost=m_t, guide=g)


if __name__ == "__main__":
    main()
