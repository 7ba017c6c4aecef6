"""Pin the CPU oracle (oracle/stereo_oracle.c) to golden vectors captured from the
reference's own Python source (tests/golden/make_golden.py).  CPU only."""
from __future__ import annotations

import numpy as np
import pytest

from conftest import load_golden


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def test_offsets_weights_known_answers(oracle):
    g = load_golden("offsets_weights")
    off4 = oracle.generate_offsets(4)
    off8 = oracle.generate_offsets(8)
    assert np.array_equal(off4, g["off4"]) and np.array_equal(off8, g["off8"])
    # census.py:31-45: first five offsets and the weight ladder
    assert off4[:5].tolist() == [[1, 0, 1], [2, 0, 2], [3, 0, 3], [4, 0, 4], [-1, 0, 1]]
    w4 = oracle.compute_weights(off4, 8.0)
    w8 = oracle.compute_weights(off8, 8.0)
    assert np.array_equal(w4, g["w4"]) and np.array_equal(w8, g["w8"])
    assert np.array_equal(oracle.compute_weights(off4, 3.3), g["w_nd"])
    assert oracle.large_value(w4) == float(g["large4"]) == 1200.0
    assert oracle.large_value(w8) == float(g["large8"]) == 1275.0
    with pytest.raises(ValueError):
        oracle.generate_offsets(0)


@pytest.mark.parametrize("name", ["guided_48x80x16", "guided_64x96x32", "guided_40x64x12_gr5"])
def test_guided_pipeline_bit_exact(oracle, name):
    g = load_golden(name)
    dmax, r_c = int(g["dmax"]), int(g["wct_radius"])
    dsi = oracle.wct_cost_volume(g["left"], g["right"], dmax, radius=r_c)
    assert dsi.dtype == np.float32 and np.array_equal(dsi, g["dsi"])
    gr, eps = int(g["guided_radius"]), float(g["guided_eps"])
    _, mg, vg = oracle.guided_prepare(g["left"], gr)
    assert np.array_equal(mg, g["mean_guide"]) and np.array_equal(vg, g["var_g"])
    disp, cost, layers = oracle.aggregate_and_wta(dsi, g["left"], oracle.FILTER_GUIDED, gr, eps, want_layers=True)
    if "guided_layers" in g.files:
        assert np.array_equal(layers, g["guided_layers"])
    assert np.array_equal(disp, g["disparity"]) and disp.dtype == np.int32
    assert np.array_equal(cost, g["min_cost"]) and cost.dtype == np.float32
    d2, c2 = oracle.compute_disparity(g["left"], g["right"], dmax, wct_radius=r_c, radius=gr, eps_or_sigma=eps)
    assert np.array_equal(d2, disp) and np.array_equal(c2, cost)


def test_query_returns_the_filtered_cost_at_the_queried_disparity(oracle):
    """oracle.compute_disparity_query against the reference's own filtered layers (golden): the cost it reports for
    a queried disparity per pixel is that layer's value, bit for bit; out-of-range queries give NaN."""
    g = load_golden("guided_48x80x16")
    dmax, r_c, gr, eps = int(g["dmax"]), int(g["wct_radius"]), int(g["guided_radius"]), float(g["guided_eps"])
    layers = g["guided_layers"]
    rng = np.random.default_rng(5)
    query = rng.integers(0, dmax, size=g["left"].shape).astype(np.int32)
    query[0, :5] = [-1, dmax, dmax + 7, 0, dmax - 1]
    d, c, fq = oracle.compute_disparity_query(g["left"], g["right"], dmax, query, wct_radius=r_c, radius=gr, eps=eps, threads=3)
    assert np.array_equal(d, g["disparity"]) and np.array_equal(c, g["min_cost"])
    inside = (query >= 0) & (query < dmax)
    want = np.take_along_axis(layers, np.clip(query, 0, dmax - 1)[None], axis=0)[0]
    assert np.array_equal(fq[inside], want[inside]) and np.all(np.isnan(fq[~inside])) and (~inside).sum() == 3
    # queried at the winner, it is the minimum itself
    _, _, fmin = oracle.compute_disparity_query(g["left"], g["right"], dmax, d, wct_radius=r_c, radius=gr, eps=eps)
    assert np.array_equal(fmin, c)


def test_operators_bit_exact(oracle):
    g = load_golden("operators_48x80")
    assert np.array_equal(oracle.integral_image(g["left"]), g["integral_left"])
    assert np.array_equal(oracle.integral_image(g["layer"]), g["integral_layer"])
    assert np.array_equal(oracle.box_filter_mean(g["left"], 3), g["box3_left"])
    assert np.array_equal(oracle.box_filter_mean(g["left"], 0), g["box0_left"])
    assert np.array_equal(oracle.box_filter_mean(g["layer"], 9), g["box9_layer"])
    assert np.array_equal(oracle.guided_filter(g["left"], g["layer"], 3, 0.0154), g["guided_layer"])
    assert np.array_equal(oracle.guided_filter(g["rnd_guide"], g["rnd_src"], 2, 1e-3), g["guided_rnd_r2"])


def test_filters_single_layer(oracle):
    g = load_golden("filters_48x80")
    assert np.array_equal(oracle.median_filter(g["layer"], 3), g["median_r3"])
    assert np.array_equal(oracle.median_filter(g["rnd_src"], 1), g["median_rnd_r1"])
    assert np.array_equal(oracle.median_filter_u8q(g["layer"], 3), g["median_opencv_pin_r3"])
    # Gaussian kernel: np.exp on float32 (SIMD polynomial) vs libm expf may differ in the last ulp
    assert rel_err(oracle.gaussian_kernel(3, 1.0), g["gaussian_kernel_r3_s1"]) < 5e-7
    assert rel_err(oracle.gaussian_filter(g["layer"], 1.0), g["gaussian_s1"]) < 1e-6
    assert rel_err(oracle.gaussian_filter(g["rnd_src"], 0.7), g["gaussian_rnd_s07"]) < 1e-6
    assert rel_err(oracle.bilateral_filter(g["layer"], 1.0), g["bilateral_s1"]) < 2e-6
    assert rel_err(oracle.bilateral_filter(g["rnd_src"], 1.5), g["bilateral_rnd_s15"]) < 2e-6


def _wta_close(oracle, dsi, guide, kind, radius, param, disp_g, cost_g, tol):
    """Float filters: costs within tol, and every disparity mismatch must sit on a near-tie:
    the oracle's own filtered cost at the golden disparity is within tol of the oracle's min."""
    disp, cost, layers = oracle.aggregate_and_wta(dsi, guide, kind, radius, param, want_layers=True)
    assert rel_err(cost, cost_g) < tol
    bad = np.argwhere(disp != disp_g)
    for y, x in bad:
        at_golden = float(layers[disp_g[y, x], y, x])
        assert abs(at_golden - float(cost[y, x])) <= 1e-5 * max(1.0, abs(float(cost[y, x]))), (y, x)
    return 1.0 - len(bad) / disp.size


def test_filters_through_wta(oracle):
    g = load_golden("wta_filters_48x80x16")
    dsi, left = g["dsi"], g["left"]
    d, c = oracle.aggregate_and_wta(dsi, left, oracle.FILTER_MEDIAN, 3, 0.0)
    assert np.array_equal(d, g["disp_median"]) and np.array_equal(c, g["cost_median"])
    d, c = oracle.aggregate_and_wta(dsi, left, oracle.FILTER_MEDIAN_U8Q, 3, 0.0)
    assert np.array_equal(d, g["disp_median_opencv_pin"]) and np.array_equal(c, g["cost_median_opencv_pin"])
    assert _wta_close(oracle, dsi, left, oracle.FILTER_GAUSSIAN, 0, 1.0,
                      g["disp_gaussian_pin"], g["cost_gaussian_pin"], 1e-6) >= 0.999
    # bilateral: np.exp (SIMD polynomial) vs libm expf differ in the last ulp, which flips
    # exact-tie pixels; every flip is checked to be a near-tie by _wta_close
    assert _wta_close(oracle, dsi, left, oracle.FILTER_BILATERAL, 0, 1.0,
                      g["disp_bilateral"], g["cost_bilateral"], 2e-6) >= 0.95


def test_bilateral_larger_frame(oracle):
    """The reference's bilateral pipeline on a 120 x 200 frame (tests/golden/make_golden_bilateral.py): the oracle's
    costs within 1e-6; its disparities identical away from the left border, and every difference near the border a
    near-tie (libm's expf against NumPy's exp in the last place, on layers that are all `large` there)."""
    g = load_golden("bilateral_120x200x20")
    dmax = int(g["dmax"])
    dsi = oracle.wct_cost_volume(g["left"], g["right"], dmax)
    agree = _wta_close(oracle, dsi, g["left"], oracle.FILTER_BILATERAL, 0, 1.0, g["disp_bilateral"], g["cost_bilateral"], 2e-6)
    assert agree >= 0.99
    d = oracle.aggregate_and_wta(dsi, g["left"], oracle.FILTER_BILATERAL, 0, 1.0)[0]
    inner = (slice(8, -8), slice(dmax + 8, -8))
    assert np.mean(d[inner] == g["disp_bilateral"][inner]) >= 0.9999


def test_radius8(oracle):
    g = load_golden("radius8_40x72x12")
    dsi = oracle.wct_cost_volume(g["left"], g["right"], 12, radius=8)
    assert np.array_equal(dsi, g["dsi"])
    d, c = oracle.aggregate_and_wta(dsi, g["left"], oracle.FILTER_GUIDED, 3, 0.0154)
    assert np.array_equal(d, g["disp_guided"]) and np.array_equal(c, g["cost_guided"])
    d, c = oracle.aggregate_and_wta(dsi, g["left"], oracle.FILTER_MEDIAN_U8Q, 3, 0.0)
    assert np.array_equal(d, g["disp_median_opencv_pin"]) and np.array_equal(c, g["cost_median_opencv_pin"])
    assert _wta_close(oracle, dsi, g["left"], oracle.FILTER_GAUSSIAN, 0, 1.0,
                      g["disp_gaussian_pin"], g["cost_gaussian_pin"], 1e-6) >= 0.999
    assert _wta_close(oracle, dsi, g["left"], oracle.FILTER_BILATERAL, 0, 1.0,
                      g["disp_bilateral"], g["cost_bilateral"], 2e-6) >= 0.95


def test_edge_cases(oracle):
    g = load_golden("edge_cases")
    dsi = oracle.wct_cost_volume(g["wide_left"], g["wide_right"], 28)
    assert np.array_equal(dsi, g["wide_dsi"])
    assert np.all(dsi[:, :, 20:] == 1200.0)  # layers d >= W are entirely `large`
    d, c = oracle.aggregate_and_wta(dsi, g["wide_left"], oracle.FILTER_GUIDED, 3, 0.0154)
    assert np.array_equal(d, g["wide_disp"]) and np.array_equal(c, g["wide_cost"])

    dsi = oracle.wct_cost_volume(g["u8_left"], g["u8_right"], 8)
    assert np.array_equal(dsi, g["u8_dsi"])
    d, c = oracle.aggregate_and_wta(dsi, g["u8_left"], oracle.FILTER_GUIDED, 3, 0.0154)
    assert np.array_equal(d, g["u8_disp"]) and np.array_equal(c, g["u8_cost"])

    dsi = oracle.wct_cost_volume(g["const_img"], g["const_img"], 4)
    assert np.array_equal(dsi, g["const_dsi"])
    d, c = oracle.aggregate_and_wta(dsi, g["const_img"], oracle.FILTER_GUIDED, 3, 0.0154)
    assert np.array_equal(d, g["const_disp"]) and np.array_equal(c, g["const_cost"])

    dsi = oracle.wct_cost_volume(g["nd_left"], g["nd_right"], 8, base_weight=3.3)
    assert np.array_equal(dsi, g["nd_dsi"])
    d, c = oracle.aggregate_and_wta(dsi, g["nd_left"], oracle.FILTER_GUIDED, 3, 0.0154)
    assert np.array_equal(d, g["nd_disp"]) and np.array_equal(c, g["nd_cost"])

    assert np.array_equal(oracle.wct_cost_volume(g["f64_left"], g["f64_right"], 6), g["f64_dsi"])
    assert np.array_equal(oracle.wct_cost_volume(g["thin_img"], g["thin_img"], 3), g["thin_dsi"])
    assert np.all(g["thin_dsi"] == 1200.0)

    dsi = oracle.wct_cost_volume(g["small_left"], g["small_right"], 3, radius=1)
    assert np.array_equal(dsi, g["small_dsi"])
    d, c = oracle.aggregate_and_wta(dsi, g["small_left"], oracle.FILTER_GUIDED, 3, 0.0154)
    assert np.array_equal(d, g["small_disp"]) and np.array_equal(c, g["small_cost"])


def test_wta_semantics(oracle):
    g = load_golden("wta_semantics")
    layers = g["layers"]
    h, w, dmax = layers.shape
    min_cost = np.full((h, w), np.inf, dtype=np.float32)
    disp = np.zeros((h, w), dtype=np.int32)
    import ctypes
    for d in range(dmax):
        layer = np.ascontiguousarray(layers[:, :, d])
        oracle.lib().orc_wta_update(layer.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(h * w), ctypes.c_int(d),
                                    min_cost.ctypes.data_as(ctypes.c_void_p), disp.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(disp, g["disparity"])
    assert np.array_equal(min_cost, g["min_cost"], equal_nan=True)
    assert disp[2, 2] == 1  # tie between d=1 and d=2 keeps the lower disparity
    assert disp[0, 0] == 0 and np.isinf(min_cost[0, 0])  # all-inf pixel never updates
    assert disp[1, 1] == 0 and np.isinf(min_cost[1, 1])  # all-NaN pixel never updates


def test_roll_known_answer(oracle):
    # right = roll(left, -k): dsi[y, x, k] == 0 on the valid interior, WTA mode == k
    rng = np.random.default_rng(5)
    left = rng.random((32, 48)).astype(np.float32)
    k, r = 3, 4
    right = np.roll(left, -k, axis=1)
    dsi = oracle.wct_cost_volume(left, right, 8)
    assert np.all(dsi[r:32 - r, k + r:48 - r - k, k] == 0.0)
    assert np.all(dsi[:r] == 1200.0) and np.all(dsi[32 - r:] == 1200.0) and np.all(dsi[:, 48 - r:, :] == 1200.0)
    for d in range(8):
        assert np.all(dsi[:, : d + r, d] == 1200.0)
    disp, _ = oracle.aggregate_and_wta(dsi, left, oracle.FILTER_GUIDED, 3, 0.0154)
    assert np.bincount(disp[8:24, 16:36].ravel()).argmax() == k
    const = np.full((32, 48), 1200.0, dtype=np.float32)
    assert np.all(oracle.guided_filter(left, const, 3, 0.0154) == 1200.0)


def test_all_cores_schedule_is_bit_identical():
    """The all-cores schedule bench.py times as its second CPU figure deals the layers to workers and merges
    their (cost, d) minima; per layer it runs the same arithmetic, so the frame must equal the reference
    schedule's bit for bit (ties included: the synthetic pair has wide constant `large` regions)."""
    import sys

    from conftest import PKG_DIR
    if PKG_DIR not in sys.path:
        sys.path.insert(0, PKG_DIR)
    from oracle import oracle as orc
    from synth import synth_pair_gray

    orc.build()
    left, right, _ = synth_pair_gray(61, 97, 29, seed=11)
    d_ref, c_ref = orc.compute_disparity(left, right, 29, threads=3)
    for threads in (1, 4, 7):
        d_all, c_all = orc.compute_disparity_allcores(left, right, 29, threads=threads)
        assert np.array_equal(d_all, d_ref)
        assert np.array_equal(c_all.view(np.uint32), c_ref.view(np.uint32))
