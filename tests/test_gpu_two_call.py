"""The reference's two-call API on the GPU: census.compute_wct_cost_volume writes the float32 volume
dsi[H][W][D] (census.py:250) and stereo.aggregate_and_wta filters every layer of a CALLER-SUPPLIED volume
(stereo.py:145-170).  With the guided filter at radius 1..4 the volume is first encoded into 16-bit "cost codes"
(one [H][W] layer per disparity) and the two-column kernel of the single-call path filters those; volumes the
codes cannot hold, and radii 5..8, go to the general float64 kernel.  Both routes are checked against the oracle,
and the code route against the single-call path bit for bit (same arithmetic on the same costs)."""
import ctypes

import numpy as np
import pytest

from test_gpu_parity import assert_wta_matches, rel_err

pytestmark = pytest.mark.gpu


def assert_same_as_single_call(d2, c2, d1, c1, dmax, wct_radius=4):
    """Two-call (cost codes) against single-call results of the same frame.  Both run the two-column kernel on the
    same integer costs.  Where every disparity of a tile has a valid match in all the tile's columns -- columns from
    dmax + census radius + the tile width on -- both take the same code variant and the results are bit-identical;
    left of that the single-call path runs its variant with invalid matches where the code path needs none, and
    the two differ by rounding (1e-6)."""
    x0 = dmax + wct_radius + 128
    assert np.array_equal(d2[:, x0:], d1[:, x0:]) and np.array_equal(c2[:, x0:], c1[:, x0:])
    fin = np.isfinite(c1)
    assert np.array_equal(np.isfinite(c2), fin)
    assert rel_err(c2[fin], c1[fin]) <= 1e-6
    assert np.mean(d2 == d1) >= 0.9995


def _oracle_ref(oracle, dsi, guide, rg, eps, d0=0, d1=None):
    d1 = dsi.shape[2] if d1 is None else d1
    dref, cref, layers = oracle.aggregate_and_wta(np.ascontiguousarray(dsi[:, :, d0:d1]), guide, oracle.FILTER_GUIDED, rg, eps,
                                                  want_layers=True)
    return dref, cref, layers


@pytest.mark.parametrize("rg", [1, 2, 3, 4])
@pytest.mark.parametrize("eps", [0.0154, 0.002])     # the fast and the precise variant of the two-column kernel
def test_aggregate_census_volume_matches_oracle_and_single_call(oracle, rg, eps):
    import stereo
    from synth import synth_pair_gray
    h, w, dmax = 150, 420, 24            # tiles away from every border: the passes on full windows
    left, right, _ = synth_pair_gray(h, w, dmax, seed=40 + rg)
    dsi = oracle.wct_cost_volume(left, right, dmax)
    dref, cref, layers = _oracle_ref(oracle, dsi, left, rg, eps)
    d, c = stereo.aggregate_and_wta(dsi, left, rg, eps)
    assert_wta_matches(d, c, dref, cref, layers)
    # the single-call path forms the same integer costs from census words and runs the same arithmetic on them
    d1, c1 = stereo.compute_disparity(left, right, dmax, guided_radius=rg, guided_eps=eps, show_progress=False)
    assert_same_as_single_call(d, c, d1, c1, dmax)


@pytest.mark.parametrize("h,w,dmax", [(1, 1, 1), (9, 7, 3), (37, 53, 23), (64, 129, 5), (33, 130, 70)])
def test_aggregate_small_and_ragged_shapes(oracle, h, w, dmax):
    """Frames smaller than a tile, widths that straddle the 64-pixel slabs of the encoder, dmax beyond the width."""
    import stereo
    from synth import synth_pair_gray
    left, right, _ = synth_pair_gray(h, w, dmax, seed=h * 7 + w)
    dsi = oracle.wct_cost_volume(left, right, dmax)
    dref, cref, layers = _oracle_ref(oracle, dsi, left, 3, 0.0154)
    d, c = stereo.aggregate_and_wta(dsi, left, 3, 0.0154)
    assert_wta_matches(d, c, dref, cref, layers, min_agree=0.99)


def _volume(kind, rng, h, w, dmax):
    if kind == "sixteenths_small":       # unit 1/16, codes small enough to pack: the two-column kernel
        return (rng.integers(0, 900, size=(h, w, dmax)) / 16.0).astype(np.float32)
    if kind == "halves":                 # unit 1/2
        return (rng.integers(0, 2000, size=(h, w, dmax)) / 2.0).astype(np.float32)
    if kind == "all_zero":
        return np.zeros((h, w, dmax), np.float32)
    if kind == "random_floats":          # no common unit: the general kernel
        return (rng.random((h, w, dmax)) * 100.0).astype(np.float32)
    if kind == "negative":               # whole numbers, some negative
        return rng.integers(-5, 100, size=(h, w, dmax)).astype(np.float32)
    if kind == "too_large":              # whole numbers beyond what the packed sums hold
        return rng.integers(0, 60000, size=(h, w, dmax)).astype(np.float32)
    if kind == "beyond_u16":
        return (rng.integers(0, 100, size=(h, w, dmax)) * 1000.0).astype(np.float32)
    if kind == "one_odd_value":          # whole numbers but for ONE cost that the encoder's sample does not see
        v = rng.integers(0, 120, size=(h, w, dmax)).astype(np.float32)
        v[h // 2, w // 3, dmax // 2] = 17.03125
        return v
    if kind == "census_radius8":         # unit 1/16 with `large` = 1275: 20400 units, too large to pack
        return None
    raise AssertionError(kind)


@pytest.mark.parametrize("kind", ["sixteenths_small", "halves", "all_zero", "random_floats", "negative", "too_large",
                                  "beyond_u16", "one_odd_value", "census_radius8"])
def test_aggregate_arbitrary_volumes(oracle, kind):
    """Whatever the volume holds, the result is the oracle's: volumes of whole multiples of a power-of-two unit take
    the codes, the others the general kernel -- chosen on the device from what the encoder found."""
    import stereo
    from synth import synth_pair_gray
    rng = np.random.default_rng(len(kind))
    h, w, dmax = 90, 300, 12
    left, right, _ = synth_pair_gray(h, w, dmax, seed=3)
    dsi = _volume(kind, rng, h, w, dmax)
    if dsi is None:
        dsi = oracle.wct_cost_volume(left, right, dmax, radius=8)
    dref, cref, layers = _oracle_ref(oracle, dsi, left, 3, 0.0154)
    d, c = stereo.aggregate_and_wta(dsi, left, 3, 0.0154)
    if kind == "all_zero":
        assert np.all(c == 0.0) and np.all(d == 0)
        return
    # (random volumes have no structure: many layers come close, so the agreement bar is loose; every mismatch
    #  is gap-checked against the oracle's own layers)
    assert_wta_matches(d, c, dref, cref, layers, min_agree=0.98)


def _dev_setup(h, w, dmax, rg=3, eps=0.0154, seed=2014):
    import torch

    import _native
    from synth import synth_pair_gray
    lib = _native.lib()
    dev = torch.device("cuda", 0)
    left_np, right_np, _ = synth_pair_gray(h, w, dmax, seed)
    left, right = torch.from_numpy(left_np).to(dev), torch.from_numpy(right_np).to(dev)
    params = _native.StereoParams(wct_radius=4, base_weight=8.0, filter_kind=_native.FILTER_GUIDED, guided_radius=rg,
                                  guided_eps=eps, median_radius=3, gaussian_sigma=1.0, bilateral_sigma=1.0)
    return torch, _native, lib, dev, left_np, right_np, left, right, params


def test_device_two_call_equals_single_call_at_bench_size():
    """The benchmark's frame (2964 x 2000, 270 disparities) through the device entry points of the two-call API:
    the 6.4 GB volume is written, encoded and aggregated; the result is the single-call path's (see
    assert_same_as_single_call), for the whole range and for a disparity shard, with the large workspace (codes) --
    and with the small one (general kernel reading the volume in place) the costs agree to 1e-6."""
    h, w, dmax = 2000, 2964, 270
    torch, _native, lib, dev, _, _, left, right, params = _dev_setup(h, w, dmax)
    n = h * w
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    small, big = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _native.check(lib.stereo_fused_workspace_bytes(h, w, params, ctypes.byref(small)))
    _native.check(lib.stereo_aggregate_workspace_bytes(h, w, dmax, params, ctypes.byref(big)))
    assert big.value >= small.value + n * dmax * 2
    ws = torch.empty(big.value, dtype=torch.uint8, device=dev)
    dsi = torch.empty(n * dmax, dtype=torch.float32, device=dev)
    _native.check(lib.stereo_wct_cost_volume_dev(left.data_ptr(), right.data_ptr(), h, w, dmax, 4, 8.0, ws.data_ptr(),
                                                 big.value, dsi.data_ptr(), stream))
    single = torch.empty(n, dtype=torch.int64, device=dev)
    two = torch.empty(n, dtype=torch.int64, device=dev)
    disp = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]
    cost = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(2)]

    def unpacked():
        for keys, dd, cc in ((single, disp[0], cost[0]), (two, disp[1], cost[1])):
            _native.check(lib.stereo_unpack_keys_dev(keys.data_ptr(), n, dd.data_ptr(), cc.data_ptr(), stream))
        torch.cuda.synchronize()
        return [t.cpu().numpy().reshape(h, w) for t in (disp[1], cost[1], disp[0], cost[0])]

    for d0, d1 in ((0, dmax), (100, 171)):
        _native.check(lib.stereo_compute_disparity_dev(left.data_ptr(), right.data_ptr(), h, w, dmax, params, d0, d1,
                                                       ws.data_ptr(), big.value, single.data_ptr(), stream))
        two.fill_(0)
        _native.check(lib.stereo_aggregate_wta_dev(dsi.data_ptr(), left.data_ptr(), h, w, dmax, params, d0, d1,
                                                   ws.data_ptr(), big.value, two.data_ptr(), stream))
        assert_same_as_single_call(*unpacked(), dmax)
    # the small workspace: no room for codes, the general kernel reads the float32 volume in place
    _native.check(lib.stereo_aggregate_wta_dev(dsi.data_ptr(), left.data_ptr(), h, w, dmax, params, 100, 171,
                                               ws.data_ptr(), small.value, two.data_ptr(), stream))
    d2, c2, d1_, c1_ = unpacked()
    assert rel_err(c2, c1_) <= 1e-6
    assert np.mean(d2 == d1_) >= 0.9995


def test_encode_once_aggregate_twice_and_argument_checks():
    """stereo_encode_cost_volume_dev + stereo_aggregate_wta_codes_dev: one encoding serves several parameter sets;
    buffers that are too small and bad ranges are refused before any launch."""
    h, w, dmax = 120, 260, 20
    torch, _native, lib, dev, left_np, right_np, left, right, params = _dev_setup(h, w, dmax, seed=5)
    import stereo
    n = h * w
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws_bytes, cbytes = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _native.check(lib.stereo_fused_workspace_bytes(h, w, params, ctypes.byref(ws_bytes)))
    _native.check(lib.stereo_cost_codes_bytes(h, w, dmax, ctypes.byref(cbytes)))
    ws = torch.empty(ws_bytes.value, dtype=torch.uint8, device=dev)
    codes = torch.empty(cbytes.value, dtype=torch.uint8, device=dev)
    dsi = torch.empty(n * dmax, dtype=torch.float32, device=dev)
    _native.check(lib.stereo_wct_cost_volume_dev(left.data_ptr(), right.data_ptr(), h, w, dmax, 4, 8.0, ws.data_ptr(),
                                                 ws_bytes.value, dsi.data_ptr(), stream))
    _native.check(lib.stereo_encode_cost_volume_dev(dsi.data_ptr(), h, w, dmax, 0, dmax, codes.data_ptr(), cbytes.value, stream))
    # a volume that is not 16-byte aligned (the encoder then reads it 4 bytes at a time): the same codes
    shifted = torch.empty(n * dmax + 1, dtype=torch.float32, device=dev)
    shifted[1:] = dsi
    codes2 = torch.zeros_like(codes)
    _native.check(lib.stereo_encode_cost_volume_dev(shifted.data_ptr() + 4, h, w, dmax, 0, dmax, codes2.data_ptr(), cbytes.value, stream))
    torch.cuda.synchronize()
    assert torch.equal(codes[:32], codes2[:32]) and torch.equal(codes[256:], codes2[256:])
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    disp = torch.empty(n, dtype=torch.int32, device=dev)
    cost = torch.empty(n, dtype=torch.float32, device=dev)
    for rg, eps in ((3, 0.0154), (2, 0.01), (4, 0.001), (6, 0.0154)):    # (radius 6: the general kernel, from the volume)
        params.guided_radius, params.guided_eps = rg, eps
        _native.check(lib.stereo_aggregate_wta_codes_dev(codes.data_ptr(), dsi.data_ptr(), left.data_ptr(), h, w, dmax,
                                                         params, 0, dmax, ws.data_ptr(), ws_bytes.value, keys.data_ptr(), stream))
        _native.check(lib.stereo_unpack_keys_dev(keys.data_ptr(), n, disp.data_ptr(), cost.data_ptr(), stream))
        torch.cuda.synchronize()
        d1, c1 = stereo.compute_disparity(left_np, right_np, dmax, guided_radius=rg, guided_eps=eps, show_progress=False)
        if rg <= 4:
            assert_same_as_single_call(disp.cpu().numpy().reshape(h, w), cost.cpu().numpy().reshape(h, w), d1, c1, dmax)
        else:
            assert rel_err(cost.cpu().numpy().reshape(h, w), c1) <= 2e-6
    params.guided_radius, params.guided_eps = 3, 0.0154
    assert lib.stereo_encode_cost_volume_dev(dsi.data_ptr(), h, w, dmax, 0, dmax, codes.data_ptr(), cbytes.value - 1, stream) == _native.EINVAL
    assert lib.stereo_encode_cost_volume_dev(dsi.data_ptr(), h, w, dmax, 5, dmax + 1, codes.data_ptr(), cbytes.value, stream) == _native.EINVAL
    assert lib.stereo_aggregate_wta_codes_dev(None, dsi.data_ptr(), left.data_ptr(), h, w, dmax, params, 0, dmax, ws.data_ptr(),
                                              ws_bytes.value, keys.data_ptr(), stream) == _native.EINVAL
