"""Row-band progress reporting of the fused search (stereo_progress_plan /
stereo_compute_disparity_progress_dev / stereo_wait_progress_dev) and the one-pass slice merge, through the C ABI:
the keys must equal those of the plain entry point, every band flag must carry the epoch, and a waiter enqueued
on a second stream must let band-by-band consumers see final keys."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(native, filt="guided", wct_radius=4):
    kind = {"guided": native.FILTER_GUIDED, "median": native.FILTER_MEDIAN_U8Q, "gaussian": native.FILTER_GAUSSIAN}[filt]
    return native.StereoParams(wct_radius=wct_radius, base_weight=8.0, filter_kind=kind, guided_radius=3, guided_eps=0.0154,
                               median_radius=3, gaussian_sigma=1.0, bilateral_sigma=1.0)


@pytest.mark.parametrize("filt,h,w,dmax,bands", [("guided", 400, 500, 48, 4), ("guided", 95, 300, 20, 16),
                                                 ("guided", 25, 130, 9, 3), ("guided", 1500, 2900, 96, 5), ("median", 120, 200, 16, 4),
                                                 ("gaussian", 120, 200, 16, 2)])
def test_progress_reporting_matches_plain_search(filt, h, w, dmax, bands):
    import torch

    import _native
    from synth import synth_pair_gray

    lib = _native.lib()
    dev = torch.device("cuda", 0)
    left_np, right_np, _ = synth_pair_gray(h, w, dmax, seed=h + w)
    left, right = torch.from_numpy(left_np).to(dev), torch.from_numpy(right_np).to(dev)
    params = _params(_native, filt)
    ws_bytes = ctypes.c_size_t(0)
    _native.check(lib.stereo_fused_workspace_bytes(h, w, params, ctypes.byref(ws_bytes)))
    ws = torch.empty(ws_bytes.value, dtype=torch.uint8, device=dev)
    main = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    plain = torch.empty(h * w, dtype=torch.int64, device=dev)
    _native.check(lib.stereo_compute_disparity_dev(left.data_ptr(), right.data_ptr(), h, w, dmax, params, 0, dmax,
                                                   ws.data_ptr(), ws_bytes.value, plain.data_ptr(), main))
    row_end = (ctypes.c_int32 * 16)()
    n_bands = ctypes.c_int(0)
    _native.check(lib.stereo_progress_plan(h, w, params, bands, row_end, ctypes.byref(n_bands)))
    ends = list(row_end[:n_bands.value])
    assert 1 <= n_bands.value <= bands and ends[-1] == h and ends == sorted(set(ends)) and ends[0] > 0

    progress = torch.zeros(48, dtype=torch.int32, device=dev)
    keys = torch.empty(h * w, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    snapshots = []
    for epoch in (1, 2):  # twice: counters are cleared per call, flags move to the new epoch
        keys.fill_(0)
        torch.cuda.synchronize()
        _native.check(lib.stereo_compute_disparity_progress_dev(
            left.data_ptr(), right.data_ptr(), h, w, dmax, params, 0, dmax, ws.data_ptr(), ws_bytes.value, keys.data_ptr(),
            n_bands.value, progress.data_ptr(), epoch, main))
        with torch.cuda.stream(side):
            sp = ctypes.c_void_p(side.cuda_stream)
            y0 = 0
            for b, y1 in enumerate(ends):
                _native.check(lib.stereo_wait_progress_dev(progress.data_ptr(), b, epoch, sp))
                snapshots.append((epoch, y0, y1, keys[y0 * w:y1 * w].clone()))  # copy taken right after the wait
                y0 = y1
        torch.cuda.synchronize()
        words = progress.cpu().numpy()
        assert list(words[16:16 + n_bands.value]) == [epoch] * n_bands.value
        assert not words[32:48].any(), "a waiter timed out"
        assert torch.equal(keys, plain)
    for epoch, y0, y1, snap in snapshots:
        assert torch.equal(snap, plain[y0 * w:y1 * w]), f"band rows [{y0}, {y1}) were not final when their flag was seen"


def test_progress_rejects_foreign_band_counts():
    import torch

    import _native

    lib = _native.lib()
    params = _params(_native)
    row_end = (ctypes.c_int32 * 16)()
    n_bands = ctypes.c_int(0)
    assert lib.stereo_progress_plan(100, 100, params, 17, row_end, ctypes.byref(n_bands)) != 0
    assert lib.stereo_progress_plan(100, 100, params, 0, row_end, ctypes.byref(n_bands)) != 0
    _native.check(lib.stereo_progress_plan(100, 100, params, 16, row_end, ctypes.byref(n_bands)))
    assert n_bands.value < 16  # 100 rows are four tile rows of the two-column kernel
    dev = torch.device("cuda", 0)
    img = torch.zeros(100 * 100, dtype=torch.float32, device=dev)
    ws_bytes = ctypes.c_size_t(0)
    _native.check(lib.stereo_fused_workspace_bytes(100, 100, params, ctypes.byref(ws_bytes)))
    ws = torch.empty(ws_bytes.value, dtype=torch.uint8, device=dev)
    keys = torch.empty(100 * 100, dtype=torch.int64, device=dev)
    progress = torch.zeros(48, dtype=torch.int32, device=dev)
    rc = lib.stereo_compute_disparity_progress_dev(img.data_ptr(), img.data_ptr(), 100, 100, 8, params, 0, 8, ws.data_ptr(),
                                                   ws_bytes.value, keys.data_ptr(), 3, progress.data_ptr(), 1, None)
    assert rc != 0
    torch.cuda.synchronize()


def test_merge_key_slices_is_the_unsigned_min():
    import torch

    import _native

    lib = _native.lib()
    rng = np.random.default_rng(3)
    chunk, n_slices = 10007, 5
    raw = rng.integers(0, 2**64, size=chunk * n_slices, dtype=np.uint64)
    want = raw.reshape(n_slices, chunk).min(axis=0)
    t = torch.from_numpy(raw.view(np.int64)).cuda()
    _native.check(lib.stereo_merge_key_slices_dev(t.data_ptr(), chunk, n_slices, None))
    torch.cuda.synchronize()
    got = t.cpu().numpy().view(np.uint64)
    assert np.array_equal(got[:chunk], want) and np.array_equal(got[chunk:], raw[chunk:])


def test_device_entry_points_reject_bad_arguments():
    """stereo_compute_disparity_dev / stereo_fused_workspace_bytes: status codes, nothing launched on bad input."""
    import torch

    import _native

    lib = _native.lib()
    dev = torch.device("cuda", 0)
    h, w, dmax = 40, 60, 8
    img = torch.rand(h * w, dtype=torch.float32, device=dev)
    params = _params(_native)
    nbytes = ctypes.c_size_t(0)
    _native.check(lib.stereo_fused_workspace_bytes(h, w, params, ctypes.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    keys = torch.empty(h * w, dtype=torch.int64, device=dev)

    def call(hh=h, ww=w, dd=dmax, p=params, d0=0, d1=dmax, ws_bytes=nbytes.value, left=img.data_ptr()):
        return lib.stereo_compute_disparity_dev(left, img.data_ptr(), hh, ww, dd, p, d0, d1, ws.data_ptr(), ws_bytes,
                                                keys.data_ptr(), None)

    assert call() == _native.OK
    assert call(ws_bytes=nbytes.value - 1) == _native.EINVAL and "workspace" in _native.last_error()
    assert call(d0=5, d1=4) == _native.EINVAL
    assert call(d1=dmax + 1) == _native.EINVAL
    assert call(dd=0, d1=0) == _native.EINVAL
    assert call(hh=0) == _native.EINVAL
    assert call(left=None) == _native.EINVAL
    big_r = _native.StereoParams(wct_radius=4, base_weight=8.0, filter_kind=_native.FILTER_GUIDED, guided_radius=9,
                                 guided_eps=0.0154, median_radius=3, gaussian_sigma=1.0, bilateral_sigma=1.0)
    assert call(p=big_r) == _native.EUNSUPPORTED
    bad_r = _native.StereoParams(wct_radius=9, base_weight=8.0, filter_kind=_native.FILTER_GUIDED, guided_radius=3,
                                 guided_eps=0.0154, median_radius=3, gaussian_sigma=1.0, bilateral_sigma=1.0)
    assert call(p=bad_r) in (_native.EINVAL, _native.EUNSUPPORTED)
    assert lib.stereo_fused_workspace_bytes(h, w, params, None) == _native.EINVAL
    # an empty disparity range is legal: every key stays (+inf, 0)
    assert call(d0=3, d1=3) == _native.OK
    disp = torch.empty(h * w, dtype=torch.int32, device=dev)
    cost = torch.empty(h * w, dtype=torch.float32, device=dev)
    _native.check(lib.stereo_unpack_keys_dev(keys.data_ptr(), h * w, disp.data_ptr(), cost.data_ptr(), None))
    torch.cuda.synchronize()
    assert int(disp.abs().sum()) == 0 and bool(torch.isinf(cost).all())


def test_progress_reporting_random_shapes():
    """Random frame shapes, band counts, disparity ranges and filters: the reporting entry point writes the same keys
    as the plain one and publishes every band flag."""
    import torch

    import _native
    from synth import synth_pair_gray

    lib = _native.lib()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(99)
    main = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for case in range(40):
        h, w, dmax = int(rng.integers(1, 260)), int(rng.integers(1, 400)), int(rng.integers(1, 70))
        filt = str(rng.choice(["guided", "guided", "median", "gaussian"]))
        bands = int(rng.integers(1, 17))
        d0 = int(rng.integers(0, dmax))
        d1 = int(rng.integers(d0, dmax + 1))
        left_np, right_np, _ = synth_pair_gray(h, w, dmax, seed=case)
        left, right = torch.from_numpy(left_np).to(dev), torch.from_numpy(right_np).to(dev)
        params = _params(_native, filt, wct_radius=int(rng.choice([4, 8])))
        nbytes = ctypes.c_size_t(0)
        _native.check(lib.stereo_fused_workspace_bytes(h, w, params, ctypes.byref(nbytes)))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        plain = torch.empty(h * w, dtype=torch.int64, device=dev)
        keys = torch.zeros(h * w, dtype=torch.int64, device=dev)
        _native.check(lib.stereo_compute_disparity_dev(left.data_ptr(), right.data_ptr(), h, w, dmax, params, d0, d1,
                                                       ws.data_ptr(), nbytes.value, plain.data_ptr(), main))
        row_end = (ctypes.c_int32 * 16)()
        n_bands = ctypes.c_int(0)
        _native.check(lib.stereo_progress_plan(h, w, params, bands, row_end, ctypes.byref(n_bands)))
        progress = torch.zeros(48, dtype=torch.int32, device=dev)
        _native.check(lib.stereo_compute_disparity_progress_dev(
            left.data_ptr(), right.data_ptr(), h, w, dmax, params, d0, d1, ws.data_ptr(), nbytes.value, keys.data_ptr(),
            n_bands.value, progress.data_ptr(), 7, main))
        for b in range(n_bands.value):
            _native.check(lib.stereo_wait_progress_dev(progress.data_ptr(), b, 7, main))
        torch.cuda.synchronize()
        words = progress.cpu().numpy()
        tag = f"case {case}: {w}x{h} dmax {dmax} d[{d0},{d1}) {filt} bands {bands}->{n_bands.value}"
        assert list(words[16:16 + n_bands.value]) == [7] * n_bands.value and not words[32:48].any(), tag
        assert torch.equal(keys, plain), tag


@pytest.mark.parametrize("filt,world", [("guided", 3), ("guided", 8), ("median", 3)])
def test_sharded_frame_search_ranks_merge_to_the_plain_search(filt, world):
    """ShardedFrameSearch as every rank of a `world`-rank job in turn (no process group: each rank's keys stay in
    its exchange buffer): the key-wise minimum over the ranks' contiguous disparity ranges is the whole-range search."""
    import torch

    import _native
    import sharding
    from synth import synth_pair_gray

    lib = _native.lib()
    dev = torch.device("cuda", 0)
    h, w, dmax = 140, 330, 37
    left_np, right_np, _ = synth_pair_gray(h, w, dmax, seed=77)
    left, right = torch.from_numpy(left_np).to(dev).view(-1), torch.from_numpy(right_np).to(dev).view(-1)
    params = _params(_native, filt)
    main = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nbytes = ctypes.c_size_t(0)
    _native.check(lib.stereo_fused_workspace_bytes(h, w, params, ctypes.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    plain = torch.empty(h * w, dtype=torch.int64, device=dev)
    _native.check(lib.stereo_compute_disparity_dev(left.data_ptr(), right.data_ptr(), h, w, dmax, params, 0, dmax,
                                                   ws.data_ptr(), nbytes.value, plain.data_ptr(), main))
    slices = torch.empty(world * h * w, dtype=torch.int64, device=dev)
    searched = 0
    for rank in range(world):
        search = sharding.ShardedFrameSearch(h, w, dmax, params, world, rank, dev, n_bands=3, use_dist=False)
        searched += search.disparities_searched()
        disp, cost = search.run(left, right)
        assert (disp is None) == (rank != 0)
        torch.cuda.synchronize()
        assert search.band_waits_timed_out() == 0
        slices[rank * h * w:(rank + 1) * h * w] = search.exchange.keys[:h * w]
    assert searched == dmax
    _native.check(lib.stereo_merge_key_slices_dev(slices.data_ptr(), h * w, world, main))
    torch.cuda.synchronize()
    assert torch.equal(slices[:h * w], plain)


@pytest.mark.parametrize("h,w,dmax,bands", [(1210, 1500, 64, 1), (1210, 1500, 21, 3), (2000, 2964, 40, 4), (700, 3000, 300, 2)])
def test_tail_split_writes_the_same_keys(h, w, dmax, bands, monkeypatch):
    """Frames with more tiles than compute units: the launch cuts its last tiles into disparity slices that merge
    through atomicMin.  Keys and band flags must equal those of the same launch with the split switched off."""
    import torch

    import _native
    from synth import synth_pair_gray

    lib = _native.lib()
    dev = torch.device("cuda", 0)
    left_np, right_np, _ = synth_pair_gray(h, w, dmax, seed=h)
    left, right = torch.from_numpy(left_np).to(dev), torch.from_numpy(right_np).to(dev)
    params = _params(_native)
    nbytes = ctypes.c_size_t(0)
    _native.check(lib.stereo_fused_workspace_bytes(h, w, params, ctypes.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    main = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    row_end = (ctypes.c_int32 * 16)()
    n_bands = ctypes.c_int(0)
    _native.check(lib.stereo_progress_plan(h, w, params, bands, row_end, ctypes.byref(n_bands)))
    results = []
    for no_split in ("1", "0"):
        monkeypatch.setenv("STEREO_NO_TAIL_SPLIT", no_split)
        keys = torch.zeros(h * w, dtype=torch.int64, device=dev)
        progress = torch.zeros(48, dtype=torch.int32, device=dev)
        _native.check(lib.stereo_compute_disparity_progress_dev(
            left.data_ptr(), right.data_ptr(), h, w, dmax, params, 0, dmax, ws.data_ptr(), nbytes.value, keys.data_ptr(),
            n_bands.value, progress.data_ptr(), 3, main))
        for b in range(n_bands.value):
            _native.check(lib.stereo_wait_progress_dev(progress.data_ptr(), b, 3, main))
        plain = torch.zeros(h * w, dtype=torch.int64, device=dev)
        _native.check(lib.stereo_compute_disparity_dev(left.data_ptr(), right.data_ptr(), h, w, dmax, params, 0, dmax,
                                                       ws.data_ptr(), nbytes.value, plain.data_ptr(), main))
        torch.cuda.synchronize()
        words = progress.cpu().numpy()
        assert list(words[16:16 + n_bands.value]) == [3] * n_bands.value and not words[32:48].any()
        assert torch.equal(keys, plain)
        results.append(keys)
    assert torch.equal(results[0], results[1])
