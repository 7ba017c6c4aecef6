"""CPU rehearsal (gloo, world_size 2 and 3) of the multi-GPU path in stereo-matching_amd/sharding.py:
disparity-range shards -> packed (cost, d) keys -> all-to-all -> unsigned-min merge -> gather.
The per-shard results come from the CPU oracle; the merged frame must equal the unsharded one
bit for bit.  Also covers the scene dealer of the scene-parallel batch mode."""
from __future__ import annotations

import os
import socket
import sys
import tempfile

import numpy as np
import pytest

from conftest import PKG_DIR, REPO_ROOT


def pack_keys(cost: np.ndarray, disp: np.ndarray) -> np.ndarray:
    """(ordered uint32 image of float32 cost) << 32 | d -- the key layout of include/stereo_hip.h."""
    b = cost.astype(np.float32).view(np.uint32).astype(np.uint64)
    neg = (b >> np.uint64(31)) != 0
    ordered = np.where(neg, (~b) & np.uint64(0xFFFFFFFF), b | np.uint64(0x80000000))
    return (ordered << np.uint64(32)) | disp.astype(np.uint64)


def unpack_keys(keys: np.ndarray):
    ordered = (keys >> np.uint64(32)).astype(np.uint32)
    pos = (ordered & np.uint32(0x80000000)) != 0
    bits = np.where(pos, ordered & np.uint32(0x7FFFFFFF), ~ordered)
    return (keys & np.uint64(0xFFFFFFFF)).astype(np.int32), bits.astype(np.uint32).view(np.float32)


def _worker(rank: int, world: int, port: int, out_path: str) -> None:
    for p in (PKG_DIR, REPO_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    import sharding
    from oracle import oracle as orc
    from synth import synth_pair_gray

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        h, w, dmax = 37, 53, 23
        left, right, _ = synth_pair_gray(h, w, dmax, seed=5)
        dsi = orc.wct_cost_volume(left, right, dmax)
        d0, d1 = sharding.shard_range(dmax, world, rank)
        n = h * w
        chunk = sharding.padded_chunk(n, world)
        keys = np.full(chunk * world, np.uint64(0xFF800000) << np.uint64(32), dtype=np.uint64)
        if d1 > d0:
            disp, cost = orc.aggregate_and_wta(np.ascontiguousarray(dsi[:, :, d0:d1]), left, orc.FILTER_GUIDED, 3, 0.0154)
            disp = np.where(np.isfinite(cost), disp + d0, 0)
            keys[:n] = pack_keys(cost.ravel(), disp.ravel())
        kt = torch.from_numpy(keys.view(np.int64))

        def merge(dst, src):  # unsigned 64-bit min on CPU tensors (the GPU path uses stereo_merge_keys_dev)
            flip = torch.tensor(-(1 << 63), dtype=torch.int64)
            torch.minimum(dst ^ flip, src ^ flip, out=dst)
            dst ^= flip

        merged = sharding.exchange_and_merge(kt, world, merge)
        d_part, c_part = unpack_keys(merged.numpy().view(np.uint64))
        d_all = sharding.gather_slices(torch.from_numpy(d_part.copy()), world, rank)
        c_all = sharding.gather_slices(torch.from_numpy(c_part.copy()), world, rank)
        # the path bench.py takes: one gather of the merged key slices into a preallocated frame
        frame = torch.empty(merged.numel() * world, dtype=torch.int64) if rank == 0 else None
        got = sharding.gather_slices(merged.contiguous(), world, rank, out=frame)
        if rank == 0:
            dref, cref = orc.aggregate_and_wta(dsi, left, orc.FILTER_GUIDED, 3, 0.0154)
            ok = np.array_equal(d_all.numpy()[:n].reshape(h, w), dref) and np.array_equal(
                c_all.numpy()[:n].reshape(h, w), cref)
            d_f, c_f = unpack_keys(got.numpy().view(np.uint64))
            ok = ok and got is frame and np.array_equal(d_f[:n].reshape(h, w), dref) and np.array_equal(
                c_f[:n].reshape(h, w), cref)
        # the band-by-band overlapped exchange (bench.py's N > 1 path): uneven bands, the last one short
        frame_keys = keys[:n].copy()
        ox = sharding.OverlappedExchange(h, w, world, rank, [10, 20, 37], "cpu", use_dist=True)
        flip = torch.tensor(-(1 << 63), dtype=torch.int64)
        waited, out_d, out_c = [], np.zeros(n, np.int32), np.zeros(n, np.float32)

        def search(kbuf, epoch):
            kbuf.copy_(torch.from_numpy(frame_keys.view(np.int64)))

        def merge_slices(slices, chunk_, n_slices):
            v = (slices[:chunk_ * n_slices].view(n_slices, chunk_) ^ flip).min(dim=0).values ^ flip
            slices[:chunk_] = v

        def unpack(kbuf, count, first):
            dd, cc = unpack_keys(kbuf.numpy().view(np.uint64)[:count])
            out_d[first:first + count], out_c[first:first + count] = dd, cc

        for _ in range(2):  # twice: buffers are reused from step to step
            ox.run(search, lambda band, epoch: waited.append((band, epoch)), merge_slices, unpack)
        # the collective error check: nobody failed -> nobody raises; the LAST rank alone reports a timed-out band
        # wait -> every rank raises, with the per-rank counts in the message
        sharding.raise_if_any_rank_failed(0, "band waits", True)
        try:
            sharding.raise_if_any_rank_failed(1 if rank == world - 1 else 0, "band waits", True)
            raised = False
        except RuntimeError as exc:
            expected = [0] * (world - 1) + [1]
            raised = str(expected) in str(exc)
        with open(f"{out_path}.raised{rank}", "w") as f:
            f.write("raised" if raised else "silent")
        if rank == 0:
            ok = ok and np.array_equal(out_d.reshape(h, w), dref) and np.array_equal(out_c.reshape(h, w), cref)
            ok = ok and waited == [(0, 1), (1, 1), (2, 1), (0, 2), (1, 2), (2, 2)]
            with open(out_path, "w") as f:
                f.write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_disparity_shards_merge_over_gloo(world):
    import torch.multiprocessing as mp
    from oracle import oracle as orc

    orc.build()
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "result.txt")
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert open(out).read() == "ok"
        # an injected timed-out band wait on one rank raised on all of them
        assert [open(f"{out}.raised{r}").read() for r in range(world)] == ["raised"] * world


def test_key_roundtrip_and_order():
    rng = np.random.default_rng(0)
    cost = np.concatenate([rng.normal(0, 100, 500).astype(np.float32),
                           np.array([0.0, -0.0, np.inf, 1200.0, 1e-30, -1e-30], dtype=np.float32)])
    disp = rng.integers(0, 512, cost.size).astype(np.int32)
    keys = pack_keys(cost, disp)
    d2, c2 = unpack_keys(keys)
    assert np.array_equal(d2, disp) and np.array_equal(c2.view(np.uint32), cost.view(np.uint32))
    order = np.argsort(keys, kind="stable")
    lex = np.lexsort((disp, cost))
    # -0.0 sorts just below +0.0 as a key; numerically they compare equal, so compare (cost, d) pairs loosely
    assert np.all(np.diff(cost[order]) >= 0)
    assert np.array_equal(np.sort(cost[order]), np.sort(cost[lex]))
    init = np.uint64(0xFF800000) << np.uint64(32)
    assert np.all(pack_keys(np.array([np.inf], np.float32), np.array([3])) > init)


def test_shard_ranges_cover_and_scene_dealer():
    import sharding
    for dmax in (1, 7, 270, 512):
        for world in (1, 2, 3, 8):
            edges = [sharding.shard_range(dmax, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == dmax
            assert all(a[1] == b[0] for a, b in zip(edges[:-1], edges[1:]))
            assert max(e[1] - e[0] for e in edges) - min(e[1] - e[0] for e in edges) <= 1
    sizes = [50, 10, 40, 30, 20, 60, 5]
    plan = sharding.deal_scenes(sizes, 3)
    assert sorted(i for p in plan for i in p) == list(range(len(sizes)))
    loads = [sum(sizes[i] for i in p) for p in plan]
    assert max(loads) - min(loads) <= max(sizes)
    out = sharding.run_scenes_parallel(["b", "a"], [1, 2], 1, 0, lambda s: {"pbm": float(len(s))})
    assert list(out) == ["a", "b"]


# ---- scene-parallel batches (reference stereo.py:722-778 is the sequential loop they replace) ----------------------
SCENES = {   # scene -> (width, height, ndisp): heterogeneous work, names out of sorted order
    "Vintage-perfect": (40, 24, 76), "Adirondack-perfect": (36, 25, 29), "Piano-perfect": (35, 24, 26),
    "Jadeplant-perfect": (33, 25, 64), "Motorcycle-perfect": (37, 25, 27), "Backpack-perfect": (36, 25, 26),
    "Umbrella-perfect": (37, 25, 25),
}


def _make_fake_dataset(root: str) -> dict:
    """The dataset layout the batch mode walks: <root>/<scene>/im0.png (only its header is read, to size the work).
    One listed scene has no folder: it must be skipped, as the reference does (stereo.py:724-727)."""
    from PIL import Image
    for scene, (w, h, _) in SCENES.items():
        os.makedirs(os.path.join(root, scene))
        Image.fromarray(np.zeros((h, w), np.uint8)).save(os.path.join(root, scene, "im0.png"))
    ndisp = {scene: nd for scene, (_, _, nd) in SCENES.items()}
    ndisp["Missing-perfect"] = 50
    return ndisp


def _stub_run_scene(fail_scene=None, log=None):
    """Stands in for stereo._run_scene: metrics that depend on the scene alone."""
    def run(args, run_dir, left, right, gt, gt_mask, dmax, **kw):
        scene = kw["dataset_name"]
        if log is not None:
            log.append(scene)
        if scene == fail_scene:
            raise ValueError(f"cannot read {scene}")
        return {"pbm": float(dmax) / 7.0, "rms": float(len(scene)) * 0.37, "bad_threshold": float(args.bad_threshold),
                "valid_pixel_count": float(dmax * 11)}
    return run


def _batch_args():
    import stereo
    return stereo._parse_args(["--all-datasets", "--bad_threshold", "1.0", "--guided_eps", "0.0154"])


def _scene_worker(rank: int, world: int, port: int, root: str, run_dir: str, fail_scene) -> None:
    for p in (PKG_DIR, REPO_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import json
    from pathlib import Path

    import torch.distributed as dist

    import stereo

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ndisp = {scene: nd for scene, (_, _, nd) in SCENES.items()}
        ndisp["Missing-perfect"] = 50
        ran = []
        stereo._run_scene = _stub_run_scene(fail_scene, ran)
        stereo._resolve_scene_inputs = lambda root_, scene, table, override: (Path("l"), Path("r"), Path("g"), Path("m"),
                                                                              table[scene])
        status = "ok"
        try:
            summary = stereo._run_all_datasets(_batch_args(), Path(root), ndisp, Path(run_dir), world, rank)
            if (summary is None) != (rank != 0):
                status = "summary on the wrong rank"
        except RuntimeError as exc:
            status = f"raised: {exc}"
        with open(os.path.join(run_dir, f"rank{rank}.json"), "w") as f:
            json.dump({"status": status, "ran": ran}, f)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_scene_parallel_batch_over_gloo(world, monkeypatch):
    """stereo._run_all_datasets on `world` gloo ranks with a stub per-scene function: metrics_summary.json is byte
    for byte the single-rank file, scenes in sorted order, every scene run exactly once and the load spread over
    the ranks; a scene that raises makes every rank raise."""
    import json
    from pathlib import Path

    import torch.multiprocessing as mp

    import stereo

    with tempfile.TemporaryDirectory() as tmp:
        root = os.path.join(tmp, "dataset")
        ndisp = _make_fake_dataset(root)
        # world 1, in this process
        single = os.path.join(tmp, "single")
        os.makedirs(single)
        monkeypatch.setattr(stereo, "_run_scene", _stub_run_scene())
        monkeypatch.setattr(stereo, "_resolve_scene_inputs",
                            lambda root_, scene, table, override: (Path("l"), Path("r"), Path("g"), Path("m"), table[scene]))
        summary = stereo._run_all_datasets(_batch_args(), Path(root), ndisp, Path(single), 1, 0)
        assert list(summary["scenes"]) == sorted(SCENES) and summary["scene_count"] == len(SCENES)
        assert summary["bad_threshold"] == 1.0
        want = open(os.path.join(single, "metrics_summary.json"), "rb").read()
        # the same batch on `world` ranks
        multi = os.path.join(tmp, "multi")
        os.makedirs(multi)
        mp.spawn(_scene_worker, args=(world, _free_port(), root, multi, None), nprocs=world, join=True)
        assert open(os.path.join(multi, "metrics_summary.json"), "rb").read() == want
        reports = [json.load(open(os.path.join(multi, f"rank{r}.json"))) for r in range(world)]
        assert all(r["status"] == "ok" for r in reports), reports
        ran = [s for r in reports for s in r["ran"]]
        assert sorted(ran) == sorted(SCENES)
        work = {s: w * h * nd for s, (w, h, nd) in SCENES.items()}
        loads = [sum(work[s] for s in r["ran"]) for r in reports]
        assert max(loads) - min(loads) <= max(work.values()) and min(loads) > 0
        # every scene got its own output folder, on whichever rank ran it
        assert sorted(d for d in os.listdir(multi) if os.path.isdir(os.path.join(multi, d))) == sorted(SCENES)
        # one scene raises on the rank that runs it: every rank raises, nobody hangs in the gather
        failing = os.path.join(tmp, "failing")
        os.makedirs(failing)
        mp.spawn(_scene_worker, args=(world, _free_port(), root, failing, "Piano-perfect"), nprocs=world, join=True)
        reports = [json.load(open(os.path.join(failing, f"rank{r}.json"))) for r in range(world)]
        assert all(r["status"].startswith("raised:") and "Piano-perfect" in r["status"] for r in reports), reports
        assert not os.path.exists(os.path.join(failing, "metrics_summary.json"))
    # with one rank the scene's own exception propagates
    with tempfile.TemporaryDirectory() as tmp:
        root = os.path.join(tmp, "dataset")
        ndisp = _make_fake_dataset(root)
        monkeypatch.setattr(stereo, "_run_scene", _stub_run_scene("Piano-perfect"))
        os.makedirs(os.path.join(tmp, "run"))
        with pytest.raises(ValueError):
            stereo._run_all_datasets(_batch_args(), Path(root), ndisp, Path(tmp) / "run", 1, 0)
