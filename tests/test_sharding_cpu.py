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
