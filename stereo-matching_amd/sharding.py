"""Multi-GPU execution of the stereo hot path on one node (one process per GPU, RCCL over xGMI).

Two partitionings, both taken from how the reference's work decomposes:

* Disparity-range sharding of ONE frame.  Every disparity layer is filtered independently
  (reference stereo.py:145-166) and WTA is a lexicographic min over (cost, d)
  (stereo.py:167-170), so rank r searches d in [r*D/N, (r+1)*D/N) on replicated images and the
  per-pixel results are merged.  Results travel as packed 64-bit keys
  (ordered float32 cost << 32 | d): an UNSIGNED min of keys is exactly the reference's rule
  "lowest cost, ties to the lowest disparity", so the merged frame is bit-identical to the
  unsharded one for any partition.
  The exchange is an all-to-all: rank r sends pixel slice j of its keys straight to rank j (one
  hop on the point-to-point xGMI mesh, all links busy at once), min-merges the N slices it
  receives and hands its merged slice to rank 0, which unpacks the frame.  Ring all-reduce would move the same bytes
  over one link at a time.

* Scene-parallel batches (reference stereo.py:722-778 loops scenes sequentially): scenes are
  dealt to ranks longest-first by H*W*ndisp; no collective on the data path, only a gather of
  the per-scene metric dicts.

The functions take the collective backend from the initialised torch.distributed process group
(`nccl` on GPUs; the CPU rehearsal tests run the same code over `gloo`).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple


def shard_range(dmax: int, world: int, rank: int) -> Tuple[int, int]:
    """Disparities [d_begin, d_end) of `rank`: contiguous, near-equal, covering [0, dmax)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    return rank * dmax // world, (rank + 1) * dmax // world


def padded_chunk(n_pixels: int, world: int) -> int:
    """Pixels per rank after padding the frame to a multiple of `world`."""
    return (n_pixels + world - 1) // world


def exchange_and_merge(keys, world: int, merge_fn: Callable, recv=None):
    """All-to-all the key slices and min-merge them.

    keys : int64 tensor [world * chunk] holding the uint64 key bits of THIS rank's shard
    merge_fn(dst, src) : in-place unsigned 64-bit min of two [chunk] tensors
    Returns the [chunk] tensor of merged keys for this rank's pixel slice.
    """
    import torch
    import torch.distributed as dist

    if keys.numel() % world:
        raise ValueError("keys must be padded to a multiple of the world size")
    chunk = keys.numel() // world
    if world == 1:
        return keys[:chunk]
    if recv is None:
        recv = torch.empty_like(keys)
    dist.all_to_all_single(recv, keys)
    merged = recv[:chunk]
    for j in range(1, world):
        merge_fn(merged, recv[j * chunk:(j + 1) * chunk])
    return merged


def gather_slices(part, world: int, rank: int, dst: int = 0, out=None):
    """Gather equal-length slices on `dst`; returns the concatenated tensor there, None elsewhere.
    `out` (on `dst`, world * len(part) elements) receives the slices in place when given."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return part
    if rank != dst:
        dist.gather(part, None, dst=dst)
        return None
    n = part.numel()
    if out is None:
        out = torch.empty(world * n, dtype=part.dtype, device=part.device)
    dist.gather(part, [out[j * n:(j + 1) * n] for j in range(world)], dst=dst)
    return out


def raise_if_any_rank_failed(local_failures: int, what: str, use_dist: bool, device="cpu") -> None:
    """Collective error check: every rank passes the number of failures IT saw (timed-out band waits, pixels that
    differ from a reference, ...).  One small all_reduce makes the counts known everywhere, and RuntimeError is
    raised on EVERY rank when any rank counted one -- a rank that raised alone would leave its peers blocked in the
    next collective.  Without a process group (`use_dist` false) only the local count is looked at."""
    import torch
    import torch.distributed as dist

    counts = [int(local_failures)]
    if use_dist and dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(), dist.get_rank()
        t = torch.zeros(world, dtype=torch.int64, device=device)
        t[rank] = int(local_failures)
        dist.all_reduce(t)
        counts = [int(v) for v in t.tolist()]
    if any(counts):
        raise RuntimeError(f"{what}: failures per rank {counts}")


class _Band:
    __slots__ = ("y0", "y1", "n", "chunk", "send", "recv", "frame")


class OverlappedExchange:
    """Disparity-sharded search of one frame with the exchange of finished rows overlapped with
    the search of the remaining ones.

    The fused kernel works through the frame from the top and reports, per row band, when the
    band's keys are final (stereo_compute_disparity_progress_dev).  A side stream waits for band
    b, all-to-alls its key slices, min-merges what it receives and hands the merged slice to rank
    0 (gather), which unpacks it -- while the kernel is still searching the bands below.  Only the
    last band's exchange is exposed.  The collectives are issued in the same order on every rank
    (all-to-all, then gather, band by band).  All buffers are allocated once, here.

    row_end[b] is the end row (exclusive) of band b (stereo_progress_plan).

    search_fn(keys, epoch)        enqueue the whole-frame search of this rank's disparities on the
                                  current stream; `keys`: height * width int64 key bits
    wait_band_fn(band, epoch)     make the current stream wait until `band` of `epoch` is final
    merge_fn(slices, chunk, n)    unsigned 64-bit min over the `n` slices of `chunk` keys laid end to end in
                                  `slices`, into slice 0, on the current stream
    unpack_fn(keys, n, first)     rank 0: write disparity / cost of `n` pixels from pixel `first`
    """

    def __init__(self, height: int, width: int, world: int, rank: int, row_end: Sequence[int], device,
                 use_dist: Optional[bool] = None):
        import torch

        if not row_end or list(row_end) != sorted(set(row_end)) or row_end[-1] != height or row_end[0] <= 0:
            raise ValueError("row_end must increase strictly and end at the frame height")
        self.world, self.rank, self.width, self.height = world, rank, width, height
        self.use_dist = (world > 1) if use_dist is None else use_dist
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.side = self._side_stream() if self.on_gpu else None
        self.epoch = 0
        n = height * width
        last_first = (row_end[-2] if len(row_end) > 1 else 0) * width
        last_chunk = padded_chunk(n - last_first, world)
        # the padded exchange window of the last band may run past the frame: give the buffer room
        self.keys = torch.empty(max(n, last_first + last_chunk * world), dtype=torch.int64, device=self.device)
        self.keys.fill_(-(1 << 63))  # pad beyond the frame: travels, is never unpacked
        self.bands: List[_Band] = []
        y0 = 0
        for y1 in row_end:
            b = _Band()
            b.y0, b.y1 = y0, y1
            b.n = (y1 - y0) * width
            b.chunk = padded_chunk(b.n, world)
            # a band's padded window may reach into the first pixels of the next band: those
            # travel as whatever they hold at the time and are never unpacked
            b.send = self.keys[y0 * width:y0 * width + b.chunk * world]
            b.recv = torch.empty(b.chunk * world, dtype=torch.int64, device=self.device) if self.use_dist else None
            b.frame = (torch.empty(b.chunk * world, dtype=torch.int64, device=self.device)
                       if self.use_dist and rank == 0 else None)
            self.bands.append(b)
            y0 = y1

    def _side_stream(self):
        """A stream whose work runs BESIDE the current stream's.  The runtime maps streams onto a few hardware queues
        (four by default) and serves two streams that share one in submission order: waits and collectives on such a
        side stream would queue up behind the search instead of accompanying it (seen on this stack: the first side
        stream PyTorch hands out and the default stream share a queue).  The mapping cannot be asked for, so
        candidates are tried with stereo_streams_overlap and the first that overlaps is kept; if none does, the last
        one serves -- correct, merely without the overlap.  (Not a high-priority stream: a polling wait on a
        high-priority queue was seen to hold back the search it waits for, for as long as the wait's time limit.)"""
        import ctypes

        import torch

        import _native

        lib = _native.lib()
        main = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self.side_overlaps = False
        self._candidates = []   # kept alive: a stream that is released may be handed out again as the next candidate
        side = None
        with torch.cuda.device(self.device):
            for _ in range(8):
                side = torch.cuda.Stream(device=self.device)
                self._candidates.append(side)
                overlap = ctypes.c_int(0)
                _native.check(lib.stereo_streams_overlap(main, ctypes.c_void_p(side.cuda_stream), ctypes.byref(overlap)))
                if overlap.value:
                    self.side_overlaps = True
                    break
        return side

    def run(self, search_fn: Callable, wait_band_fn: Callable, merge_fn: Callable, unpack_fn: Callable) -> None:
        import contextlib

        import torch
        import torch.distributed as dist

        self.epoch += 1
        epoch = self.epoch
        world, rank = self.world, self.rank
        search_fn(self.keys[:self.height * self.width], epoch)
        side_ctx = torch.cuda.stream(self.side) if self.on_gpu else contextlib.nullcontext()
        with side_ctx:
            for i, b in enumerate(self.bands):
                wait_band_fn(i, epoch)
                if not self.use_dist:
                    if rank == 0:   # (a rank without a process group keeps its keys in self.keys)
                        unpack_fn(b.send, b.n, b.y0 * self.width)
                    continue
                # one band at a time, start to finish: this stream must not block on the flag of band
                # i+1 (which may be a long way off) with work of band i still undone
                dist.all_to_all_single(b.recv, b.send)
                merge_fn(b.recv, b.chunk, world)
                merged = b.recv[:b.chunk]
                parts = [b.frame[j * b.chunk:(j + 1) * b.chunk] for j in range(world)] if rank == 0 else None
                dist.gather(merged, parts, dst=0)
                if rank == 0:
                    unpack_fn(b.frame, b.n, b.y0 * self.width)
        if self.on_gpu:
            # the next search overwrites the key buffer, and the caller reads the outputs on its own stream
            torch.cuda.current_stream(self.device).wait_stream(self.side)


class ShardedFrameSearch:
    """compute_disparity of one frame with the disparity range split over the ranks of the default
    process group (reference stereo.py:177-245 is the single-process original).  Device-resident:
    float32 gray images in, int32 disparity and float32 min cost out on rank 0; every buffer is
    allocated here, once.  Built on the library's device entry points and OverlappedExchange.

    Only parameter sets with a fused kernel are accepted (stereo_compute_disparity_dev raises
    NotImplementedError for the others at the first run).
    """

    def __init__(self, height: int, width: int, dmax: int, params, world: int, rank: int, device, n_bands: int = 0,
                 use_dist: Optional[bool] = None):
        import ctypes

        import torch

        import _native

        self._native, self._ctypes, self._torch = _native, ctypes, torch
        self.lib = _native.lib()
        self.H, self.W, self.D, self.params = height, width, dmax, params
        self.world, self.rank, self.device = world, rank, torch.device(device)
        self.use_dist = (world > 1) if use_dist is None else use_dist
        self.d_begin, self.d_end = shard_range(dmax, world, rank)
        nbytes = ctypes.c_size_t(0)
        _native.check(self.lib.stereo_fused_workspace_bytes(height, width, params, ctypes.byref(nbytes)))
        self.ws_bytes = nbytes.value
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)
        row_end = (ctypes.c_int32 * 16)()
        bands = ctypes.c_int(0)
        want = n_bands or (4 if self.use_dist else 1)
        _native.check(self.lib.stereo_progress_plan(height, width, params, max(1, min(want, 16)), row_end, ctypes.byref(bands)))
        self.n_bands = bands.value
        self.progress = torch.zeros(48, dtype=torch.int32, device=self.device)
        self.exchange = OverlappedExchange(height, width, world, rank, list(row_end[:self.n_bands]), self.device,
                                           use_dist=self.use_dist)
        if self.exchange.side is not None:
            # the first band wait polls the progress words on the side stream: it must not run ahead of the
            # kernel that clears them above
            self.exchange.side.wait_stream(torch.cuda.current_stream(self.device))
        n = height * width
        self.disparity = torch.zeros(n, dtype=torch.int32, device=self.device) if rank == 0 else None
        self.min_cost = torch.full((n,), float("inf"), dtype=torch.float32, device=self.device) if rank == 0 else None
        self._left = self._right = None

    def _stream(self):
        return self._ctypes.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _search(self, keys, epoch) -> None:
        self._native.check(self.lib.stereo_compute_disparity_progress_dev(
            self._left.data_ptr(), self._right.data_ptr(), self.H, self.W, self.D, self.params, self.d_begin, self.d_end,
            self.workspace.data_ptr(), self.ws_bytes, keys.data_ptr(), self.n_bands, self.progress.data_ptr(), epoch,
            self._stream()))

    def disparities_searched(self) -> int:
        """Number of disparities this rank searches per frame."""
        return self.d_end - self.d_begin

    def _wait(self, band, epoch) -> None:
        self._native.check(self.lib.stereo_wait_progress_dev(self.progress.data_ptr(), band, epoch, self._stream()))

    def _merge(self, slices, chunk, n_slices) -> None:
        self._native.check(self.lib.stereo_merge_key_slices_dev(slices.data_ptr(), chunk, n_slices, self._stream()))

    def _unpack(self, keys, count, first) -> None:
        self._native.check(self.lib.stereo_unpack_keys_dev(keys.data_ptr(), count, self.disparity.data_ptr() + first * 4,
                                                           self.min_cost.data_ptr() + first * 4, self._stream()))

    def run(self, left, right):
        """left, right: contiguous float32 device tensors of H * W elements.  Enqueues one frame on the
        current stream (+ the exchange's side stream) and returns (disparity, min_cost) flat device
        tensors on rank 0, (None, None) elsewhere.  No host synchronisation."""
        torch = self._torch
        for img in (left, right):
            if img.dtype != torch.float32 or img.numel() != self.H * self.W or not img.is_contiguous() or \
                    img.device != self.device:
                raise ValueError("images must be contiguous float32 tensors of H * W elements on the search's device")
        self._left, self._right = left, right
        self.exchange.run(self._search, self._wait, self._merge, self._unpack)
        return self.disparity, self.min_cost

    def band_waits_timed_out(self) -> int:
        """Band waits that gave up (a producer that never ran); 0 in a healthy run.  Synchronises."""
        return int((self.progress[32:48] != 0).sum().item())

    def check(self) -> None:
        """Collective: raises RuntimeError on every rank if a band wait timed out on any rank since this object
        was made.  A timed-out wait lets keys travel that were not final, so every frame produced since then is
        suspect; call this after synchronising and before using the results.  Synchronises."""
        raise_if_any_rank_failed(self.band_waits_timed_out(),
                                 "row-band waits of the sharded search timed out (the rows shipped were not final)",
                                 self.use_dist, self.device)


def deal_scenes(sizes: Sequence[int], world: int) -> List[List[int]]:
    """Longest-first greedy assignment of scene indices to ranks (balanced total work)."""
    if world <= 0:
        raise ValueError("world must be positive")
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    load = [0] * world
    plan: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        plan[r].append(i)
        load[r] += sizes[i]
    return plan


def run_scenes_parallel(scene_names: Sequence[str], sizes: Sequence[int], world: int, rank: int,
                        run_scene: Callable[[str], Dict[str, float]]) -> Optional[Dict[str, Dict[str, float]]]:
    """Run this rank's share of the scenes and gather {scene: metrics} on rank 0 (sorted by scene name, whatever
    the dealing was); None on the other ranks.

    A scene that raises stops its rank's share; the error texts are exchanged before the gather, so that EVERY rank
    raises RuntimeError naming the failed scenes -- a rank that raised alone would leave its peers blocked in the
    gather.  With one rank the scene's own exception propagates, as in the reference's sequential loop
    (reference stereo.py:722-778)."""
    plan = deal_scenes(sizes, world)
    mine: Dict[str, Dict[str, float]] = {}
    if world == 1:
        for i in plan[rank]:
            mine[scene_names[i]] = run_scene(scene_names[i])
        return {k: mine[k] for k in sorted(mine)}
    import torch.distributed as dist

    failure: Optional[str] = None
    for i in plan[rank]:
        try:
            mine[scene_names[i]] = run_scene(scene_names[i])
        except Exception as exc:   # noqa: BLE001 -- reported on every rank below
            failure = f"rank {rank}, scene {scene_names[i]}: {type(exc).__name__}: {exc}"
            break
    failures: List[Optional[str]] = [None] * world
    dist.all_gather_object(failures, failure)
    if any(failures):
        raise RuntimeError("scene-parallel batch failed: " + "; ".join(f for f in failures if f))
    gathered: Optional[list] = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)
    if rank != 0:
        return None
    out: Dict[str, Dict[str, float]] = {}
    for part in gathered:
        out.update(part)
    return {k: out[k] for k in sorted(out)}
