"""Utterance-parallel inference across the GPUs of one node (one process per GPU, RCCL over xGMI).

The generator forward has no cross-utterance coupling (InstanceNorm is per (b, c), F.normalize per
row; SURVEY.md §8 e), so the path shards by UTTERANCE with no collective on the data path:

  * weights: rank 0 folds weight-norm and packs the kernel-layout blob once, then broadcasts it
    (`broadcast_packed_weights`: every kernel-layout copy of the weights) - the other ranks never
    touch the checkpoint;
  * work: the utterance list is split by a longest-processing-time greedy so that every rank gets
    the same number of frames (`shard_utterances`); each rank runs the single-GPU path on
    same-length buckets, or - `ragged=True` - on padded batches of SIMILAR length with per-utterance
    `lengths` (the kernels then keep every utterance's own zero padding and InstanceNorm length, so
    padding does not change the result);
  * results: waveforms are all-gathered in ROUNDS (`GatherSchedule`): the shard of every rank and
    its batches are a deterministic function of the frame counts, which every rank knows, so the
    shape of every round's buffer is computed locally - no metadata collective, no host
    synchronisation; round r's `all_gather_into_tensor` is issued asynchronously right after the
    rank's r-th batch and overlaps the next batch's kernels.  A rank with fewer batches (or none at
    all: fewer utterances than ranks) contributes zero rows, on ITS device.
  * staging: host-resident inputs go through two pinned staging buffers and a copy stream, so
    batch k+1 is uploaded while batch k computes (`_Stager`); device-resident inputs are stacked in
    place.

The reference has no distributed code at all; this module is new.  It is exercised
on CPU with the gloo backend in tests/test_distributed_cpu.py (world_size 2, including a rank with
an empty shard) using an injected forward function, on one GPU with the nccl backend in
tests/test_parity_gpu.py, and on N GPUs by `bench.py --workload cfg4`.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch


def shard_utterances(n_frames: Sequence[int], world_size: int) -> List[List[int]]:
    """Indices of the utterances each rank processes: longest first, each to the least loaded rank
    (ties -> lowest rank), so that the frame totals are balanced.  Deterministic on every rank."""
    order = sorted(range(len(n_frames)), key=lambda i: (-int(n_frames[i]), i))
    loads = [0] * world_size
    shards: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += int(n_frames[i])
    for s in shards:
        s.sort()
    return shards


def bucket_by_length(indices: Sequence[int], n_frames: Sequence[int]) -> Dict[int, List[int]]:
    """Group utterance indices by frame count (same-length batches; no padding - see module doc)."""
    out: Dict[int, List[int]] = {}
    for i in indices:
        out.setdefault(int(n_frames[i]), []).append(i)
    return out


def bucket_ragged(indices: Sequence[int], n_frames: Sequence[int], max_batch: int = 64,
                  pad_tolerance: float = 0.125) -> List[List[int]]:
    """Batches for the ragged path: longest first, a batch takes utterances while the shortest is
    within `pad_tolerance` of the longest (bounded padding waste) and it holds < max_batch."""
    order = sorted(indices, key=lambda i: (-int(n_frames[i]), i))
    batches: List[List[int]] = []
    for i in order:
        if batches and len(batches[-1]) < max_batch and \
                int(n_frames[i]) >= (1.0 - pad_tolerance) * int(n_frames[batches[-1][0]]):
            batches[-1].append(i)
        else:
            batches.append([i])
    return batches


def plan_batches(indices: Sequence[int], n_frames: Sequence[int], max_batch: int, ragged: bool,
                 pad_tolerance: float) -> List[List[int]]:
    """The batches one rank runs, in order (deterministic: every rank can compute every rank's)."""
    if ragged:
        return bucket_ragged(indices, n_frames, max_batch, pad_tolerance)
    out: List[List[int]] = []
    for _, idxs in sorted(bucket_by_length(indices, n_frames).items()):
        for k in range(0, len(idxs), max_batch):
            out.append(idxs[k: k + max_batch])
    return out


def broadcast_packed_weights(generator, device, src: int = 0, group=None) -> torch.Tensor:
    """Rank `src` packs its (already loaded) parameters; every rank ends up with the device blob
    installed in `generator` (RCCL broadcast when `device` is a GPU)."""
    import torch.distributed as dist
    plan = generator.plan
    if dist.get_rank(group) == src:
        blob = plan.pack(generator.state_dict()).to(device)
    else:
        blob = torch.empty(plan.blob_bytes // 4, dtype=torch.float32, device=device)
    dist.broadcast(blob, src=src, group=group)
    generator.load_packed_weights(blob)
    return blob


class GatherSchedule:
    """Round-by-round layout of the waveform all-gather, computed identically on every rank from
    the frame counts alone.  Round r carries the r-th batch of every rank: a (rows_r, C, T_r) buffer
    per rank with rows_r / T_r the largest batch / longest utterance of that round over all ranks."""

    def __init__(self, n_frames: Sequence[int], hop: int, world: int, max_batch: int, ragged: bool,
                 pad_tolerance: float, min_split: int = 8):
        self.world = world
        self.hop = hop
        self.n_frames = [int(f) for f in n_frames]
        self.shards = shard_utterances(self.n_frames, world)
        self.batches = [plan_batches(s, self.n_frames, max_batch, ragged, pad_tolerance) for s in self.shards]
        self.n_rounds = max((len(b) for b in self.batches), default=0)
        if world > 1 and self.n_rounds == 1 and max(len(b[0]) for b in self.batches if b) >= 2 * min_split:
            # one round only (e.g. BASELINE cfg4 on 8 ranks: 64 utterances each = one batch): its gather would start
            # after ALL the compute and overlap nothing - halve every rank's batch so that the first half's
            # waveforms travel while the second half computes
            self.batches = [[b[0][: (len(b[0]) + 1) // 2], b[0][(len(b[0]) + 1) // 2:]] if b else [] for b in self.batches]
            self.batches = [[c for c in b if c] for b in self.batches]
            self.n_rounds = max((len(b) for b in self.batches), default=0)
        self.rows: List[int] = []
        self.cols: List[int] = []
        for r in range(self.n_rounds):
            live = [b[r] for b in self.batches if r < len(b)]
            self.rows.append(max(len(c) for c in live))
            self.cols.append(max(self.n_frames[i] for c in live for i in c) * hop)

    def batch(self, rank: int, r: int) -> List[int]:
        return self.batches[rank][r] if r < len(self.batches[rank]) else []


def all_gather_waveforms(local: List[Tuple[int, torch.Tensor]], n_total: int, group=None,
                         device=None, channels: int = 1) -> List[Optional[torch.Tensor]]:
    """Collect (utterance index, waveform (C, T)) pairs from every rank when the ranks do NOT share
    a schedule (ad-hoc use; `run_utterance_parallel` needs no metadata exchange).  `device` /
    `channels` say where and how wide this rank's buffers are even when it holds nothing - a rank
    with an empty list must still join the collectives with tensors of the backend's device.
    ONE packed metadata gather (count, then (index, T) pairs), one host read of it, one payload
    gather padded to the longest row."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if device is None:
        if not local:
            raise ValueError("all_gather_waveforms: pass `device` (this rank holds no waveform to infer it from)")
        device = local[0][1].device
    device = torch.device(device)
    if local:
        channels = int(local[0][1].shape[0])
    cap = torch.tensor([len(local)], dtype=torch.int64, device=device)
    caps = [torch.zeros_like(cap) for _ in range(world)]
    dist.all_gather(caps, cap, group=group)
    counts = torch.stack(caps).flatten().tolist()                # the one unavoidable host read #1
    nmax = max(counts) if counts else 0
    if nmax == 0:
        return [None] * n_total
    meta = torch.full((nmax, 2), -1, dtype=torch.int64)
    for k, (i, y) in enumerate(local):
        meta[k, 0] = i
        meta[k, 1] = y.shape[-1]
    meta = meta.to(device)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    table = torch.stack(metas).tolist()                          # host read #2: every (index, T)
    tmax = max(t for m in table for _, t in m)
    rows = torch.zeros((nmax, channels, tmax), dtype=torch.float32, device=device)
    for k, (_, y) in enumerate(local):
        rows[k, :, : y.shape[-1]] = y
    gathered = _gather_rows(rows, world, group)
    out: List[Optional[torch.Tensor]] = [None] * n_total
    for r in range(world):
        for k in range(counts[r]):
            i, t = table[r][k]
            out[i] = gathered[r * nmax + k, :, :t]
    return out


def _gather_rows(rows: torch.Tensor, world: int, group, async_op: bool = False):
    """(n, C, T) per rank -> (world * n, C, T) on every rank; returns the tensor, or (tensor, work)."""
    import torch.distributed as dist
    if rows.is_cuda:
        gathered = torch.empty((world * rows.shape[0],) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
        work = dist.all_gather_into_tensor(gathered, rows, group=group, async_op=async_op)
        return (gathered, work) if async_op else gathered
    parts = [torch.empty_like(rows) for _ in range(world)]      # gloo: list form
    dist.all_gather(parts, rows, group=group)
    gathered = torch.cat(parts, dim=0)
    return (gathered, None) if async_op else gathered


class _Stager:
    """Device batches one step ahead of the compute stream.

    Host-resident utterances: rows are written into one of TWO pinned staging sets and copied on a
    side stream (`non_blocking`), so the upload of batch k+1 overlaps the kernels of batch k; a
    staging set is reused only after the copy that last read it has completed (event).  Utterances
    already on the device are stacked / padded there.  On CPU (tests) everything is plain tensors."""

    def __init__(self, utterances, device, hop: int):
        self.utts = utterances
        self.device = torch.device(device)
        self.hop = hop
        self.cuda = self.device.type == "cuda"
        self.copy_stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.pinned: List[Dict[str, torch.Tensor]] = [{}, {}]
        self.pin_free: List[Optional[torch.cuda.Event]] = [None, None]
        self.turn = 0

    @staticmethod
    def _fill(jobs) -> None:
        """Rows into the page-locked batch, zero padded.  Through numpy views: one host thread moves 12-14 GB/s that way
        (plain row memcpys), where `Tensor.copy_` into the strided destination managed 2 GB/s (its 128-thread iterator
        costs more than it brings at 1 MB per row) and a thread pool on top of either was erratic - a pass over the
        512 utterances of 2 - 10 s stages 874 MB: 615 ms then, against 125 ms of GPU work (tools/ragged_check.py)."""
        for dst, src in jobs:
            d = dst.numpy()
            a = src.numpy() if isinstance(src, torch.Tensor) else np.asarray(src)
            n = a.shape[-1]
            d[..., :n] = a
            if n < d.shape[-1]:
                d[..., n:] = 0

    def _host_buffer(self, slot: int, key: str, shape, dtype) -> torch.Tensor:
        need = 1
        for s in shape:
            need *= int(s)
        buf = self.pinned[slot].get(key)
        if buf is None or buf.numel() < need or buf.dtype != dtype:
            buf = torch.empty(need, dtype=dtype, pin_memory=True)
            self.pinned[slot][key] = buf
        return buf[:need].view(*shape)

    def stage(self, chunk: Sequence[int], fmax: int):
        """-> (ppg (b, C, fmax), sine (b, 1, fmax*hop), lft, emb or None, ready event or None); rows
        shorter than fmax are zero padded."""
        first = self.utts[chunk[0]]
        has_emb = first.get("spk_emb") is not None
        keys = [("ppg", fmax), ("sine", fmax * self.hop), ("lft", fmax * self.hop)]
        on_device = torch.as_tensor(first["ppg"]).device == self.device and self.cuda
        if not self.cuda or on_device:
            out = []
            for key, width in keys:
                ts = [torch.as_tensor(self.utts[i][key]) for i in chunk]
                if all(t.shape[-1] == width for t in ts):
                    out.append(torch.stack(ts).to(self.device))
                    continue
                if self.cuda:
                    # ragged, on the device: one gather launch per tensor (csrc/fastsvc_stage.hip) - 64 separate copies
                    # queue at ~15 us apiece behind a busy stream: 42 ms of the 185 ms pass over the 2 - 10 s set
                    # (tools/ragged_check.py)
                    from .engine import gather_padded
                    out.append(gather_padded([t.to(torch.float32) for t in ts], width))
                    continue
                batch = torch.zeros((len(ts),) + tuple(ts[0].shape[:-1]) + (width,), dtype=ts[0].dtype)
                for j, t in enumerate(ts):
                    batch[j, ..., : t.shape[-1]].copy_(t)
                out.append(batch)
            emb = torch.stack([torch.as_tensor(self.utts[i]["spk_emb"]) for i in chunk]).to(self.device) if has_emb else None
            return out[0], out[1], out[2], emb, None
        slot = self.turn & 1
        self.turn += 1
        if self.pin_free[slot] is not None:
            self.pin_free[slot].synchronize()                      # the copy that last read this set is done
        host = {}
        jobs = []
        for key, width in keys:
            t0 = torch.as_tensor(first[key])
            hb = self._host_buffer(slot, key, (len(chunk), t0.shape[0], width), t0.dtype)
            for j, i in enumerate(chunk):
                jobs.append((hb[j], torch.as_tensor(self.utts[i][key])))
            host[key] = hb
        self._fill(jobs)
        if has_emb:
            e0 = torch.as_tensor(first["spk_emb"])
            hb = self._host_buffer(slot, "spk_emb", (len(chunk), e0.shape[0]), e0.dtype)
            for j, i in enumerate(chunk):
                hb[j] = torch.as_tensor(self.utts[i]["spk_emb"])
            host["spk_emb"] = hb
        compute = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.copy_stream):
            dev = {k: v.to(self.device, non_blocking=True) for k, v in host.items()}
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        # The device batch is allocated from the COPY stream's pool but consumed by kernels on the compute stream
        # (which only waits on `ready`): tell the caching allocator, or the blocks go back to the copy-stream pool
        # when the caller drops the batch and a later stage() may start its H2D copy into them while batch r's
        # kernels are still queued (the host runs ahead of the GPU) - ADVICE r2.
        for t in dev.values():
            t.record_stream(compute)
        self.pin_free[slot] = ready
        return dev["ppg"], dev["sine"], dev["lft"], dev.get("spk_emb"), ready


def run_utterance_parallel(forward_fn: Callable[..., torch.Tensor],
                           utterances: Sequence[Optional[dict]], device, max_batch: int = 64, group=None,
                           ragged: bool = False, pad_tolerance: float = 0.125,
                           out_channels: int = 1, n_frames: Optional[Sequence[int]] = None,
                           hop: Optional[int] = None, forward_into: bool = False) -> List[Optional[torch.Tensor]]:
    """Shard `utterances` (dicts with 'ppg' (C,F), 'sine' (1,T), 'lft' (1,T), optional 'spk_emb'
    (E,)) over the ranks, run `forward_fn(ppg, sine, lft, emb)` on same-length batches - or, with
    `ragged`, `forward_fn(ppg, sine, lft, emb, lengths)` on zero-padded batches of similar length -
    and all-gather the waveforms ((out_channels, T) each, indexed like `utterances`).

    Every rank passes the same list; only its own shard is touched, so entries of other ranks'
    utterances may be None when `n_frames` (frame count of every utterance) and `hop` are given.
    `forward_into`: `forward_fn` accepts `out=` (a (b, C, T) float32 view) and writes its result there - the
    waveforms then land in the gather's send buffer without a copy."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    device = torch.device(device)
    if n_frames is None:
        n_frames = [int(u["ppg"].shape[-1]) for u in utterances]
    n_total = len(n_frames)
    if n_total == 0:
        return []
    if hop is None:
        u0 = next(u for u in utterances if u is not None)
        hop = int(u0["sine"].shape[-1]) // int(u0["ppg"].shape[-1])
    sched = GatherSchedule(n_frames, hop, world, max_batch, ragged, pad_tolerance)
    mine = sched.batches[rank]
    stager = _Stager(utterances, device, hop)
    staged = stager.stage(mine[0], max(sched.n_frames[i] for i in mine[0])) if mine else None
    rounds: List[Tuple[torch.Tensor, object]] = []
    for r in range(sched.n_rounds):
        # this rank's rows of round r's gather: (largest batch, C, longest utterance) over all ranks; what the
        # forward does not write (fewer / shorter utterances than the round's maximum, or no batch at all) is zeroed
        rows = torch.empty((sched.rows[r], out_channels, sched.cols[r]), dtype=torch.float32, device=device)
        if r < len(mine):
            chunk = mine[r]
            ppg, sine, lft, emb, ready = staged
            if r + 1 < len(mine):                                  # upload the next batch meanwhile
                staged = stager.stage(mine[r + 1], max(sched.n_frames[i] for i in mine[r + 1]))
            if ready is not None:
                torch.cuda.current_stream(device).wait_event(ready)
            nb, width = len(chunk), int(sine.shape[-1])
            # the forward writes STRAIGHT into the send buffer when its output is a contiguous leading block of it
            # (always, for equal-length sets such as BASELINE cfg4) and the forward function takes `out=`
            direct = forward_into and width == sched.cols[r]
            kw = {"out": rows[:nb]} if direct else {}
            if ragged:
                lens = [sched.n_frames[i] for i in chunk]
                y = forward_fn(ppg, sine, lft, emb, lens, **kw)
            else:
                y = forward_fn(ppg, sine, lft, emb, **kw)
            if y.shape[1] != out_channels:
                raise ValueError(f"forward_fn returned {y.shape[1]} channels, out_channels={out_channels}")
            if direct and y.data_ptr() != rows.data_ptr():
                raise ValueError("forward_fn ignored `out=`: pass forward_into=False")
            if not direct:
                rows[:nb, :, :width] = y
                if width < sched.cols[r]:
                    rows[:nb, :, width:].zero_()
            if nb < sched.rows[r]:
                rows[nb:].zero_()
        else:
            rows.zero_()
        rounds.append(_gather_rows(rows, world, group, async_op=True))
    out: List[Optional[torch.Tensor]] = [None] * n_total
    for r, (gathered, work) in enumerate(rounds):
        if work is not None:
            work.wait()
        for rk in range(world):
            for k, i in enumerate(sched.batch(rk, r)):
                out[i] = gathered[rk * sched.rows[r] + k, :, : sched.n_frames[i] * hop]
    return out
