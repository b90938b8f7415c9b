"""Utterance-parallel inference across the GPUs of one node (one process per GPU, RCCL over xGMI).

The generator forward has no cross-utterance coupling (InstanceNorm is per (b, c), F.normalize per
row; SURVEY.md §8 e), so the path shards by UTTERANCE with no collective on the data path:

  * weights: rank 0 folds weight-norm and packs the kernel-layout blob once, then broadcasts it
    (`broadcast_packed_weights`, 34.5 MB fp32: every kernel-layout copy of the weights) - the other ranks never touch the checkpoint;
  * work: the utterance list is split by a longest-processing-time greedy so that every rank gets
    the same number of frames (`shard_utterances`); each rank runs the single-GPU path on
    same-length buckets, or - `ragged=True` - on padded batches of SIMILAR length with per-utterance
    `lengths` (the kernels then keep every utterance's own zero padding and InstanceNorm length, so
    padding does not change the result);
  * results: waveforms are all-gathered (`all_gather_waveforms`): one `all_gather_into_tensor`
    when every rank holds the same shape, otherwise lengths first, then padded rows.

The reference has no distributed code at all; this module is new, and is
exercised on CPU with the gloo backend in tests/test_distributed_cpu.py (world_size 2) using an
injected forward function, and on GPUs by bench.py (`--gpus N`).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

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


def all_gather_waveforms(local: List[Tuple[int, torch.Tensor]], n_total: int, group=None
                         ) -> List[Optional[torch.Tensor]]:
    """Collect (utterance index, waveform (C, T)) pairs from every rank; returns the list indexed by
    utterance.  Equal shapes everywhere -> one all_gather_into_tensor; else lengths, then rows
    padded to the longest."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = local[0][1].device if local else torch.device("cpu")
    n_local = torch.tensor([len(local)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts) if counts else 0
    if nmax == 0:
        return [None] * n_total
    chans = local[0][1].shape[0] if local else 1
    meta = torch.full((nmax, 2), -1, dtype=torch.int64, device=dev)       # (index, T)
    for k, (i, y) in enumerate(local):
        meta[k, 0] = i
        meta[k, 1] = y.shape[-1]
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    tmax = max(int(m[:, 1].max().item()) for m in metas)
    rows = torch.zeros((nmax, chans, tmax), dtype=torch.float32, device=dev)
    for k, (_, y) in enumerate(local):
        rows[k, :, : y.shape[-1]] = y
    gathered = torch.empty((world * nmax, chans, tmax), dtype=torch.float32, device=dev)
    if dev.type == "cuda":
        dist.all_gather_into_tensor(gathered, rows, group=group)
    else:                                                 # gloo: list form
        parts = [torch.empty_like(rows) for _ in range(world)]
        dist.all_gather(parts, rows, group=group)
        gathered = torch.cat(parts, dim=0)
    out: List[Optional[torch.Tensor]] = [None] * n_total
    for r in range(world):
        for k in range(counts[r]):
            i, t = int(metas[r][k, 0].item()), int(metas[r][k, 1].item())
            out[i] = gathered[r * nmax + k, :, :t]
    return out


def run_utterance_parallel(forward_fn: Callable[..., torch.Tensor],
                           utterances: Sequence[dict], device, max_batch: int = 64, group=None,
                           ragged: bool = False, pad_tolerance: float = 0.125
                           ) -> List[Optional[torch.Tensor]]:
    """Shard `utterances` (dicts with 'ppg' (C,F), 'sine' (1,T), 'lft' (1,T), optional 'spk_emb'
    (E,)) over the ranks, run `forward_fn(ppg, sine, lft, emb)` on same-length batches - or, with
    `ragged`, `forward_fn(ppg, sine, lft, emb, lengths)` on zero-padded batches of similar length -
    and all-gather the waveforms.  Every rank passes the same list (only its shard is moved to
    `device`)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_frames = [int(u["ppg"].shape[-1]) for u in utterances]
    mine = shard_utterances(n_frames, world)[rank]
    local: List[Tuple[int, torch.Tensor]] = []
    if ragged:
        for chunk in bucket_ragged(mine, n_frames, max_batch, pad_tolerance):
            fmax = int(n_frames[chunk[0]])
            hop = int(utterances[chunk[0]]["sine"].shape[-1]) // fmax

            def pad(key, width):
                rows = []
                for i in chunk:
                    t = torch.as_tensor(utterances[i][key])
                    rows.append(torch.nn.functional.pad(t, (0, width - t.shape[-1])))
                return torch.stack(rows).to(device)

            emb = None
            if utterances[chunk[0]].get("spk_emb") is not None:
                emb = torch.stack([torch.as_tensor(utterances[i]["spk_emb"]) for i in chunk]).to(device)
            lens = [int(n_frames[i]) for i in chunk]
            y = forward_fn(pad("ppg", fmax), pad("sine", fmax * hop), pad("lft", fmax * hop), emb, lens)
            for j, i in enumerate(chunk):
                local.append((i, y[j][:, : lens[j] * hop]))
        return all_gather_waveforms(local, len(utterances), group=group)
    for _, idxs in sorted(bucket_by_length(mine, n_frames).items()):
        for k in range(0, len(idxs), max_batch):
            chunk = idxs[k: k + max_batch]
            ppg = torch.stack([torch.as_tensor(utterances[i]["ppg"]) for i in chunk]).to(device)
            sine = torch.stack([torch.as_tensor(utterances[i]["sine"]) for i in chunk]).to(device)
            lft = torch.stack([torch.as_tensor(utterances[i]["lft"]) for i in chunk]).to(device)
            emb = None
            if utterances[chunk[0]].get("spk_emb") is not None:
                emb = torch.stack([torch.as_tensor(utterances[i]["spk_emb"]) for i in chunk]).to(device)
            y = forward_fn(ppg, sine, lft, emb)
            for j, i in enumerate(chunk):
                local.append((i, y[j]))
    return all_gather_waveforms(local, len(utterances), group=group)
