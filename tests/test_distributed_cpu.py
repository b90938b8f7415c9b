"""N > 1 host path on CPU: world_size 2, gloo backend, 127.0.0.1 rendezvous.  The GPU forward is
replaced by an injected deterministic stand-in (the real one needs a GPU); what is tested here is
the sharding, the packed-weight broadcast and the waveform all-gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
from svcc23_fastsvc_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_forward(ppg, sine, lft, emb):
    """Stand-in with the generator's shape contract: (B, C, F) ... -> (B, 1, T)."""
    base = sine * 2.0 + lft
    return base + ppg.sum(dim=(1, 2), keepdim=True) + (0.0 if emb is None else emb.sum(dim=1)[:, None, None])


def _fake_forward_ragged(ppg, sine, lft, emb, lengths):
    """Ragged stand-in: per utterance, the same function of its VALID part only (like the kernels)."""
    hop = sine.shape[-1] // ppg.shape[-1]
    out = torch.zeros_like(sine)
    for j, n in enumerate(lengths):
        out[j:j + 1, :, : n * hop] = _fake_forward(ppg[j:j + 1, :, :n], sine[j:j + 1, :, : n * hop],
                                                   lft[j:j + 1, :, : n * hop], None if emb is None else emb[j:j + 1])
    return out


def _utterances():
    cfg = S.TINY_CONFIG
    out = []
    for i, F in enumerate([5, 9, 5, 3, 9, 7, 5]):
        b = S.synth_batch(cfg, 1, F, 100 + i)
        out.append(dict(ppg=b.ppg[0], sine=b.sine[0], lft=b.lft[0], spk_emb=b.spk_emb[0]))
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(rank)                      # ranks start with DIFFERENT random weights
        g = A.FastSVCGenerator(in_channels=8, mid_channels=[16, 8, 8, 4], upsampling_scales=[2, 4, 4, 5],
                               out_channels=1, spk_emb_size=16, use_spk_emb=True)
        if rank == 0:
            g.load_state_dict({k: torch.from_numpy(v) for k, v in S.synth_state_dict(S.TINY_CONFIG, 77).items()})
        blob = D.broadcast_packed_weights(g, torch.device("cpu"), src=0)
        utts = _utterances()
        ys = D.run_utterance_parallel(_fake_forward, utts, torch.device("cpu"), max_batch=2)
        yr = D.run_utterance_parallel(_fake_forward_ragged, utts, torch.device("cpu"), max_batch=3,
                                      ragged=True, pad_tolerance=0.5)
        # fewer utterances than ranks: rank 1's shard is EMPTY, it must still join every collective
        # (the gather used to fall back to a CPU tensor / 1 channel on such a rank)
        y1 = D.run_utterance_parallel(_fake_forward, utts[:1], torch.device("cpu"), max_batch=2)
        # ad-hoc gather without a shared schedule: rank 1 passes nothing, device and channels explicit
        adhoc = D.all_gather_waveforms([(0, ys[3].clone())] if rank == 0 else [], 2, device=torch.device("cpu"), channels=1)
        # an equal-length set that is ONE batch per rank (the shape of BASELINE cfg4 on 8 ranks): the schedule splits
        # the lone round in two so that the first half's gather overlaps the second half's compute, and a forward
        # that takes `out=` writes straight into the gather's send buffer
        cfg = S.TINY_CONFIG
        eq = []
        for i in range(32):
            b = S.synth_batch(cfg, 1, 6, 500 + i)
            eq.append(dict(ppg=b.ppg[0], sine=b.sine[0], lft=b.lft[0], spk_emb=b.spk_emb[0]))
        calls = []

        def fwd_into(ppg, sine, lft, emb, out=None):
            calls.append((int(ppg.shape[0]), out is not None))
            y = _fake_forward(ppg, sine, lft, emb)
            out.copy_(y)
            return out

        yq = D.run_utterance_parallel(fwd_into, eq, torch.device("cpu"), max_batch=64, forward_into=True)
        # ragged AND `out=`: a rank whose batch is as wide as the round's widest writes into the send buffer, a rank
        # with a narrower batch is called without `out` and copied (bench.py --workload cfg4var on several GPUs)
        direct = []

        def fwd_ragged_into(ppg, sine, lft, emb, lengths, out=None):
            direct.append(out is not None)
            y = _fake_forward_ragged(ppg, sine, lft, emb, lengths)
            if out is None:
                return y
            out.copy_(y)
            return out

        yrf = D.run_utterance_parallel(fwd_ragged_into, utts, torch.device("cpu"), max_batch=3, ragged=True,
                                       pad_tolerance=0.5, forward_into=True)
        q.put((rank, blob.numpy().copy(), [y.numpy().copy() for y in ys], [y.numpy().copy() for y in yr],
               [y.numpy().copy() for y in y1], [None if y is None else y.numpy().copy() for y in adhoc],
               [y.numpy().copy() for y in yq], calls, [y.numpy().copy() for y in yrf], direct))
    finally:
        dist.destroy_process_group()


def test_shard_is_balanced_and_deterministic():
    frames = [1500, 300, 900, 900, 300, 1500, 600, 600]
    shards = D.shard_utterances(frames, 4)
    assert sorted(i for s in shards for i in s) == list(range(8))
    loads = [sum(frames[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= 300
    assert shards == D.shard_utterances(frames, 4)
    assert D.shard_utterances([600] * 512, 8)[3] == list(range(3, 512, 8)) or \
        all(len(s) == 64 for s in D.shard_utterances([600] * 512, 8))       # cfg4: 64 utterances per rank


def test_two_ranks_broadcast_shard_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, blob, ys, yr, y1, adhoc, yq, calls, yrf, direct = q.get(timeout=120)
        res[rank] = (blob, ys, yr, y1, adhoc, yq, calls, yrf, direct)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank holds rank 0's packed weights
    ref_blob = A.Plan(S.TINY_CONFIG).pack(S.synth_state_dict(S.TINY_CONFIG, 77)).numpy()
    for r in range(world):
        assert np.array_equal(res[r][0], ref_blob)
    # every rank holds every waveform, equal to the single-process result, in utterance order
    utts = _utterances()
    for i, u in enumerate(utts):
        want = _fake_forward(torch.from_numpy(u["ppg"])[None], torch.from_numpy(u["sine"])[None],
                             torch.from_numpy(u["lft"])[None], torch.from_numpy(u["spk_emb"])[None])[0].numpy()
        for r in range(world):
            for got in (res[r][1][i], res[r][2][i], res[r][7][i]):      # same-length buckets, padded ragged batches (+ `out=`)
                assert got.shape == want.shape
                assert np.allclose(got, want, atol=1e-6)
    assert any(any(res[r][8]) for r in range(world))             # somebody wrote straight into a send buffer
    # the one-batch-per-rank set: two rounds of 8 utterances per rank, every forward wrote into the send buffer
    cfg = S.TINY_CONFIG
    for r in range(world):
        assert res[r][6] == [(8, True), (8, True)]
        for i in range(32):
            b = S.synth_batch(cfg, 1, 6, 500 + i)
            want = _fake_forward(*(torch.from_numpy(a) for a in (b.ppg, b.sine, b.lft, b.spk_emb)))[0].numpy()
            assert np.allclose(res[r][5][i], want, atol=1e-6)
    # the one-utterance set (empty shard on rank 1) and the schedule-free gather
    for r in range(world):
        assert len(res[r][3]) == 1 and np.allclose(res[r][3][0], res[0][1][0], atol=1e-6)
        assert res[r][4][1] is None and np.allclose(res[r][4][0], res[0][1][3], atol=1e-6)


def test_gather_schedule_is_rank_independent_and_covers_everything():
    frames = [5, 9, 5, 3, 9, 7, 5]
    for ragged in (False, True):
        s = D.GatherSchedule(frames, 160, 3, max_batch=2, ragged=ragged, pad_tolerance=0.5)
        seen = sorted(i for rk in range(3) for r in range(s.n_rounds) for i in s.batch(rk, r))
        assert seen == list(range(len(frames)))
        for r in range(s.n_rounds):
            for rk in range(3):
                b = s.batch(rk, r)
                assert len(b) <= s.rows[r]
                assert all(frames[i] * 160 <= s.cols[r] for i in b)
    empty = D.GatherSchedule([4], 160, 8, max_batch=64, ragged=False, pad_tolerance=0.125)
    assert empty.n_rounds == 1 and sum(len(empty.batch(rk, 0)) for rk in range(8)) == 1
    # BASELINE cfg4 on 8 ranks: 64 utterances per rank would be ONE round; it is split so the gather overlaps compute
    c4 = D.GatherSchedule([1500] * 512, 160, 8, max_batch=64, ragged=False, pad_tolerance=0.125)
    assert c4.n_rounds == 2 and all(len(c4.batch(rk, r)) == 32 for rk in range(8) for r in range(2))
    assert sorted(i for rk in range(8) for r in range(2) for i in c4.batch(rk, r)) == list(range(512))
    # a single rank has nothing to overlap with: one round of 64
    assert D.GatherSchedule([1500] * 64, 160, 1, max_batch=64, ragged=False, pad_tolerance=0.125).n_rounds == 1


def test_ragged_buckets_bound_the_padding():
    frames = [1500, 1490, 1400, 900, 880, 300, 1300, 1301]
    batches = D.bucket_ragged(range(len(frames)), frames, max_batch=3, pad_tolerance=0.125)
    assert sorted(i for b in batches for i in b) == list(range(len(frames)))
    for b in batches:
        assert len(b) <= 3
        assert min(frames[i] for i in b) >= 0.875 * max(frames[i] for i in b)


def test_variable_length_workload_is_deterministic_and_buckets_tightly():
    """bench.py --workload cfg4var (SURVEY 8d's variant of cfg4): 512 lengths uniform in 2 - 10 s, the same on every
    rank and run; length-bucketed into padded batches whose padding stays within the tolerance."""
    from svcc23_fastsvc_amd import synth as S
    from svcc23_fastsvc_amd import distributed as D
    f1, f2 = S.workload_frames("cfg4var"), S.workload_frames("cfg4var")
    assert f1 == f2 and len(f1) == 512 and min(f1) >= 300 and max(f1) <= 1500
    assert S.workload_frames("cfg4") == [1500] * 512
    for world in (1, 8):
        sched = D.GatherSchedule(f1, 160, world, 64, True, 0.125)
        seen = []
        for rank in range(world):
            for chunk in sched.batches[rank]:
                assert 1 <= len(chunk) <= 64
                longest, shortest = max(f1[i] for i in chunk), min(f1[i] for i in chunk)
                assert shortest >= (1.0 - 0.125) * longest - 1          # padding within the tolerance
                seen += list(chunk)
        assert sorted(seen) == list(range(512))                          # every utterance exactly once
    # LPT sharding: the ranks' total frames differ by little
    totals = [sum(f1[i] for i in D.shard_utterances(f1, 8)[r]) for r in range(8)]
    assert max(totals) - min(totals) <= 0.02 * max(totals)
