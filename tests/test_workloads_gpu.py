"""Full-size workloads under pytest -m gpu (BASELINE.json configs[2] and the single-GPU leg of
configs[3]): cfg3 = 64 x 10 s in float32 AND bfloat16 storage with the SHIPPED launch-shape table
(`|64|...` and `...|b` entries), and the utterance-parallel host path with the nccl (RCCL) backend
on the real generator.  The oracle checks utterances run ALONE (batch items are independent -
SURVEY.md §8 e - and one 10 s utterance costs the CPU oracle about a second); everything at full
size is a size-independent property: shape, finiteness, determinism, batch-permutation
equivariance, batch item == the same utterance alone."""
import os
import socket

import numpy as np
import pytest
import torch

import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S

pytestmark = pytest.mark.gpu

TIGHT = 1e-4
# bf16 products and activations: observed mean-abs 1.25e-2 / max-abs 0.11 on an output of rms 0.7 (the
# reference's own bf16 autocast sits at 1.3e-2 / 0.2; SURVEY 8c proposes the ceiling 2e-2 / 0.3)
BF16_MEAN, BF16_MAX = 2e-2, 0.25


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (and fail loudly without one)"
    A.load_library()
    return torch.device("cuda:0")


def _oracle_alone(sd_folded, cfg, ins, i):
    from oracle import fastsvc_oracle as O
    one = [t[i:i + 1].cpu().numpy() for t in ins]
    return O.forward_dedup(sd_folded, cfg.upsampling_scales, *one)


@pytest.mark.parametrize("storage", ["float32", "bfloat16"])
def test_cfg3_full_size_with_the_shipped_table(dev, storage):
    cfg = S.FULL_CONFIG
    wl = S.WORKLOADS["cfg3"]
    B, F = wl["B"], wl["F"]
    T = F * cfg.hop
    sd = S.synth_state_dict(cfg, 201)
    plan = A.Plan(cfg, storage=storage)                   # loads svcc23_fastsvc_amd/tuned_mi355x.json
    suffix = "|b" if storage == "bfloat16" else ""
    keys = [k for k in plan.tuned_shapes() if k.split("|")[1] == str(B) and k.endswith("|b") == (storage == "bfloat16")]
    assert len(keys) >= 30, f"the shipped table has no |{B}|...{suffix} entries for cfg3"
    blob = plan.pack(sd).to(dev)
    ins = list(S.device_batch(cfg, B, F, wl["seed"], dev))
    ws = torch.empty(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    y1 = plan.forward(blob, *ins, workspace=ws).clone()
    assert tuple(y1.shape) == (B, 1, T) and bool(torch.isfinite(y1).all())
    rms = float(y1.pow(2).mean().sqrt())
    assert 0.05 < rms < 10.0
    # determinism (f64 atomics jitter only) and batch-permutation equivariance at full size
    y2 = plan.forward(blob, *ins, workspace=ws)
    jitter = 1e-5 if storage == "float32" else 2e-2       # a flipped bf16 rounding moves an element by one bf16 ulp
    assert float((y1 - y2).abs().max()) <= jitter
    perm = torch.roll(torch.arange(B, device=dev), 17)
    yp = plan.forward(blob, *[t[perm].contiguous() for t in ins], workspace=ws)
    assert float((y1[perm] - yp).abs().max()) <= 2 * jitter
    # first and last utterance against the oracle run alone, and against the HIP path run alone
    wf = S.fold_weight_norm(sd)
    for i in (0, B - 1):
        ref = _oracle_alone(wf, cfg, ins, i)
        err = (y1[i:i + 1].cpu() - ref).abs()
        if storage == "float32":
            assert float(err.max()) <= TIGHT * max(1.0, float(ref.abs().max()))
        else:
            assert float(err.mean()) <= BF16_MEAN and float(err.max()) <= BF16_MAX
        alone = plan.forward(blob, *[t[i:i + 1] for t in ins])
        # (bfloat16: other tile shapes round other elements; a few bf16 ulps of the unit-scale waveform)
        assert float((alone - y1[i:i + 1]).abs().max()) <= (2e-5 if storage == "float32" else 1e-1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_utterance_parallel_path_on_the_nccl_backend(dev):
    """broadcast -> shard -> forward -> all-gather with backend="nccl" (RCCL), world_size 1, the real
    generator and HOST-resident utterances of three lengths (pinned double-buffered staging): every
    gathered waveform equals the utterance run alone; same-length buckets and ragged batches."""
    import torch.distributed as dist
    from svcc23_fastsvc_amd import distributed as D
    cfg = S.FULL_CONFIG
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        g = A.FastSVCGenerator(in_channels=cfg.in_channels, mid_channels=list(cfg.mid_channels),
                               upsampling_scales=list(cfg.upsampling_scales), out_channels=1,
                               spk_emb_size=cfg.spk_emb_size, use_spk_emb=True)
        g.load_state_dict({k: torch.from_numpy(v) for k, v in S.synth_state_dict(cfg, 55).items()})
        g = g.eval()                                     # parameters stay on the host: only the blob travels
        blob = D.broadcast_packed_weights(g, dev, src=0)
        assert blob.is_cuda and blob.numel() * 4 == g.plan.blob_bytes
        utts = []
        for i, F in enumerate([40, 48, 45, 40, 47, 44, 41, 36, 48, 37, 44]):        # (odd lengths too: padded to 4 frames, run ragged)
            b = S.synth_batch(cfg, 1, F, 300 + i)
            utts.append(dict(ppg=torch.from_numpy(b.ppg[0]), sine=torch.from_numpy(b.sine[0]),
                             lft=torch.from_numpy(b.lft[0]), spk_emb=torch.from_numpy(b.spk_emb[0])))
        plan = g.plan

        def fwd(ppg, sine, lft, emb, lengths=None, out=None):
            return plan.forward(blob, ppg, sine, lft, emb, lengths=lengths, out=out)

        with torch.no_grad():
            ys = D.run_utterance_parallel(fwd, utts, dev, max_batch=3)
            yr = D.run_utterance_parallel(fwd, utts, dev, max_batch=4, ragged=True, pad_tolerance=0.3)
            # device-resident utterances take the in-place stacking route
            utts_dev = [{k: v.to(dev) for k, v in u.items()} for u in utts]
            yd = D.run_utterance_parallel(fwd, utts_dev, dev, max_batch=64)
            # ... and, ragged, the batch-assembly kernel (fastsvc_gather_padded), the forward writing straight into
            # the gather's send buffer
            ydr = D.run_utterance_parallel(fwd, utts_dev, dev, max_batch=4, ragged=True, pad_tolerance=0.3, forward_into=True)
            yhr = D.run_utterance_parallel(fwd, utts, dev, max_batch=5, ragged=True, pad_tolerance=0.3, forward_into=True)
            for i, u in enumerate(utts):
                alone = plan.forward(blob, u["ppg"][None].to(dev), u["sine"][None].to(dev), u["lft"][None].to(dev),
                                     u["spk_emb"][None].to(dev))[0]
                for got in (ys[i], yr[i], yd[i], ydr[i], yhr[i]):
                    assert got.is_cuda and tuple(got.shape) == tuple(alone.shape)
                    assert float((got - alone).abs().max()) <= 2e-5
        # ad-hoc gather on the GPU backend with an EMPTY local list (used to pick a CPU tensor)
        none = D.all_gather_waveforms([], 3, device=dev, channels=1)
        assert none == [None, None, None]
    finally:
        dist.destroy_process_group()

def test_cfg4_single_gpu_leg_512_utterances_through_the_sharded_path(dev):
    """BASELINE config 4 at N = 1: the 512 x 10 s set through `run_utterance_parallel` (the multi-GPU code path:
    LPT shard, batches of 64, forward straight into the gather's send buffer, RCCL all-gather) on a 1-rank nccl
    group with the compact workspace.  Full size: shape / finiteness / every utterance present; utterances 0 and 511
    against the oracle; and the same utterance at two positions of the set gives the same waveform."""
    import torch.distributed as dist
    from svcc23_fastsvc_amd import distributed as D
    cfg = S.FULL_CONFIG
    wl = S.WORKLOADS["cfg4"]
    n_utts, F = wl["B"], wl["F"]
    T = F * cfg.hop
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sd = S.synth_state_dict(cfg, 201)
        plan = A.Plan(cfg, compact_workspace=True)
        blob = plan.pack(sd).to(dev)
        utts = []
        for c0 in range(0, n_utts, 64):
            ppg, sine, lft, emb = S.device_batch(cfg, 64, F, wl["seed"] + c0, dev)
            utts += [dict(ppg=ppg[j], sine=sine[j], lft=lft[j], spk_emb=emb[j]) for j in range(64)]
        utts[300] = utts[7]                               # the same utterance twice, in different batches
        ws = torch.empty(plan.workspace_bytes(64, F), dtype=torch.uint8, device=dev)
        calls = []

        def fwd(ppg, sine, lft, emb, out=None):
            calls.append((int(ppg.shape[0]), out is not None))
            return plan.forward(blob, ppg, sine, lft, emb, workspace=ws, out=out)

        ys = D.run_utterance_parallel(fwd, utts, dev, max_batch=64, n_frames=[F] * n_utts, hop=cfg.hop, forward_into=True)
        torch.cuda.synchronize()
        assert calls == [(64, True)] * 8                  # eight batches of 64, each written into its send buffer
        assert len(ys) == n_utts and all(tuple(y.shape) == (1, T) for y in ys)
        stack = torch.stack(ys)
        assert bool(torch.isfinite(stack).all())
        rms = stack.pow(2).mean(dim=(1, 2)).sqrt()
        assert float(rms.min()) > 0.05 and float(rms.max()) < 10.0
        assert float((ys[300] - ys[7]).abs().max()) <= 1e-5
        wf = S.fold_weight_norm(sd)
        for i in (0, n_utts - 1):
            ins = [utts[i][k][None] for k in ("ppg", "sine", "lft", "spk_emb")]
            ref = _oracle_alone(wf, cfg, ins, 0)
            assert float((ys[i][None].cpu() - ref).abs().max()) <= TIGHT * max(1.0, float(ref.abs().max()))
    finally:
        dist.destroy_process_group()


def test_staged_host_batches_survive_many_rounds(dev):
    """host-resident utterances go through pinned staging sets and a copy stream; the device batch is
    allocated on the copy stream but consumed on the compute stream.  Many rounds of EQUAL shapes (so the caching
    allocator hands the same blocks out again) with a forward that is slow on the GPU and a host that runs ahead:
    every gathered waveform must still be the function of ITS utterance."""
    import torch.distributed as dist
    from svcc23_fastsvc_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        C, F, hop, n = 8, 64, 160, 48
        g = torch.Generator().manual_seed(5)
        utts = [dict(ppg=torch.randn((C, F), generator=g), sine=torch.randn((1, F * hop), generator=g),
                     lft=torch.randn((1, F * hop), generator=g), spk_emb=torch.randn((4,), generator=g)) for _ in range(n)]
        spin = torch.zeros(1 << 22, device=dev)

        def slow_forward(ppg, sine, lft, emb):
            for _ in range(20):                           # keeps the GPU busy while the host stages ahead
                spin.add_(1.0)
            return sine * 2.0 + lft + ppg.sum(dim=(1, 2), keepdim=True) + emb.sum(dim=1)[:, None, None]

        ys = D.run_utterance_parallel(slow_forward, utts, dev, max_batch=2)
        torch.cuda.synchronize()
        for i, u in enumerate(utts):
            want = u["sine"] * 2.0 + u["lft"] + u["ppg"].sum() + u["spk_emb"].sum()
            assert float((ys[i].cpu() - want).abs().max()) <= 1e-4, i
    finally:
        dist.destroy_process_group()


def test_compact_workspace_same_waveform_less_memory(dev):
    """The compact layout (intermediates of different stages share buffers; what the nn.Module mirror and bench.py
    use) gives the same waveform as the one-buffer-per-tensor layout, at 64 x 10 s in about 60 % of the memory;
    ragged batch and speaker-less path included (stream-ordered reuse of the shared slots)."""
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 33)
    full, compact = A.Plan(cfg), A.Plan(cfg, compact_workspace=True)
    assert compact.workspace_bytes(64, 1500) < 0.45 * full.workspace_bytes(64, 1500)
    blob = full.pack(sd).to(dev)
    B, F = 3, 120
    ins = list(S.device_batch(cfg, B, F, 34, dev))
    for kw in ({}, {"lengths": [120, 64, 8]}):
        y0 = full.forward(blob, *ins, **kw)
        ws = torch.full((compact.workspace_bytes(B, F),), 0xFF, dtype=torch.uint8, device=dev)     # poisoned (NaN patterns)
        y1 = compact.forward(blob, *ins, workspace=ws, **kw)
        y2 = compact.forward(blob, *ins, workspace=ws, **kw)                                        # reused workspace
        # (not the same launches: the compact plan runs conditioning stage 0 as one launch, csrc/fastsvc_cond.hip -
        # two float32-class paths, each within 1e-5 of the oracle on this output of magnitude ~7; a stale or aliased
        # buffer would show as NaN or as an error of the output's own size)
        assert float((y0 - y1).abs().max()) <= 1e-4 and float((y0 - y2).abs().max()) <= 1e-4
        assert float((y1 - y2).abs().max()) <= 2e-5
    y0 = full.forward(blob, *ins[:3], None)
    y1 = compact.forward(blob, *ins[:3], None)
    assert float((y0 - y1).abs().max()) <= 1e-4


def test_short_lived_streams_release_their_helper_streams(dev):
    """A caller that makes a stream per request.  The forward runs on that one stream (helper streams are opt-in,
    FASTSVC_STREAMS); where a context of helper streams / events was created for it (fastsvc_stream_prepare, or a
    forward under FASTSVC_STREAMS), fastsvc_stream_release frees it before the stream goes away - the context is there
    exactly once per stream, and the waveform is the same every time."""
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 35)
    plan = A.Plan(cfg)
    blob = plan.pack(sd).to(dev)
    ins = list(S.device_batch(cfg, 2, 40, 36, dev))
    want = plan.forward(blob, *ins)
    torch.cuda.synchronize()
    for _ in range(6):
        st = torch.cuda.Stream(device=dev)
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            plan.prepare_stream(dev)                       # helper streams / events of (device, st)
            y = plan.forward(blob, *ins)
        assert plan.release_stream(st) is True            # waits for the helper streams, frees them
        assert plan.release_stream(st) is False
        st.synchronize()
        assert float((y - want).abs().max()) <= 2e-5      # (InstanceNorm sums are accumulated with atomics)
    assert float((plan.forward(blob, *ins) - want).abs().max()) <= 2e-5


def test_off_table_shapes_run_close_to_their_autotuned_time(dev):
    """Decode batches are never exactly a tuned (B, F).  Without an entry a launch takes the tile shape and algorithm
    of the same layer's table entry with the nearest problem size, and the tiles per workgroup from the cost model
    (csrc/fastsvc_plan.cpp, fastsvc_plan::prior_for): held to 12 % of what on-device autotuning of the very shape
    reaches (measured: 1-3 %, tools/ragged_check.py / tools/costmodel_gap.py), for a ragged batch as the decode
    harness makes them."""
    import time
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 51)
    B, F = 30, 538                                       # padded to 540 and run ragged inside Plan.forward
    ins = list(S.device_batch(cfg, B, F, 52, dev))
    lens = [F - 2 * i for i in range(B)]

    def median_ms(plan, blob, ws):
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                plan.forward(blob, *ins, lengths=lens, workspace=ws)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 5 * 1e3)
        return sorted(ts)[len(ts) // 2]

    plan = A.Plan(cfg, compact_workspace=True)
    blob = plan.pack(sd).to(dev)
    ws = torch.empty(plan.workspace_bytes(B, plan.padded_frames(F)), dtype=torch.uint8, device=dev)
    assert not any(k.split("|")[1] == str(B) for k in plan.tuned_shapes())
    y_model = plan.forward(blob, *ins, lengths=lens, workspace=ws).clone()
    t_model = median_ms(plan, blob, ws)
    tuned = A.Plan(cfg, compact_workspace=True)
    tuned.forward(blob, *ins, workspace=ws, autotune=True)
    assert any(k.split("|")[1] == str(B) for k in tuned.tuned_shapes())
    y_tuned = tuned.forward(blob, *ins, lengths=lens, workspace=ws)
    assert float((y_model - y_tuned).abs().max()) <= 5e-5          # whatever the launch shapes, the same waveform
    # (a timing assertion on a shared machine: up to three interleaved attempts, the best ratio counts)
    ratios = []
    for _ in range(3):
        t_tuned = median_ms(tuned, blob, ws)
        ratios.append(median_ms(plan, blob, ws) / t_tuned)
        if ratios[-1] <= 1.12:
            break
    assert min(ratios) <= 1.12, (t_model, ratios)


def test_gather_padded_assembles_ragged_batches(dev):
    """csrc/fastsvc_stage.hip: utterances of different lengths (contiguous blocks and views with a row pitch,
    unaligned starts, one channel and many, more than 64 of them, widths that are not a multiple of 4) into one
    zero-padded batch - bit-exact data movement, against torch's own copies."""
    g = torch.Generator(device="cpu").manual_seed(7)
    for C, width, B in ((144, 37, 5), (1, 1003, 70), (8, 64, 64), (3, 5, 2)):
        big = torch.randn((B, C, width + 9), generator=g).to(dev)
        lens = [int(v) for v in torch.randint(0 if C > 1 else 1, width + 1, (B,), generator=g)]
        rows = []
        for b, n in enumerate(lens):
            if b % 2:
                rows.append(big[b, :, 3: 3 + n])                      # a view: pitch width + 9, unaligned start
            else:
                rows.append(big[b, :, :n].contiguous())                # its own block: pitch n
        want = torch.zeros((B, C, width), device=dev)
        for b, r in enumerate(rows):
            want[b, :, : r.shape[1]] = r
        out = torch.full((B, C, width), float("nan"), device=dev)
        got = A.gather_padded(rows, width, out=out)
        assert got.data_ptr() == out.data_ptr() and torch.equal(got, want), (C, width, B)
    with pytest.raises(ValueError):
        A.gather_padded([torch.zeros((2, 9), device=dev)], 8)          # longer than the batch is wide
    with pytest.raises(A.FastSVCError):
        A.gather_padded([torch.zeros((2, 4))], 8)                      # CPU tensors: no fallback


def test_opt_in_helper_stream_schedule_gives_the_same_waveforms(dev):
    """FASTSVC_STREAMS (read once per process) forks the FiLM nets / residual convs onto helper streams: same
    waveform as the default one-stream schedule, ragged batch and poisoned workspace included - run in a child
    process per mask, checked against this process's one-stream result."""
    import os
    import subprocess
    import sys
    import tempfile
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 71)
    plan = A.Plan(cfg)
    blob = plan.pack(sd).to(dev)
    ins = list(S.device_batch(cfg, 3, 52, 72, dev))          # (device_batch is deterministic for a seed)
    lens = [52, 33, 7]
    want = plan.forward(blob, *ins).cpu().numpy()
    want_ragged = plan.forward(blob, *ins, lengths=lens).cpu().numpy()
    child = (
        "import sys, numpy as np, torch\n"
        "import svcc23_fastsvc_amd as A\n"
        "from svcc23_fastsvc_amd import synth as S\n"
        "dev = torch.device('cuda:0'); cfg = S.FULL_CONFIG\n"
        "plan = A.Plan(cfg); blob = plan.pack(S.synth_state_dict(cfg, 71)).to(dev)\n"
        "ins = list(S.device_batch(cfg, 3, 52, 72, dev))\n"
        "ws = torch.full((plan.workspace_bytes(3, 52),), 0xFF, dtype=torch.uint8, device=dev)\n"
        "ys = [plan.forward(blob, *ins, workspace=ws).cpu().numpy() for _ in range(3)]\n"
        "yr = plan.forward(blob, *ins, lengths=[52, 33, 7], workspace=ws).cpu().numpy()\n"
        "np.savez(sys.argv[1], y0=ys[0], y2=ys[2], yr=yr)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mask in ("2", "3"):
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "y.npz")
            env = dict(os.environ, FASTSVC_STREAMS=mask, PYTHONPATH=root)
            env.pop("FASTSVC_SERIAL", None)
            r = subprocess.run([sys.executable, "-c", child, out], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            got = np.load(out)
            for k in ("y0", "y2"):
                assert float(np.abs(got[k] - want).max()) <= 2e-5, (mask, k)
            assert float(np.abs(got["yr"] - want_ragged).max()) <= 2e-5, mask
