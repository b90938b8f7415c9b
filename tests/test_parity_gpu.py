"""Parity tests proper (-m gpu): HIP path, called through the C ABI, vs the CPU oracle and the
golden vectors of the live reference.  Tolerance: 1e-3 max-abs in fp32 (BASELINE.json
north_star); the path actually sits at ~1e-5, so most checks use the tighter 1e-4."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S

pytestmark = pytest.mark.gpu

TOL = 1e-3          # the bar north_star states
TIGHT = 1e-4        # what we hold ourselves to (fp32 reference noise floor ~1e-5)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (and fail loudly without one)"
    A.load_library()
    return torch.device("cuda:0")


def _to(dev, *arrs):
    return [None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]


def _oracle():
    from oracle import fastsvc_oracle as O
    return O


def _module(cfg, sd, dev, fold=False):
    g = A.FastSVCGenerator(in_channels=cfg.in_channels, mid_channels=list(cfg.mid_channels),
                           upsampling_scales=list(cfg.upsampling_scales), out_channels=cfg.out_channels,
                           spk_emb_size=cfg.spk_emb_size, use_spk_emb=cfg.use_spk_emb)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    if fold:
        g.remove_weight_norm()
    return g.eval().to(dev)


def test_tiny_golden_forward_and_taps(dev):
    """Tiny-width generator vs the live reference's output AND its per-stage taps."""
    g = load_golden("tiny_forward.npz")
    cfg = S.TINY_CONFIG
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd/")}
    _, _, B, F = (int(v) for v in g["meta"])
    plan = A.Plan(cfg)
    blob = plan.pack(sd).to(dev)
    ws = torch.zeros(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    ppg, sine, lft, emb = _to(dev, g["ppg"], g["sine"], g["lft"], g["spk_emb"])
    y = plan.forward(blob, ppg, sine, lft, emb, workspace=ws)
    torch.cuda.synchronize()
    assert np.abs(y.cpu().numpy() - g["y"]).max() <= TIGHT
    n = cfg.n_stages
    for k in range(n):
        h = plan.tap(f"down_h.{k}", B, F, ws).cpu().numpy()
        for sig, sl in (("lft", slice(0, B)), ("sine", slice(B, 2 * B))):
            ref = g[f"tap/down_{sig}.{k}"]
            assert np.abs(h[sl] - ref).max() <= TIGHT * max(1.0, np.abs(ref).max()), (sig, k)
        ss = plan.tap(f"ss.{k}", B, F, ws).cpu().numpy()
        C = ss.shape[1] // 2
        ref_scale = g[f"tap/film_lft.{k}.scale"] + g[f"tap/film_sine.{k}.scale"]
        ref_shift = g[f"tap/film_lft.{k}.shift"] + g[f"tap/film_sine.{k}.shift"]
        assert np.abs(ss[:, :C] - ref_scale).max() <= TIGHT * max(1.0, np.abs(ref_scale).max())
        assert np.abs(ss[:, C:] - ref_shift).max() <= TIGHT * max(1.0, np.abs(ref_shift).max())
        ref_up = g[f"tap/up.{k}.out"]
        mine = plan.tap(f"up.{k}.out", B, F, ws).cpu().numpy()
        assert np.abs(mine - ref_up).max() <= TIGHT * max(1.0, np.abs(ref_up).max())
    # spk_emb=None path (no InstanceNorm, no speaker bias; fastsvc.py:134-140)
    y0 = plan.forward(blob, ppg, sine, lft, None, workspace=ws)
    torch.cuda.synchronize()
    ref0 = g["y_nospk"]
    assert np.abs(y0.cpu().numpy() - ref0).max() <= TIGHT * max(1.0, np.abs(ref0).max())


def test_full_width_short_utterance_golden(dev):
    """F=7: every stage is shorter than the dilation-27 receptive field (all-halo tiles)."""
    g = load_golden("full_forward_f7.npz")
    seed_w, seed_x, B, F = (int(v) for v in g["meta"])
    cfg = S.FULL_CONFIG
    m = _module(cfg, S.synth_state_dict(cfg, seed_w), dev)
    b = S.synth_batch(cfg, B, F, seed_x)
    with torch.no_grad():
        y = m(*_to(dev, b.ppg, b.sine, b.lft, b.spk_emb))
        y0 = m(*_to(dev, b.ppg, b.sine, b.lft), None)
    assert np.abs(y.cpu().numpy() - g["y"]).max() <= TIGHT
    assert np.abs(y0.cpu().numpy() - g["y_nospk"]).max() <= TIGHT * max(1.0, np.abs(g["y_nospk"]).max())


def test_cfg1_golden_weight_norm_and_folded(dev):
    """BASELINE cfg1 (1 x 2 s) vs the live reference's output; checkpoint layout (weight_g/v) and
    post-remove_weight_norm layout must give the same waveform."""
    g = load_golden("full_forward_f300.npz")
    seed_w, seed_x, B, F = (int(v) for v in g["meta"])
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, seed_w)
    b = S.synth_batch(cfg, B, F, seed_x)
    ins = _to(dev, b.ppg, b.sine, b.lft, b.spk_emb)
    with torch.no_grad():
        y_wn = _module(cfg, sd, dev)(*ins).cpu().numpy()
        y_fold = _module(cfg, sd, dev, fold=True)(*ins).cpu().numpy()
    assert y_wn.shape == (1, 1, 48000)
    assert np.abs(y_wn - g["y"]).max() <= TIGHT
    assert np.abs(y_fold - g["y"]).max() <= TIGHT
    assert np.abs(y_fold - y_wn).max() <= 1e-5


def test_cfg2_golden_slices_and_checksums(dev):
    """BASELINE cfg2 (8 x 4 s, fp32) at full size: 16 slices + per-utterance checksums of the live
    reference's output; tolerance 1e-3 as north_star states (observed ~1e-5)."""
    g = load_golden("full_forward_cfg2.npz")
    seed_w, seed_x, B, F = (int(v) for v in g["meta"])
    cfg = S.FULL_CONFIG
    m = _module(cfg, S.synth_state_dict(cfg, seed_w), dev)
    b = S.synth_batch(cfg, B, F, seed_x)
    with torch.no_grad():
        y = m(*_to(dev, b.ppg, b.sine, b.lft, b.spk_emb)).cpu().numpy()
    assert y.shape == (8, 1, 96000)
    for i, st in enumerate(g["starts"]):
        got = y[i % B, 0, st:st + 256]
        assert np.abs(got - g["slices"][i]).max() <= TOL
        assert np.abs(got - g["slices"][i]).max() <= TIGHT
    T = y.shape[-1]
    assert np.abs(y.astype(np.float64).sum(-1) - g["sum"]).max() <= 1e-4 * T ** 0.5 + 1e-2
    assert np.abs((y.astype(np.float64) ** 2).sum(-1) / g["sumsq"] - 1).max() <= 1e-4
    assert np.abs(np.abs(y).max(-1) - g["absmax"]).max() <= TIGHT


def test_inference_call_sequence_golden(dev):
    """decode_fastsvc.py:187-189: time-major single utterance through inference(); the sine is the
    reference SignalGenerator's own (noise_amp=0), fed through a stand-in signal_generator."""
    g = load_golden("inference_f40.npz")
    seed_w, seed_x, B, F = (int(v) for v in g["meta"])
    cfg = S.FULL_CONFIG
    m = _module(cfg, S.synth_state_dict(cfg, seed_w), dev, fold=True)
    b = S.synth_batch(cfg, B, F, seed_x)
    sine = torch.from_numpy(g["sine"]).to(dev)
    ppg_tm, f0_tm, lft_tm, emb = _to(dev, b.ppg[0].T, b.f0[0].T, b.lft[0].T, b.spk_emb)
    with torch.no_grad():
        y = m.inference(ppg_tm, f0_tm, lft_tm, lambda f0: sine, torch.nn.ReplicationPad1d(0), emb)
    assert y.shape == (6400, 1)
    assert np.abs(y.cpu().numpy() - g["y"]).max() <= TIGHT


@pytest.mark.parametrize("B,F,spk", [(1, 1, True), (3, 33, True), (2, 75, False), (5, 13, True)])
def test_ragged_sizes_vs_oracle(dev, B, F, spk):
    """Odd sizes: F=1 (T=160: a single partial tile), T_k not a multiple of 4 (scalar epilogue
    path at the 2F-rate stage), partial last tiles; vs the oracle on the same seeded inputs."""
    O = _oracle()
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 31 + B)
    b = S.synth_batch(cfg, B, F, 900 + F)
    m = _module(cfg, sd, dev)
    with torch.no_grad():
        y = m(*_to(dev, b.ppg, b.sine, b.lft), _to(dev, b.spk_emb)[0] if spk else None).cpu()
    ref = O.forward_dedup(S.fold_weight_norm(sd), cfg.upsampling_scales, b.ppg, b.sine, b.lft,
                          b.spk_emb if spk else None)
    assert float((y - ref).abs().max()) <= TIGHT * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("F", [41, 42, 43])
def test_frame_counts_not_multiple_of_four_stay_on_the_fast_path(dev, F):
    """T_k = F, 2F are then not float4-aligned: the pipelined kernels treat the float4 that
    straddles a row end by element (masked staging, element stores, masked InstanceNorm sums) instead
    of falling back to the scalar kernel.  Every tap must match the oracle and no conv may be
    launched on the generic kernel."""
    O = _oracle()
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 61)
    B = 3
    b = S.synth_batch(cfg, B, F, 62)
    plan = A.Plan(cfg)
    plan.keep_last_block_output(B, F)                  # the `up.3.out` tap below (else conv_last rides on up.3.d27)
    plan.keep_block_heads_separate(B, F)               # the `up.k.a` taps (else conv_first is fused with the stretched convs)
    blob = plan.pack(sd).to(dev)
    ws = torch.zeros(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    recs = []
    y = plan.forward(blob, *_to(dev, b.ppg, b.sine, b.lft, b.spk_emb), workspace=ws, profile=recs)
    torch.cuda.synchronize()
    assert not [r["layer"] for r in recs if r["kernel"].startswith("conv_mfma<")]
    ref, taps = O.forward_dedup(S.fold_weight_norm(sd), cfg.upsampling_scales, b.ppg, b.sine, b.lft,
                                b.spk_emb, return_taps=True)
    assert float((y.cpu() - ref).abs().max()) <= TIGHT
    for k in range(cfg.n_stages):
        h = plan.tap(f"down_h.{k}", B, F, ws).cpu()
        for sig, sl in (("lft", slice(0, B)), ("sine", slice(B, 2 * B))):
            want = taps[f"down_{sig}.{k}"]
            assert float((h[sl] - want).abs().max()) <= TIGHT * max(1.0, float(want.abs().max()))
    for i in range(cfg.n_stages):
        for name in ("a", "xr", "u1", "xmid", "u2", "u3", "out"):
            got = plan.tap(f"up.{i}.{name}", B, F, ws).cpu()
            want = taps[f"up.{i}.{name}"]
            assert float((got - want).abs().max()) <= TIGHT * max(1.0, float(want.abs().max())), (i, name)


@pytest.mark.parametrize("spk", [True, False])
def test_ragged_batch_equals_each_utterance_alone(dev, spk):
    """`lengths` (C ABI: device int32 frame counts): a padded batch of utterances of different
    lengths must give, for every utterance, the result of running it alone (zero padding at its OWN
    end, InstanceNorm over its OWN length) - checked against the oracle run per utterance - and
    zeros in the padding of the output.  The padding of the inputs is filled with garbage."""
    O = _oracle()
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 71)
    wf = S.fold_weight_norm(sd)
    lens = [37, 64, 5, 50]
    B, F = len(lens), max(lens)
    hop = cfg.hop
    b = S.synth_batch(cfg, B, F, 72)
    ppg, sine, lft = b.ppg.copy(), b.sine.copy(), b.lft.copy()
    for i, n in enumerate(lens):                       # poison the padding
        ppg[i, :, n:] = 1e3
        sine[i, :, n * hop:] = -7.0
        lft[i, :, n * hop:] = 9.0
    m = _module(cfg, sd, dev)
    emb = _to(dev, b.spk_emb)[0] if spk else None
    with torch.no_grad():
        y = m(*_to(dev, ppg, sine, lft), emb, lengths=lens).cpu()
    for i, n in enumerate(lens):
        ref = O.forward_dedup(wf, cfg.upsampling_scales, b.ppg[i:i + 1, :, :n], b.sine[i:i + 1, :, :n * hop],
                              b.lft[i:i + 1, :, :n * hop], b.spk_emb[i:i + 1] if spk else None)
        assert float((y[i:i + 1, :, :n * hop] - ref).abs().max()) <= TIGHT * max(1.0, float(ref.abs().max())), (i, n)
        assert float(y[i, :, n * hop:].abs().max()) == 0.0 if n < F else True


def test_batch_items_do_not_bleed(dev):
    """Zero padding is per utterance: item b of a batch == the same utterance run alone
    (InstanceNorm statistics and conv halos never cross batch items)."""
    cfg = S.FULL_CONFIG
    m = _module(cfg, S.synth_state_dict(cfg, 7), dev)
    b = S.synth_batch(cfg, 4, 50, 8)
    ins = _to(dev, b.ppg, b.sine, b.lft, b.spk_emb)
    with torch.no_grad():
        yb = m(*ins)
        for i in (0, 3):
            yi = m(*[t[i:i + 1] for t in ins])
            assert float((yb[i:i + 1] - yi).abs().max()) <= 2e-5


def test_determinism_and_batch_permutation_equivariance(dev):
    """Size-independent properties at cfg2's utterance length: determinism run-to-run within fp64
    atomics jitter, and permutation equivariance over the batch axis."""
    cfg = S.FULL_CONFIG
    m = _module(cfg, S.synth_state_dict(cfg, 9), dev)
    b = S.synth_batch(cfg, 3, 600, 10)
    ins = _to(dev, b.ppg, b.sine, b.lft, b.spk_emb)
    perm = torch.tensor([2, 0, 1], device=dev)
    with torch.no_grad():
        y1 = m(*ins)
        y2 = m(*ins)
        yp = m(*[t[perm] for t in ins])
    assert float((y1 - y2).abs().max()) <= 1e-5
    assert float((y1[perm] - yp).abs().max()) <= 2e-5


def test_output_is_affine_in_conv_last(dev):
    """The one exactly linear stage (fastsvc.py:301,330, no output non-linearity): scaling conv_last's
    weight and bias by a power of two scales the waveform by exactly that factor."""
    cfg = S.FULL_CONFIG
    sd = S.fold_weight_norm(S.synth_state_dict(cfg, 9))
    b = S.synth_batch(cfg, 2, 64, 11)
    plan = A.Plan(cfg)
    ins = _to(dev, b.ppg, b.sine, b.lft, b.spk_emb)
    y = plan.forward(plan.pack(sd).to(dev), *ins)
    sd2 = dict(sd)
    sd2["conv_last.weight"] = sd["conv_last.weight"] * 4.0
    sd2["conv_last.bias"] = sd["conv_last.bias"] * 4.0
    y4 = plan.forward(plan.pack(sd2).to(dev), *ins)
    assert float((y4 - 4.0 * y).abs().max()) <= 4e-5     # exact up to the f64-atomics jitter of the norms upstream


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7524])
def test_single_frame_utterances_over_several_seeds(dev, seed):
    """F = 1: the first up block normalises over TWO samples per channel, so (x - mean) * rstd amplifies
    rounding by up to 1/sqrt(eps) ~ 316.  With the conv epilogues' one-pass float32 partial sums a near-constant
    pair lost several per cent of its variance (seed 7524 = the randomised sweep's seed 75, case 24: 2.5e-3 of the
    output's range, with every kernel variant); batches of at most 4 frames now recompute the sums exactly in float64
    (stats_exact_kernel) and run on the f32-input kernels: 7e-5 there, held to 3e-4 here (north-star bar: 1e-3)."""
    O = _oracle()
    cfg = S.FULL_CONFIG
    if seed == 7524:
        sd = S.synth_state_dict(cfg, 975)
        b = S.synth_batch(cfg, 1, 1, 5000 + 24 + 7500)
    else:
        sd = S.synth_state_dict(cfg, 400 + seed)
        b = S.synth_batch(cfg, 2, 1, 500 + seed)
    m = _module(cfg, sd, dev)
    with torch.no_grad():
        y = m(*_to(dev, b.ppg, b.sine, b.lft, b.spk_emb)).cpu()
    ref = O.forward_dedup(S.fold_weight_norm(sd), cfg.upsampling_scales, b.ppg, b.sine, b.lft, b.spk_emb)
    assert float((y - ref).abs().max()) <= 3e-4 * max(1.0, float(ref.abs().max()))


def test_errors_mirror_reference(dev):
    cfg = S.FULL_CONFIG
    m = _module(cfg, S.synth_state_dict(cfg, 7), dev)
    b = S.synth_batch(cfg, 1, 8, 8)
    ppg, sine, lft, emb = _to(dev, b.ppg, b.sine, b.lft, b.spk_emb)
    with pytest.raises(ValueError):
        m(ppg, sine[..., :-1], lft[..., :-1], emb)          # T != F * hop
    with pytest.raises(A.FastSVCError):
        m(ppg.cpu(), sine.cpu(), lft.cpu(), emb.cpu())       # no CPU fallback
    m.train()
    with pytest.raises(NotImplementedError):
        m(ppg, sine, lft, emb, lengths=[8])                  # ragged batches are an inference extension: no backward


def test_profile_records_cover_every_launch(dev):
    cfg = S.FULL_CONFIG
    plan = A.Plan(cfg)
    blob = plan.pack(S.synth_state_dict(cfg, 3)).to(dev)
    b = S.synth_batch(cfg, 2, 20, 4)
    recs = []
    y = plan.forward(blob, *_to(dev, b.ppg, b.sine, b.lft, b.spk_emb), profile=recs)
    # launch_count is the unfused schedule; fused launches save up to 2 (stage 0) + 1 per stage (c2 -> c3) + 1 per
    # stage (FiLM conv -> heads) + 1 per block (the residual conv inside the d = 3 launch) + 1 (conv_last)
    assert plan.launch_count(True) - (3 * cfg.n_stages + 2) <= len(recs) <= plan.launch_count(True)
    assert all(r["ms"] > 0 for r in recs)
    total = sum(r["flops"] for r in recs)
    assert abs(total / (2 * 20 * 160) / plan.flops_per_sample - 1) < 0.02
    y2 = plan.forward(blob, *_to(dev, b.ppg, b.sine, b.lft, b.spk_emb))
    assert float((y - y2).abs().max()) <= 1e-5


def test_autotuned_launch_shapes_keep_parity(dev):
    """fastsvc_autotune picks different tile shapes / tiles-per-workgroup per layer; the result must
    not depend on them (same oracle tolerance, and equal to the cost-model run to fp32 noise)."""
    O = _oracle()
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 17)
    b = S.synth_batch(cfg, 3, 148, 18)                   # (a batch shape the shipped table has no entries for)
    ins = _to(dev, b.ppg, b.sine, b.lft, b.spk_emb)
    plan = A.Plan(cfg)
    assert not any(k.split("|")[1] == "3" and k.split("|")[2] in ("148", "296") for k in plan.tuned_shapes())
    blob = plan.pack(sd).to(dev)
    y0 = plan.forward(blob, *ins).cpu()
    y1 = plan.forward(blob, *ins, autotune=True).cpu()
    assert plan.last_autotune_trials > 100
    y2 = plan.forward(blob, *ins).cpu()                  # now uses the tuned shapes
    ref = O.forward_dedup(S.fold_weight_norm(sd), cfg.upsampling_scales, b.ppg, b.sine, b.lft, b.spk_emb)
    for y in (y0, y1, y2):
        assert float((y - ref).abs().max()) <= TIGHT
    assert float((y0 - y2).abs().max()) <= 2e-5


@pytest.mark.parametrize("storage", ["float32", "bfloat16"])
def test_fused_conditioning_stages_match_the_separate_launches(dev, storage):
    """A down stage's c2 -> c3 pair runs as ONE launch where the stage has the variant (kernel mode 6: the
    intermediate tile stays in LDS), stage 0 as one launch altogether (mode 7: its 1 -> C conv is computed by
    the staging waves from the raw signal), and so does the FiLM net of the narrow stages (mode 6 on the
    channel-concatenated outputs of both chains: block-diagonal conv, then the heads).  Same oracle tolerance as the separate launches, which a launch table
    with algorithm 0 under the fused keys selects; ragged batches included (the intermediate tile's zero padding
    follows each utterance's own length)."""
    O = _oracle()
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 91)
    B, F = 3, 44
    lengths = [44, 29, 8]
    b = S.synth_batch(cfg, B, F, 92)
    ins = _to(dev, b.ppg, b.sine, b.lft, b.spk_emb)
    hop = cfg.hop
    sfx = "|b" if storage == "bfloat16" else ""
    n = cfg.n_stages
    Ts, T = [], F * hop
    for k in range(n):
        T //= ([1] + list(reversed(cfg.upsampling_scales[1:])))[k]
        Ts.append(T)
    unfused = {f"down.{k}.c23|{B}|{Ts[k]}{sfx}": [3, 1, 4, 1, 0] for k in range(n)}
    unfused.update({f"film.{k}.chain|{B}|{Ts[k]}{sfx}": [3, 1, 4, 1, 0] for k in range(n)})     # FiLM conv -> heads
    unfused[f"down.0.c123|{B}|{Ts[0]}{sfx}"] = [3, 1, 4, 1, 0]
    outs = {}
    for name, table in (("fused", {}), ("separate", unfused)):
        plan = A.Plan(cfg, load_shipped_table=False, storage=storage)
        plan.load_tuned(table)
        blob = plan.pack(sd).to(dev)
        recs = []
        y = plan.forward(blob, *ins, profile=recs)
        yr = plan.forward(blob, *ins, lengths=lengths)
        modes = sorted({int(r["kernel"].split(",")[4]) for r in recs if r["kernel"].startswith("conv_hx")})
        layers = {r["layer"] for r in recs}
        assert ("film.0.chain" in layers) == (name == "fused") and ("film.0.heads" in layers) == (name != "fused")
        outs[name] = (y.cpu(), yr.cpu(), modes, len(recs))
    assert 6 in outs["fused"][2] and 7 in outs["fused"][2]
    assert 6 not in outs["separate"][2] and 7 not in outs["separate"][2]
    assert outs["fused"][3] < outs["separate"][3]
    ref = O.forward_dedup(S.fold_weight_norm(sd), cfg.upsampling_scales, b.ppg, b.sine, b.lft, b.spk_emb)
    tol = TIGHT if storage == "float32" else 0.25
    for name in outs:
        assert float((outs[name][0] - ref).abs().max()) <= tol
    close = 2e-5 if storage == "float32" else 0.25
    assert float((outs["fused"][0] - outs["separate"][0]).abs().max()) <= close
    assert float((outs["fused"][1] - outs["separate"][1]).abs().max()) <= close
    # each utterance of the ragged batch equals the same utterance run alone
    plan = A.Plan(cfg, load_shipped_table=False, storage=storage)
    blob = plan.pack(sd).to(dev)
    for i, L in enumerate(lengths):
        one = plan.forward(blob, *[t[i:i + 1, ..., :L * (hop if t.shape[-1] == F * hop else 1)].contiguous() for t in ins[:3]],
                           ins[3][i:i + 1]).cpu()
        assert float((outs["fused"][1][i:i + 1, ..., :L * hop] - one).abs().max()) <= close


@pytest.mark.parametrize("storage", ["float32", "bfloat16"])
def test_conv_last_rides_on_the_last_block(dev, storage):
    """conv_last (fastsvc.py:301,330; C -> 1) is computed by the epilogue of the last block's final conv where that
    launch holds all C channels in one wave (the C-channel tensor is then never written); `keep_last_block_output`
    keeps the two launches.  Same waveform either way, ragged batches keep their zero padding."""
    O = _oracle()
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 95)
    B, F = 2, 52
    b = S.synth_batch(cfg, B, F, 96)
    ins = _to(dev, b.ppg, b.sine, b.lft, b.spk_emb)
    ys = []
    for keep in (False, True):
        plan = A.Plan(cfg, load_shipped_table=False, storage=storage)
        if keep:
            plan.keep_last_block_output(B, F)
        blob = plan.pack(sd).to(dev)
        recs = []
        y = plan.forward(blob, *ins, profile=recs).cpu()
        yr = plan.forward(blob, *ins, lengths=[52, 20]).cpu()
        assert ("conv_last" in {r["layer"] for r in recs}) == keep
        total = sum(r["flops"] for r in recs)
        assert abs(total / (B * F * cfg.hop) / plan.flops_per_sample - 1) < 0.02       # conv_last's work is still accounted
        assert float(yr[1, :, 20 * cfg.hop:].abs().max()) == 0.0
        ys.append((y, yr))
    ref = O.forward_dedup(S.fold_weight_norm(sd), cfg.upsampling_scales, b.ppg, b.sine, b.lft, b.spk_emb)
    tol = TIGHT if storage == "float32" else 0.25
    assert float((ys[0][0] - ref).abs().max()) <= tol
    close = 2e-5 if storage == "float32" else 0.1
    assert float((ys[0][0] - ys[1][0]).abs().max()) <= close
    assert float((ys[0][1] - ys[1][1]).abs().max()) <= close


@pytest.mark.parametrize("F,expect_poly", [(40, (True, True, True, True)), (42, (True, True, True, True)),
                                           (41, (True, True, True, True))])
def test_polyphase_stretch_convs_taps_and_fallback(dev, F, expect_poly):
    """Stretch2d + conv (upsample.py:21-50 + fastsvc.py:57-62,72-75) runs at the INPUT rate
    (kernel mode 3, s-times fewer MACs), also when the block's input length is not a multiple of 4
    (row-end handling by element); the gathered 3-tap kernel (mode 2) remains for the configurations
    the polyphase variant is not built for.  Must reproduce the oracle's xr (stretched residual
    conv) and u1 (FiLM-affined up conv) taps of every block."""
    O = _oracle()
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 41)
    B = 2
    b = S.synth_batch(cfg, B, F, 42)
    plan = A.Plan(cfg)
    plan.keep_block_heads_separate(B, F)               # (the stretched convs as launches of their own)
    blob = plan.pack(sd).to(dev)
    ws = torch.zeros(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    recs = []
    y = plan.forward(blob, *_to(dev, b.ppg, b.sine, b.lft, b.spk_emb), workspace=ws, profile=recs)
    torch.cuda.synchronize()
    ref, taps = O.forward_dedup(S.fold_weight_norm(sd), cfg.upsampling_scales, b.ppg, b.sine, b.lft,
                                b.spk_emb, return_taps=True)
    assert float((y.cpu() - ref).abs().max()) <= TIGHT
    kern = {r["layer"]: r["kernel"] for r in recs}
    for i in range(cfg.n_stages):
        for layer in (f"up.{i}.res_stretch", f"up.{i}.up_stretch"):
            # conv_mfma_ws<MW,NW,WM,WN,mode,ntaps,kind,..> / conv_hx<MW,NW,WM,WN,mode,kind,S,x3> or the generic conv_mfma<..>
            k = kern[layer]
            mode = int(k.split("<")[1].split(",")[4]) if k.startswith(("conv_mfma_ws<", "conv_hx<")) else 2
            assert mode == (3 if expect_poly[i] else 2), (layer, k)
        for name in ("xr", "u1"):
            got = plan.tap(f"up.{i}.{name}", B, F, ws).cpu()
            want = taps[f"up.{i}.{name}"]
            assert float((got - want).abs().max()) <= TIGHT * max(1.0, float(want.abs().max())), (i, name)


@pytest.mark.parametrize("spk", [True, False])
def test_fused_block_heads_match_the_separate_launches_and_the_oracle(dev, spk):
    """The head of every up block - conv_first, the stretched residual conv and the stretched up conv with its FiLM
    affine (fastsvc.py:92-97) - runs as ONE launch (kernel mode 8: the tensor `a` stays in LDS, twice: raw and
    LeakyReLU'd): xr / u1 taps, the InstanceNorm sums and the waveform against the oracle and against the three
    separate launches; ragged batch included."""
    O = _oracle()
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 71)
    B, F = 3, 44
    b = S.synth_batch(cfg, B, F, 72)
    emb = b.spk_emb if spk else None
    ins = _to(dev, b.ppg, b.sine, b.lft, emb)
    fused, sep = A.Plan(cfg), A.Plan(cfg)
    fused.fuse_block_heads(B, F)                       # (table-controlled: algorithm 3 under up.<i>.head|B|T_in)
    fused.load_tuned({f"up.{i}.head|1|{t}": [2, 1, 4, 1, 3] for i, t in enumerate((F, 2 * F, 8 * F, 32 * F))})   # the runs alone below
    sep.keep_block_heads_separate(B, F)
    blob = fused.pack(sd).to(dev)
    ref, taps = O.forward_dedup(S.fold_weight_norm(sd), cfg.upsampling_scales, b.ppg, b.sine, b.lft, emb, return_taps=True)
    ws_f = torch.zeros(fused.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    ws_s = torch.zeros(sep.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    recs = []
    y_f = fused.forward(blob, *ins, workspace=ws_f, profile=recs)
    y_s = sep.forward(blob, *ins, workspace=ws_s)
    torch.cuda.synchronize()
    layers = {r["layer"]: r["kernel"] for r in recs}
    for i in range(cfg.n_stages):
        assert f"up.{i}.head" in layers and layers[f"up.{i}.head"].split(",")[4] == "8", sorted(layers)
        assert f"up.{i}.conv_first" not in layers
    assert float((y_f.cpu() - ref).abs().max()) <= TIGHT * max(1.0, float(ref.abs().max()))
    assert float((y_f - y_s).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    for i in range(cfg.n_stages):
        for name in ("xr", "u1"):
            got, want = fused.tap(f"up.{i}.{name}", B, F, ws_f).cpu(), taps[f"up.{i}.{name}"]
            assert float((got - want).abs().max()) <= TIGHT * max(1.0, float(want.abs().max())), (i, name)
        if spk:
            st_f = fused.tap(f"up.{i}.stats", B, F, ws_f)[:B].cpu()
            st_s = sep.tap(f"up.{i}.stats", B, F, ws_s)[:B].cpu()
            assert float(((st_f - st_s).abs() / (st_s.abs() + 1.0)).max()) <= 1e-5
    # ragged: every utterance exactly what it is alone
    lengths = [44, 28, 12]
    y_r = fused.forward(blob, *ins, lengths=lengths)
    for j, n in enumerate(lengths):
        alone = fused.forward(blob, ins[0][j:j + 1, :, :n].contiguous(), ins[1][j:j + 1, :, :n * cfg.hop].contiguous(),
                              ins[2][j:j + 1, :, :n * cfg.hop].contiguous(), None if ins[3] is None else ins[3][j:j + 1])
        assert float((y_r[j:j + 1, :, :n * cfg.hop] - alone).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_stale_launch_shape_entries_are_ignored(dev):
    """A table entry naming a tile shape the layer is not compiled for (or nonsense) must not break
    the forward: run_conv falls back to the cost model for that launch."""
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 5)
    b = S.synth_batch(cfg, 2, 40, 6)
    ins = _to(dev, b.ppg, b.sine, b.lft, b.spk_emb)
    ref_plan = A.Plan(cfg, load_shipped_table=False)
    blob = ref_plan.pack(sd).to(dev)
    y0 = ref_plan.forward(blob, *ins).cpu()
    plan = A.Plan(cfg, load_shipped_table=False)
    T3 = 40 * 160
    plan.load_tuned({f"up.3.d3|2|{T3}": [4, 4, 1, 2],          # WM=4 needs 4 channel groups: C=24 has one
                     f"up.3.up_stretch|2|{T3 // 5}": [4, 1, 4, 1],   # polyphase is not built for NW=4
                     f"film.2.heads|2|{40 * 8}": [3, 7, 9, 100],
                     f"down.2.c2_d2|2|{40 * 8}": [2, 2, 2, 1]})      # a valid one
    y1 = plan.forward(blob, *ins).cpu()
    assert float((y0 - y1).abs().max()) <= 2e-5


@pytest.mark.parametrize("algo", [1, 2])
def test_winograd_time_convs_match_oracle_taps(dev, algo):
    """The d in {1,2,4} k=3 convs of the wide layers can run as Winograd F(2,3) along time (kernel
    mode 4; algo 1 = 48-channel groups, algo 2 = 32-channel groups).  Force it on every eligible
    launch through the launch-shape table and check the taps those layers produce (down-chain
    outputs, FiLM scale/shift, conv_first) and the waveform against the oracle."""
    O = _oracle()
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 51)
    B, F = 2, 44                                   # T_k = 88 .. 7040: partial tiles at every rate
    b = S.synth_batch(cfg, B, F, 52)
    plan = A.Plan(cfg, load_shipped_table=False)
    blob = plan.pack(sd).to(dev)
    rates = {0: 160 * F, 1: 32 * F, 2: 8 * F, 3: 2 * F}
    table = {}
    for k in (1, 2, 3):                            # stage 0 has C = 24: not eligible
        for layer in ("c2_d2", "c3_d4"):
            table[f"down.{k}.{layer}|{B}|{rates[k]}"] = [1, 1, 4, 1, algo]
        table[f"down.{k}.c23|{B}|{rates[k]}"] = [3, 1, 4, 1, 0]      # the pair as two launches (algorithm 0 under the fused key)
        table[f"film.{k}.chain|{B}|{rates[k]}"] = [3, 1, 4, 1, 0]
        table[f"film.{k}.conv|{B}|{rates[k]}"] = [1, 1, 4, 2, algo]
        table[f"film.{k}.heads|{B}|{rates[k]}"] = [1, 1, 4, 1, algo]
    table[f"film.0.heads|{B}|{rates[0]}"] = [2 if algo == 1 else 1, 1, 4, 3, algo]
    table[f"film.0.chain|{B}|{rates[0]}"] = [3, 1, 4, 1, 0]
    for i, t_in in enumerate((F, 2 * F, 8 * F)):
        table[f"up.{i}.conv_first|{B}|{t_in}"] = [1, 1, 4, 1, algo]
    plan.keep_block_heads_separate(B, F)               # (conv_first as a launch of its own: a Winograd candidate)
    plan.load_tuned(table)
    ws = torch.zeros(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    recs = []
    y = plan.forward(blob, *_to(dev, b.ppg, b.sine, b.lft, b.spk_emb), workspace=ws, profile=recs)
    torch.cuda.synchronize()
    kern = {r["layer"]: r["kernel"] for r in recs}
    n_wino = sum(1 for k in kern.values() if k.startswith("conv_mfma_ws<") and k.split(",")[4] == "4")
    # eligible: down.k c2/c3 and film.k conv/heads for k = 1..3, film.0.heads, conv_first of blocks
    # 0..2 = 16 launches.  The 48-channel ones (stage 1, film.0.heads, up.2.conv_first) have no
    # 32-channel grouping: for algo 2 their entries are ignored and the cost model picks for them
    # (the half-precision MFMA kernel, conv_hx), so 11 launches run Winograd with 32-channel groups (MW = 2).
    assert n_wino == (16 if algo == 1 else 11), sorted(kern.items())
    n_mw2 = sum(1 for k in kern.values() if k.startswith("conv_mfma_ws<2,") and k.split(",")[4] == "4")
    assert n_mw2 == (0 if algo == 1 else 11), sorted(kern.items())
    ref, taps = O.forward_dedup(S.fold_weight_norm(sd), cfg.upsampling_scales, b.ppg, b.sine, b.lft,
                                b.spk_emb, return_taps=True)
    assert float((y.cpu() - ref).abs().max()) <= TIGHT

    def close(got, want, what):
        assert float((got.cpu() - want).abs().max()) <= TIGHT * max(1.0, float(want.abs().max())), what

    for k in range(cfg.n_stages):
        h = plan.tap(f"down_h.{k}", B, F, ws)
        close(h[:B], taps[f"down_lft.{k}"], f"down_lft.{k}")
        close(h[B:], taps[f"down_sine.{k}"], f"down_sine.{k}")
        ss = plan.tap(f"ss.{k}", B, F, ws)
        C = ss.shape[1] // 2
        close(ss[:, :C], taps[f"scale.{k}"], f"scale.{k}")
        close(ss[:, C:], taps[f"shift.{k}"], f"shift.{k}")
    for i in range(cfg.n_stages):
        close(plan.tap(f"up.{i}.a", B, F, ws), taps[f"up.{i}.a"], f"up.{i}.a")


def test_batched_decode_equals_the_reference_decode_loop(dev):
    """the decode harness (F0 mean shift -> device excitation -> ONE ragged batch of the
    three utterances) against tests/golden/decode_chain.npz, which decode_fastsvc.py:160-189's
    one-utterance-at-a-time loop produced on the live reference (noise_amp = 0)."""
    from svcc23_fastsvc_amd import decode as Dc
    g = load_golden("decode_chain.npz")
    cfg = S.FULL_CONFIG
    seed_w = int(g["meta"][0])
    frames = [int(v) for v in g["frames"]]
    batches = [S.synth_batch(cfg, 1, F, 400 + i) for i, F in enumerate(frames)]
    feats = [dict(f0=b.f0[0].T.copy(), ppg=b.ppg[0].T.copy(), lft=b.lft[0].T.copy()) for b in batches]
    m = _module(cfg, S.synth_state_dict(cfg, seed_w), dev, fold=True)
    sg = A.SignalGenerator(sample_rate=24000, hop_size=cfg.hop, sine_amp=0.1, noise_amp=0.0, signal_types=["sine"])
    ys = Dc.decode_utterances(m, feats, sg, dev, trg_emb=batches[0].spk_emb,
                              src_f0_stats=[g["srcstats"]] * 3, trg_f0_stats=g["trgstats"],
                              max_batch=8, pad_tolerance=0.9)               # all three in one batch
    for i, y in enumerate(ys):
        want = g[f"y.{i}"]
        assert y.shape == want.shape
        # fp32 phase of the excitation differs from the reference's fp32 cumsum by < 1e-4 cycles here
        assert float(np.abs(y - want).max()) <= TOL, i
    pcm = [Dc.to_pcm16(y) for y in ys]
    assert all(p.dtype == np.int16 and len(p) == f * cfg.hop for p, f in zip(pcm, frames))


def test_pipelined_decode_is_batching_invariant(dev):
    """The harness overlaps staging, upload, compute and download across batches with two page-locked sets each
    way: many small batches (every set reused several times), a few large ones and one utterance at a time must give
    the same waveforms - slot reuse before a copy has finished would show up here."""
    from svcc23_fastsvc_amd import decode as Dc
    cfg = S.FULL_CONFIG
    frames = [31, 7, 25, 26, 18, 40, 12, 33, 9, 21, 38]
    rng = np.random.default_rng(11)
    feats = []
    for f in frames:
        f0 = np.where(rng.random((f, 1)) < 0.3, 0.0, rng.uniform(80, 400, (f, 1)))
        feats.append(dict(f0=f0, ppg=rng.standard_normal((f, cfg.in_channels)).astype(np.float32),
                          lft=rng.uniform(-9, 1, (f * cfg.hop, 1)).astype(np.float32)))
    m = _module(cfg, S.synth_state_dict(cfg, 12), dev, fold=True)
    sg = A.SignalGenerator(sample_rate=24000, hop_size=cfg.hop, sine_amp=0.1, noise_amp=0.0, signal_types=["sine"])
    emb = rng.standard_normal(cfg.spk_emb_size).astype(np.float32)
    kw = dict(trg_emb=emb, src_f0_stats=[[5.0, 1.0]] * len(feats), trg_f0_stats=[5.2, 1.0])
    alone = [Dc.decode_utterances(m, [u], sg, dev, trg_emb=emb, src_f0_stats=[[5.0, 1.0]], trg_f0_stats=[5.2, 1.0])[0] for u in feats]
    for max_batch, tol in ((2, 0.125), (3, 0.9), (16, 0.9)):
        for _ in range(2):                                   # (a second pass reuses warm buffers)
            ys = Dc.decode_utterances(m, feats, sg, dev, max_batch=max_batch, pad_tolerance=tol, **kw)
            for i, (y, a) in enumerate(zip(ys, alone)):
                assert y.shape == a.shape == (frames[i] * cfg.hop,)
                assert float(np.abs(y - a).max()) <= 2e-5, (max_batch, i)
    assert Dc.decode_utterances(m, [], sg, dev) == []


def test_bfloat16_activation_storage_mode(dev):
    """BASELINE config 3's dtype: workspace tensors stored as bfloat16 and the convolutions multiplied on
    the bf16 MFMA (bf16-rounded activations AND weights, f32 accumulation; InstanceNorm statistics in
    f64) - what the reference does under torch.autocast(bfloat16).  Not the 1e-3 parity path: the bound
    is ~2x what is observed (mean-abs 1.25e-2, max-abs 0.11 on an output of rms 0.66; the reference's own
    bf16 autocast sits at 1.3e-2 / 0.2 and SURVEY 8(c)'s proposed ceiling is 2e-2 / 0.3), and the
    float32 path must stay an order of magnitude closer.  Also: half the workspace, bf16 taps,
    ragged batches work, frame counts that are not multiples of 4 are padded and run as ragged batches."""
    O = _oracle()
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 81)
    wf = S.fold_weight_norm(sd)
    B, F = 2, 48
    b = S.synth_batch(cfg, B, F, 82)
    ins = _to(dev, b.ppg, b.sine, b.lft, b.spk_emb)
    ref = O.forward_dedup(wf, cfg.upsampling_scales, b.ppg, b.sine, b.lft, b.spk_emb)
    p32 = A.Plan(cfg)
    p16 = A.Plan(cfg, storage="bfloat16")
    assert p16.workspace_bytes(B, F) < 0.56 * p32.workspace_bytes(B, F)
    blob = p32.pack(sd).to(dev)
    ws = torch.zeros(p16.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    y16 = p16.forward(blob, *ins, workspace=ws).cpu()
    y32 = p32.forward(blob, *ins).cpu()
    e16, e32 = (y16 - ref).abs(), (y32 - ref).abs()
    assert float(e16.mean()) <= 2e-2 and float(e16.max()) <= 0.25
    assert float(e32.max()) <= TIGHT and float(e16.mean()) > 10 * float(e32.mean())     # it really is a different mode
    tap = p16.tap("up.3.out", B, F, ws)
    assert tap.dtype == torch.bfloat16 and tuple(tap.shape) == (B, 24, F * 160)
    assert p16.tap("up.3.stats", B, F, ws).dtype == torch.float64
    yr = p16.forward(blob, *ins, lengths=[36, 48]).cpu()                                # ragged
    r0 = O.forward_dedup(wf, cfg.upsampling_scales, b.ppg[:1, :, :36], b.sine[:1, :, :36 * 160],
                         b.lft[:1, :, :36 * 160], b.spk_emb[:1])
    assert float((yr[:1, :, :36 * 160] - r0).abs().mean()) <= 2e-2 and float(yr[0, :, 36 * 160:].abs().max()) == 0.0
    # frame counts that are not a multiple of 4 (three of four real utterances): padded and run as a ragged batch
    b2 = S.synth_batch(cfg, 2, 41, 83)
    y41 = p16.forward(blob, *_to(dev, b2.ppg, b2.sine, b2.lft, b2.spk_emb)).cpu()
    r41 = O.forward_dedup(wf, cfg.upsampling_scales, b2.ppg, b2.sine, b2.lft, b2.spk_emb)
    assert tuple(y41.shape) == (2, 1, 41 * 160)
    assert float((y41 - r41).abs().mean()) <= 2e-2 and float((y41 - r41).abs().max()) <= 0.25
    m = _module(cfg, sd, dev)
    m.activation_storage = "bfloat16"
    with torch.no_grad():
        ym = m(*ins).cpu()
    # (the module's compact-workspace plan runs conditioning stage 0 as ONE launch - csrc/fastsvc_cond.hip - whose
    # bf16 roundings fall elsewhere than the separate launches': the same distance from the oracle, not the same bits)
    em = (ym - ref).abs()
    assert float(em.mean()) <= 2e-2 and float(em.max()) <= 0.25 and float((ym - y16).abs().mean()) <= 1e-2


def test_signal_generator_matches_reference_sine(dev):
    """SignalGenerator on the GPU vs the reference's own output (golden, noise_amp=0):
    the reference accumulates the phase in fp32 (features.py:188-190), ours in f64 mod 1, so the
    comparison carries a phase tolerance: |d sine| <= 1e-3 on an amplitude-0.1 signal."""
    g = load_golden("inference_f40.npz")
    sg = A.SignalGenerator(sample_rate=24000, hop_size=160, sine_amp=0.1, noise_amp=0.0, signal_types=["sine"])
    f0 = torch.from_numpy(g["f0"]).to(dev)                           # (1, 1, 40)
    s = sg(f0)
    assert s.shape == (1, 1, 6400)
    assert float((s.cpu() - torch.from_numpy(g["sine"])).abs().max()) <= 1e-3
    # full decode call sequence with OUR signal generator: inference() vs the reference waveform
    cfg = S.FULL_CONFIG
    seed_w, seed_x, B, F = (int(v) for v in g["meta"])
    m = _module(cfg, S.synth_state_dict(cfg, seed_w), dev, fold=True)
    b = S.synth_batch(cfg, B, F, seed_x)
    ppg_tm, f0_tm, lft_tm, emb = _to(dev, b.ppg[0].T, b.f0[0].T, b.lft[0].T, b.spk_emb)
    with torch.no_grad():
        y = m.inference(ppg_tm, f0_tm, lft_tm, sg, torch.nn.ReplicationPad1d(0), emb)
    assert np.abs(y.cpu().numpy() - g["y"]).max() <= 5e-3            # sine phase tolerance propagated


def test_signal_generator_long_utterance_and_noise_statistics(dev):
    """10 s utterance: phase stays accurate where the reference's fp32 cumsum has drifted
    (compare with an f64 host restatement), voiced / unvoiced noise levels, signal-type order."""
    cfg = S.FULL_CONFIG
    f0 = S.synth_f0(2, 1500, 5)                                      # (2, 1, 1500)
    sg = A.SignalGenerator(sample_rate=24000, hop_size=160, sine_amp=0.1, noise_amp=0.003,
                           signal_types=["sine", "noise", "uv"], seed=3)
    out = sg(torch.from_numpy(f0).to(dev)).cpu().numpy().astype(np.float64)
    assert out.shape == (2, 3, 240000)
    f0u = np.repeat(f0.astype(np.float32), 160, axis=2)
    rad = ((f0u / np.float32(24000.0)) % np.float32(1.0)).astype(np.float64)
    vuv = (f0u > 0).astype(np.float64)
    clean = 0.1 * vuv * np.sin(2 * np.pi * (np.cumsum(rad, axis=2) % 1.0))
    resid = out[:, 0:1] - clean                                      # what is left is the additive noise
    v, u = vuv[:, 0] > 0, vuv[:, 0] == 0
    assert abs(resid[:, 0][v].std() / 0.003 - 1) < 0.05
    assert abs(resid[:, 0][u].std() / 0.001 - 1) < 0.05
    assert abs(resid.mean()) < 1e-4
    assert abs(out[:, 1].std() - 1) < 0.02 and abs(out[:, 1].mean()) < 0.01
    assert np.array_equal(out[:, 2], vuv[:, 0])


def test_half_precision_and_f32_mfma_families_agree(dev):
    """The two kernel families (csrc/fastsvc_hx.hip: split-binary16 products; csrc/fastsvc_kernels.hip:
    f32-input MFMA) compute the same dataflow: force each through the launch-shape table on the same
    inputs and compare the waveform and every up-block output - both fp32-class, so they must agree
    to a few 1e-6, far inside the 1e-4 the parity tests hold against the oracle."""
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 61)
    B, F = 2, 48                                       # every rate a multiple of 4: all 45 convs are eligible
    b = S.synth_batch(cfg, B, F, 62)
    ins = _to(dev, b.ppg, b.sine, b.lft, b.spk_emb)
    p_hx = A.Plan(cfg, load_shipped_table=False)       # cost model: the half-precision family wherever it exists
    p_hx.keep_last_block_output(B, F)                  # (the up.3.out tap is compared below)
    blob = p_hx.pack(sd).to(dev)
    ws_hx = torch.zeros(p_hx.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    recs = []
    y_hx = p_hx.forward(blob, *ins, workspace=ws_hx, profile=recs)
    # every conv but conv_last (stage 0 runs as one fused launch, in1_conv included)
    assert all(r["kernel"].startswith("conv_hx<") for r in recs if r["layer"] not in ("conv_last", "spk_proj", "amax_inputs")), \
        sorted((r["layer"], r["kernel"]) for r in recs)
    p_32 = A.Plan(cfg, load_shipped_table=False)
    layers = {r["layer"] for r in recs} | {f"down.{k}.{c}" for k in range(cfg.n_stages) for c in ("c2_d2", "c3_d4", "c23")} | \
             {f"film.{k}.{c}" for k in range(cfg.n_stages) for c in ("conv", "heads", "chain")} | \
             {f"up.{i}.{c}" for i in range(cfg.n_stages) for c in ("res_stretch", "d3")}     # (folded into up.<i>.d3x on the other plan)
    p_32.load_tuned({f"{layer}|{B}|{t}": [1, 1, 4, 1, 0] for layer in layers
                     for t in (F, 2 * F, 8 * F, 32 * F, 160 * F)})          # algo 0 = the f32-input MFMA kernels, unfused
    ws_32 = torch.zeros(p_32.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    recs32 = []
    y_32 = p_32.forward(blob, *ins, workspace=ws_32, profile=recs32)
    assert not any(r["kernel"].startswith("conv_hx<") for r in recs32)
    assert float((y_hx - y_32).abs().max()) <= 2e-5
    for i in range(cfg.n_stages):
        a, c = p_hx.tap(f"up.{i}.out", B, F, ws_hx), p_32.tap(f"up.{i}.out", B, F, ws_32)
        assert float((a - c).abs().max()) <= 2e-5 * max(1.0, float(c.abs().max()))


def test_ragged_batch_on_a_non_default_configuration(dev):
    """with `lengths` one utterance's row length can be a non-multiple of 4 while the padded maximum
    is one; variants compiled without the row-end handling must then not be picked.  A generator with other
    widths / scales (hop 60, the 24-channel stages at 4F and 12F) and odd frame counts: every utterance of
    the ragged batch equals the utterance run alone, and the oracle."""
    O = _oracle()
    cfg = S.GeneratorConfig.from_kwargs(in_channels=48, mid_channels=[96, 48, 24, 24], upsampling_scales=[2, 2, 3, 5],
                                        out_channels=1, spk_emb_size=32, use_spk_emb=True)
    sd = S.synth_state_dict(cfg, 71)
    lens = [13, 16, 9]
    B, F = len(lens), max(lens)
    b = S.synth_batch(cfg, B, F, 72)
    plan = A.Plan(cfg, load_shipped_table=False)
    blob = plan.pack(sd).to(dev)
    ins = _to(dev, b.ppg, b.sine, b.lft, b.spk_emb)
    y = plan.forward(blob, *ins, lengths=lens).cpu()
    wf = S.fold_weight_norm(sd)
    hop = cfg.hop
    for j, n in enumerate(lens):
        ref = O.forward_dedup(wf, cfg.upsampling_scales, b.ppg[j:j + 1, :, :n], b.sine[j:j + 1, :, : n * hop],
                              b.lft[j:j + 1, :, : n * hop], b.spk_emb[j:j + 1])
        assert float((y[j:j + 1, :, : n * hop] - ref).abs().max()) <= TIGHT * max(1.0, float(ref.abs().max())), j
        assert float(y[j, :, n * hop:].abs().max() if n < F else 0.0) == 0.0


@pytest.mark.parametrize("width,storage,spk", [("tiny", "float32", True), ("full", "float32", True), ("full", "bfloat16", True),
                                               ("full", "float32", False), ("tiny", "float32", False)])
def test_ragged_batch_with_odd_lengths_equals_every_utterance_alone(dev, width, storage, spk):
    """A padded batch with per-utterance lengths: utterance b is what it would be alone at lengths[b] frames, for
    lengths that are odd, = 2 mod 4 and multiples of 4 (rows at the frame rate and twice it then end inside a
    float4 - float32 storage runs them on the row-end instances of the split-binary16 kernels, narrow
    configurations on the gathered kernel), with a POISONED workspace and garbage in the inputs' padding: nothing
    past an utterance's end may reach its output or its InstanceNorm statistics."""
    cfg = S.TINY_CONFIG if width == "tiny" else S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 41)
    lens = [25, 22, 23, 7, 28, 9]
    B, Fp = len(lens), 28
    b = S.synth_batch(cfg, B, Fp, 42)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ppg, sine, lft = b.ppg.copy(), b.sine.copy(), b.lft.copy()
    for i, n in enumerate(lens):                        # garbage past the end of every utterance
        ppg[i, :, n:] = 1e3; sine[i, :, n * cfg.hop:] = -1e3; lft[i, :, n * cfg.hop:] = 1e3
    plan = A.Plan(cfg, storage=storage)
    blob = plan.pack(sd).to(dev)
    emb = t(b.spk_emb) if spk else None
    ws = torch.empty(plan.workspace_bytes(B, Fp) // 4 * 4 + 4, dtype=torch.uint8, device=dev)
    ws[: ws.numel() // 4 * 4].view(torch.float32).fill_(1000.0)
    y = plan.forward(blob, t(ppg), t(sine), t(lft), emb, lengths=lens, workspace=ws).cpu().numpy()
    ws.fill_(0xFF)                                      # and once more over NaN patterns
    y2 = plan.forward(blob, t(ppg), t(sine), t(lft), emb, lengths=lens, workspace=ws).cpu().numpy()
    assert np.isfinite(y2).all()
    # ... and over huge finite values: a stale 1e30 past a row's end once counted into the running maximum that scales
    # the next layer's split-binary16 staging (no-speaker path, f32-family epilogue) - found by tools/stress_parity.py
    ws[: ws.numel() // 4 * 4].view(torch.float32).fill_(1e30)
    y3 = plan.forward(blob, t(ppg), t(sine), t(lft), emb, lengths=lens, workspace=ws).cpu().numpy()
    alone = A.Plan(cfg, storage=storage)
    alone.pad_odd_lengths = False
    for i, n in enumerate(lens):
        T = n * cfg.hop
        if storage == "bfloat16" and n % 4:
            alone.pad_odd_lengths = True                # (bfloat16 storage has no unpadded route for such lengths)
        yi = alone.forward(blob, t(b.ppg[i:i + 1, :, :n]), t(b.sine[i:i + 1, :, :T]), t(b.lft[i:i + 1, :, :T]),
                           t(b.spk_emb[i:i + 1]) if spk else None).cpu().numpy()[0]
        tol = (2e-5 if spk else 2e-4) if storage == "float32" else 0.15      # (no norm: the unnormalised activations run larger)
        for got in (y, y2, y3):
            assert float(np.abs(got[i, :, :T] - yi).max()) <= tol, (i, n, float(np.abs(got[i, :, :T] - yi).max()))
            assert not got[i, :, T:].any()


def test_single_frame_member_of_a_longer_ragged_batch(dev):
    """The ill-conditioned single frame of the randomised sweep (seed 75, case 24) as one member of a 20-frame ragged
    batch: its InstanceNorm sums are recomputed exactly (stats_exact_kernel, short members of small ragged batches) -
    3.4e-3 of the output's range with the conv epilogues' float32 partial sums, 3e-6 now; the long member untouched."""
    O = _oracle()
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 975)
    wf = S.fold_weight_norm(sd)
    b1 = S.synth_batch(cfg, 1, 1, 5000 + 24 + 7500)
    b2 = S.synth_batch(cfg, 1, 20, 3)
    F = 20
    ppg = np.zeros((2, cfg.in_channels, F), np.float32)
    sine = np.zeros((2, 1, F * cfg.hop), np.float32)
    lft = np.zeros((2, 1, F * cfg.hop), np.float32)
    ppg[0], sine[0], lft[0] = b2.ppg[0], b2.sine[0], b2.lft[0]
    ppg[1, :, :1], sine[1, :, :cfg.hop], lft[1, :, :cfg.hop] = b1.ppg[0], b1.sine[0], b1.lft[0]
    emb = np.concatenate([b2.spk_emb, b1.spk_emb])
    plan = A.Plan(cfg)
    blob = plan.pack(sd).to(dev)
    y = plan.forward(blob, *_to(dev, ppg, sine, lft, emb), lengths=[20, 1]).cpu()
    r1 = O.forward_dedup(wf, cfg.upsampling_scales, b1.ppg, b1.sine, b1.lft, b1.spk_emb)
    r2 = O.forward_dedup(wf, cfg.upsampling_scales, b2.ppg, b2.sine, b2.lft, b2.spk_emb)
    assert float((y[1:2, :, :cfg.hop] - r1).abs().max()) <= 1e-4 * max(1.0, float(r1.abs().max()))
    assert float((y[0:1] - r2).abs().max()) <= 2e-5 * max(1.0, float(r2.abs().max()))
    assert not y[1, :, cfg.hop:].any()


@pytest.mark.parametrize("storage", ["float32", "bfloat16"])
def test_whole_stage_conditioning_launch_taps_vs_oracle(dev, storage):
    """Conditioning stages 0 and 1 as ONE launch each (csrc/fastsvc_cond.hip; compact-workspace plans - what the module
    and bench.py run): their outputs against the oracle's taps - ss.k = summed FiLM scale / shift of both signals
    (fastsvc.py:129-130,220-232) and down_hd.(k+1) = h_k[..., ::s] of both chains (fastsvc.py:164-193; Squeeze2d,
    upsample.py:53-74) - on a batch whose rows end inside a workgroup tile, then ragged on a poisoned workspace with
    garbage behind the inputs' row ends; the waveform against the oracle; and that the launch really ran."""
    O = _oracle()
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 61)
    wf = S.fold_weight_norm(sd)
    B, F = 3, 52                                           # T = 8320 = 34 tiles of 240 + 160 columns
    b = S.synth_batch(cfg, B, F, 62)
    plan = A.Plan(cfg, storage=storage, compact_workspace=True)
    blob = plan.pack(sd).to(dev)
    tol_t, tol_y = (TIGHT, TIGHT) if storage == "float32" else (4e-2, 0.25)
    ref, taps = O.forward_dedup(wf, cfg.upsampling_scales, b.ppg, b.sine, b.lft, b.spk_emb, return_taps=True)
    ws = torch.full((plan.workspace_bytes(B, F),), 0xFF, dtype=torch.uint8, device=dev)          # NaN patterns
    recs = []
    y = plan.forward(blob, *_to(dev, b.ppg, b.sine, b.lft, b.spk_emb), workspace=ws, profile=recs)
    layers = [r["layer"] for r in recs]
    assert "cond.0" in layers and "down.0.c123" not in layers and "film.0.chain" not in layers
    assert "cond.1" in layers and "down.1.c1_res1x1" not in layers and "film.1.chain" not in layers     # stage 1 (C = 48) likewise
    for k, dec in ((0, 5), (1, 4)):
        ss = plan.tap(f"ss.{k}", B, F, ws).float().cpu()
        want = torch.cat([taps[f"scale.{k}"], taps[f"shift.{k}"]], dim=1)
        assert float((ss - want).abs().max()) <= tol_t * max(1.0, float(want.abs().max())), k
        hd = plan.tap(f"down_hd.{k + 1}", B, F, ws).float().cpu()
        for sig, sl in (("lft", slice(0, B)), ("sine", slice(B, 2 * B))):
            wh = taps[f"down_{sig}.{k}"][..., ::dec]
            assert tuple(hd[sl].shape) == tuple(wh.shape)
            assert float((hd[sl] - wh).abs().max()) <= tol_t * max(1.0, float(wh.abs().max())), (k, sig)
    e = (y.cpu() - ref).abs()
    assert float(e.max()) <= tol_y and (storage == "float32" or float(e.mean()) <= 2e-2)
    # ragged: every utterance as if alone (lengths 52, 31, 4 frames: the last one is a single partial tile), with
    # NaN behind the row ends of the inputs and in the workspace
    lens = [52, 31, 4]
    ppg, sine, lft = b.ppg.copy(), b.sine.copy(), b.lft.copy()
    for i, n in enumerate(lens):
        ppg[i, :, n:] = np.nan; sine[i, :, n * 160:] = np.nan; lft[i, :, n * 160:] = np.nan
    ws.fill_(0xFF)
    yr = plan.forward(blob, *_to(dev, ppg, sine, lft, b.spk_emb), lengths=lens, workspace=ws).cpu()
    for i, n in enumerate(lens):
        r1 = O.forward_dedup(wf, cfg.upsampling_scales, b.ppg[i:i + 1, :, :n], b.sine[i:i + 1, :, :n * 160],
                             b.lft[i:i + 1, :, :n * 160], b.spk_emb[i:i + 1])
        ei = (yr[i:i + 1, :, :n * 160] - r1).abs()
        assert float(ei.max()) <= tol_y and (n == F or float(yr[i, :, n * 160:].abs().max()) == 0.0), i


@pytest.mark.parametrize("excitation", ["silent", "constant", "sine"])
def test_long_near_constant_rows_under_instance_norm(dev, excitation):
    """The worst inputs for the one-pass float32 InstanceNorm partial sums (VERDICT r3 task 6): ONE PPG
    frame repeated over a whole 2 s utterance, with a silent / constant / ordinary excitation - rows of the first up
    blocks whose variance is far below their squared mean (fastsvc.py:76,138: InstanceNorm2d over the whole time axis,
    eps 1e-5).  Held against the float64 oracle at the north-star bar, 1e-3 of the output's range, through the module's default
    path (compact workspace, whole-stage conditioning launch).  The "constant" case is ill-conditioned: over eight generator /
    input seeds its error ranges 5e-5 ... 8e-4 of the range, with the same spread whichever way the conditioning stages order
    their float32 sums (profiles/r6_near_constant_sweep.txt; seed 431 alone measured 1.9e-4 in round 5 and 4.8e-4 in round 6,
    seed 434 8e-4 in both); "silent" and "sine" sit at 2e-6."""
    O = _oracle()
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 431)
    B, F = 2, 300
    b = S.synth_batch(cfg, B, F, 432)
    ppg = np.repeat(b.ppg[:, :, 7:8], F, axis=2).copy()
    sine, lft = b.sine.copy(), b.lft.copy()
    if excitation == "silent":
        sine[:] = 0.0
    elif excitation == "constant":
        sine[:] = 0.05
        lft[:] = -3.0
    m = _module(cfg, sd, dev)
    with torch.no_grad():
        y = m(*_to(dev, ppg, sine, lft, b.spk_emb)).cpu().double()
    ref = O.forward_dedup(S.fold_weight_norm(sd), cfg.upsampling_scales, ppg, sine, lft, b.spk_emb, dtype=torch.float64)
    rng = float(ref.abs().max())
    err = float((y - ref).abs().max())
    assert np.isfinite(err) and err <= (1e-3 if excitation == "constant" else 3e-4) * max(1.0, rng), (excitation, err, rng)


@pytest.mark.parametrize("storage", ["float32", "bfloat16"])
def test_layer_pipelines_vs_oracle_phase_kernel_and_themselves(dev, storage):
    """Conditioning stage 0 - and, in bfloat16 storage, stage 1 (cond_stage1_pipe_kernel) - as a LAYER PIPELINE (cond_stage0_pipe_kernel, csrc/fastsvc_cond.hip: a wave owns a layer, chunks
    of an utterance stream through LDS rings; what long batches run) - forced through the launch table (algorithm 5 under
    "cond.0|B|T") on a batch small enough for the oracle: ss.0 / down_hd.1 against the oracle's taps
    (fastsvc.py:164-193,220-232; Squeeze2d upsample.py:53-74), full and ragged on a poisoned workspace; then at 8 x 600
    frames BIT-IDENTICAL to the phase kernel (algorithm 4: same products in the same order) and to itself over repeated
    runs (the float32-storage instance once returned run-to-run different tiles: a VALU -> MFMA C-operand hazard)."""
    O = _oracle()
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 61)
    wf = S.fold_weight_norm(sd)

    def plan_for(B, F, algo):
        pl = A.Plan(cfg, storage=storage, compact_workspace=True)
        T = F * cfg.hop
        pl.load_tuned({f"cond.0|{B}|{T}": [1, 1, 1, 1, algo], f"cond.0|{B}|{T}|b": [1, 1, 1, 1, algo],
                       f"cond.1|{B}|{T // 5}|b": [1, 1, 1, 1, algo]})      # stage 1's pipeline: bfloat16 storage only
        return pl

    B, F = 3, 52
    b = S.synth_batch(cfg, B, F, 62)
    plan = plan_for(B, F, 5)
    blob = plan.pack(sd).to(dev)
    tol_t, tol_y = (TIGHT, TIGHT) if storage == "float32" else (4e-2, 0.25)
    ref, taps = O.forward_dedup(wf, cfg.upsampling_scales, b.ppg, b.sine, b.lft, b.spk_emb, return_taps=True)
    ws = torch.full((plan.workspace_bytes(B, F),), 0xFF, dtype=torch.uint8, device=dev)
    recs = []
    y = plan.forward(blob, *_to(dev, b.ppg, b.sine, b.lft, b.spk_emb), workspace=ws, profile=recs)
    assert [r["kernel"] for r in recs if r["layer"] == "cond.0"][0].startswith("cond_stage0_pipe")
    assert [r["kernel"] for r in recs if r["layer"] == "cond.1"][0].startswith("cond_stage1_pipe") == (storage == "bfloat16")
    for k, dec in ((0, 5), (1, 4)):
        ss = plan.tap(f"ss.{k}", B, F, ws).float().cpu()
        want = torch.cat([taps[f"scale.{k}"], taps[f"shift.{k}"]], dim=1)
        assert float((ss - want).abs().max()) <= tol_t * max(1.0, float(want.abs().max())), k
        hd = plan.tap(f"down_hd.{k + 1}", B, F, ws).float().cpu()
        for sig, sl in (("lft", slice(0, B)), ("sine", slice(B, 2 * B))):
            wh = taps[f"down_{sig}.{k}"][..., ::dec]
            assert float((hd[sl] - wh).abs().max()) <= tol_t * max(1.0, float(wh.abs().max())), (k, sig)
    e = (y.cpu() - ref).abs()
    assert float(e.max()) <= tol_y
    lens = [52, 31, 4]
    ppg, sine, lft = b.ppg.copy(), b.sine.copy(), b.lft.copy()
    for i, n in enumerate(lens):
        ppg[i, :, n:] = np.nan; sine[i, :, n * 160:] = np.nan; lft[i, :, n * 160:] = np.nan
    ws.fill_(0xFF)
    yr = plan.forward(blob, *_to(dev, ppg, sine, lft, b.spk_emb), lengths=lens, workspace=ws).cpu()
    for i, n in enumerate(lens):
        r1 = O.forward_dedup(wf, cfg.upsampling_scales, b.ppg[i:i + 1, :, :n], b.sine[i:i + 1, :, :n * 160],
                             b.lft[i:i + 1, :, :n * 160], b.spk_emb[i:i + 1])
        assert float((yr[i:i + 1, :, :n * 160] - r1).abs().max()) <= tol_y, i
        assert n == F or float(yr[i, :, n * 160:].abs().max()) == 0.0
    # a batch long enough for many chunks per workgroup: pipeline == phase kernel == pipeline again, bit for bit
    B2, F2 = 8, 600
    ins = list(S.device_batch(cfg, B2, F2, 4321, dev))
    pipe, phase = plan_for(B2, F2, 5), plan_for(B2, F2, 4)
    wp = torch.full((pipe.workspace_bytes(B2, F2),), 0xFF, dtype=torch.uint8, device=dev)
    wq = torch.full((phase.workspace_bytes(B2, F2),), 0xFF, dtype=torch.uint8, device=dev)
    yq = phase.forward(blob, *ins, workspace=wq)
    first = None
    for _ in range(4):
        wp.fill_(0xFF)
        yp = pipe.forward(blob, *ins, workspace=wp)
        torch.cuda.synchronize()
        names = ("ss.0", "down_hd.1", "ss.1", "down_hd.2")
        cur = [pipe.tap(t, B2, F2, wp).clone() for t in names] + [yp.clone()]
        if first is None:
            first = cur
            for t, got in zip(names, cur):
                assert torch.equal(got, phase.tap(t, B2, F2, wq)), t
            assert torch.equal(yp, yq)
        else:
            assert all(torch.equal(a_, b_) for a_, b_ in zip(cur, first))


@pytest.mark.parametrize("storage", ["float32", "bfloat16"])
def test_residual_conv_folded_into_d3(dev, storage):
    """The stretched residual conv of every up block (fastsvc.py:72-75,94-100: xmid = conv_d3(lrelu(norm(u1))) +
    conv_res(stretch(a))) runs INSIDE the block's d = 3 launch (`up.<i>.d3x`, ConvParams::x2: a second, stretched operand
    accumulated with the main conv): the tensor xr is never written.  xmid / u2 / out taps and the waveform against the
    oracle (float32) and against the separate launches (both storages), the InstanceNorm sums, a ragged batch against
    every utterance alone."""
    O = _oracle()
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 91)
    B, F = 3, 52
    b = S.synth_batch(cfg, B, F, 92)
    ins = _to(dev, b.ppg, b.sine, b.lft, b.spk_emb)
    fused, sep = A.Plan(cfg, storage=storage), A.Plan(cfg, storage=storage)
    for pl in (fused, sep):
        pl.keep_last_block_output(B, F)
    sep.keep_residual_convs_separate(B, F)
    blob = fused.pack(sd).to(dev)
    ref, taps = O.forward_dedup(S.fold_weight_norm(sd), cfg.upsampling_scales, b.ppg, b.sine, b.lft, b.spk_emb, return_taps=True)
    ws_f = torch.zeros(fused.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    ws_s = torch.zeros(sep.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    recs, recs_s = [], []
    y_f = fused.forward(blob, *ins, workspace=ws_f, profile=recs)
    y_s = sep.forward(blob, *ins, workspace=ws_s, profile=recs_s)
    torch.cuda.synchronize()
    layers = {r["layer"]: r["kernel"] for r in recs}
    for i, s in enumerate(cfg.upsampling_scales):
        # conv_hx<MW,NW,WM,WN,mode 0,FiLM-affine epilogue 4,stretch factor,..>
        assert layers[f"up.{i}.d3x"].split(",")[4:7] == ["0", "4", str(s)], layers
        assert f"up.{i}.res_stretch" not in layers and f"up.{i}.d3" not in layers
        assert f"up.{i}.res_stretch" in {r["layer"] for r in recs_s}
    scale = max(1.0, float(ref.abs().max()))
    if storage == "float32":
        assert float((y_f.cpu() - ref).abs().max()) <= TIGHT * scale
        assert float((y_f - y_s).abs().max()) <= 2e-5 * scale
    else:
        # bfloat16 storage rounds xmid once in both routes but accumulates the two convs in a different order in front
        # of that rounding: the two waveforms differ like two bfloat16 runs do (mean-abs 1e-2 class)
        d = (y_f - y_s).abs()
        assert float(d.mean()) <= 1e-2 and float(d.max()) <= 0.2
        assert float((y_f.cpu() - ref).abs().mean()) <= 2e-2
    for i in range(cfg.n_stages):
        for name in ("xmid", "u2", "out"):
            got, want = fused.tap(f"up.{i}.{name}", B, F, ws_f).float().cpu(), taps[f"up.{i}.{name}"]
            other = sep.tap(f"up.{i}.{name}", B, F, ws_s).float().cpu()
            mag = max(1.0, float(want.abs().max()))
            if storage == "float32":
                assert float((got - want).abs().max()) <= TIGHT * mag, (i, name)
                assert float((got - other).abs().max()) <= 2e-5 * mag, (i, name)
            else:
                assert float((got - want).abs().max()) <= 6e-2 * mag, (i, name)
                assert float((got - other).abs().max()) <= 4e-2 * mag, (i, name)
        if storage == "float32":
            st_f = fused.tap(f"up.{i}.stats", B, F, ws_f)[:3 * B].cpu()
            st_s = sep.tap(f"up.{i}.stats", B, F, ws_s)[:3 * B].cpu()
            # (sums of u2 / u3 sit behind xmid: float32 partial sums over tensors that differ in the last bits)
            assert float(((st_f - st_s).abs() / (st_s.abs() + 1.0)).max()) <= 1e-4
    # ragged: every utterance exactly what it is alone (lengths a multiple of 4: the bfloat16 route's unpadded form)
    lengths = [52, 28, 12]
    y_r = fused.forward(blob, *ins, lengths=lengths)
    for j, n in enumerate(lengths):
        alone = fused.forward(blob, ins[0][j:j + 1, :, :n].contiguous(), ins[1][j:j + 1, :, :n * cfg.hop].contiguous(),
                              ins[2][j:j + 1, :, :n * cfg.hop].contiguous(), ins[3][j:j + 1])
        tol = 2e-5 * scale if storage == "float32" else 0.2
        assert float((y_r[j:j + 1, :, :n * cfg.hop] - alone).abs().max()) <= tol, (j, n)
        if storage == "bfloat16":
            assert float((y_r[j:j + 1, :, :n * cfg.hop] - alone).abs().mean()) <= 1e-2
        if n < F:
            assert float(y_r[j, :, n * cfg.hop:].abs().max()) == 0.0


@pytest.mark.parametrize("storage", ["float32", "bfloat16"])
def test_layer_pipelines_run_to_run_identical_at_cfg3(dev, storage):
    """The shape that exposed round 5's VALU -> MFMA C-operand hazard in the float32 pipeline (64 x 1500 frames: ~10^6 tiles
    per launch, a handful of them differed from run to run).  Since round 6 no VALU result is an MFMA's C operand in
    csrc/fastsvc_cond.hip (accumulators start from the bias register, the rank-1 / residual term joins after the products):
    six forwards, the conditioning tensors of stages 0 / 1 and the waveform bit-identical every time."""
    cfg = S.FULL_CONFIG
    B, F = 64, 1500
    plan = A.Plan(cfg, storage=storage, compact_workspace=True)
    blob = plan.pack(S.synth_state_dict(cfg, 71)).to(dev)
    ins = list(S.device_batch(cfg, B, F, 72, dev))
    ws = torch.empty(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    names = ("ss.0", "down_hd.1", "ss.1", "down_hd.2")
    first = None
    for run in range(6):
        recs = []
        y = plan.forward(blob, *ins, workspace=ws, profile=recs if run == 0 else None)
        torch.cuda.synchronize()
        if run == 0:
            assert [r["kernel"] for r in recs if r["layer"] == "cond.0"][0].startswith("cond_stage0_pipe")
        cur = [plan.tap(t, B, F, ws).clone() for t in names] + [y.clone()]
        if first is None:
            first = cur
        else:
            for t, a_, b_ in zip(names + ("y",), cur, first):
                assert torch.equal(a_, b_), (t, run, int((a_ != b_).sum()))

