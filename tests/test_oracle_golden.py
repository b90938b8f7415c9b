"""Pin the CPU oracle against golden vectors produced by the live reference
(tests/golden/make_golden.py) - runs everywhere, no GPU, no /root/reference needed."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import fastsvc_oracle as O
from oracle.refimport import reference_available
from svcc23_fastsvc_amd import synth as S

# fp32 reference noise floor is ~1e-5 max-abs; the oracle must sit inside it.
TOL = 2e-5


def _tiny():
    g = load_golden("tiny_forward.npz")
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd/")}
    return g, sd


def test_synth_weights_regenerate_bit_exact():
    g, sd = _tiny()
    seed_w = int(g["meta"][0])
    regen = S.synth_state_dict(S.TINY_CONFIG, seed_w)
    assert list(regen.keys()) == S.state_dict_keys(S.TINY_CONFIG)
    assert set(regen) == set(sd)
    for k in sd:
        assert np.array_equal(regen[k], sd[k]), k


def test_synth_inputs_regenerate_bit_exact():
    g, _ = _tiny()
    _, seed_x, B, F = (int(v) for v in g["meta"])
    b = S.synth_batch(S.TINY_CONFIG, B, F, seed_x)
    assert np.array_equal(b.ppg, g["ppg"])
    assert np.array_equal(b.lft, g["lft"])
    assert np.array_equal(b.spk_emb, g["spk_emb"])
    # sine goes through libm sin/exp: allow one float32 ulp of 0.1
    assert np.abs(b.sine - g["sine"]).max() <= 2e-8


def test_weight_norm_fold_matches_torch():
    g = load_golden("weight_norm_fold.npz")
    names = sorted({k.rsplit(".", 1)[0] for k in g.files})
    assert len(names) == 3
    for n in names:
        folded = S.fold_weight_norm({n + ".weight_g": g[n + ".weight_g"], n + ".weight_v": g[n + ".weight_v"]})
        assert np.abs(folded[n + ".weight"] - g[n + ".weight"]).max() <= 2e-7


@pytest.mark.parametrize("variant", ["as_executed", "dedup"])
def test_tiny_forward_matches_reference(variant):
    g, sd = _tiny()
    w = S.fold_weight_norm(sd)
    fn = O.forward_as_executed if variant == "as_executed" else O.forward_dedup
    y = fn(w, S.TINY_CONFIG.upsampling_scales, g["ppg"], g["sine"], g["lft"], g["spk_emb"]).numpy()
    assert y.shape == g["y"].shape
    assert np.abs(y - g["y"]).max() <= TOL
    y0 = fn(w, S.TINY_CONFIG.upsampling_scales, g["ppg"], g["sine"], g["lft"], None).numpy()
    scale = max(1.0, float(np.abs(g["y_nospk"]).max()))
    assert np.abs(y0 - g["y_nospk"]).max() <= TOL * scale


def test_tiny_taps_match_reference():
    g, sd = _tiny()
    w = S.fold_weight_norm(sd)
    _, taps = O.forward_dedup(w, S.TINY_CONFIG.upsampling_scales, g["ppg"], g["sine"], g["lft"],
                              g["spk_emb"], return_taps=True)
    checked = 0
    for k in g.files:
        if not k.startswith("tap/"):
            continue
        name = k[4:]
        ref = g[k]
        mine = taps[name].numpy()
        if mine.ndim == 3 and ref.ndim == 4:
            ref = ref[:, :, 0, :]
        assert mine.shape == ref.shape, name
        assert np.abs(mine - ref).max() <= TOL * max(1.0, float(np.abs(ref).max())), name
        checked += 1
    assert checked == 28


def test_numpy64_restatement_agrees():
    g, sd = _tiny()
    w = S.fold_weight_norm(sd)
    y64 = O.forward_numpy64(w, S.TINY_CONFIG.upsampling_scales, g["ppg"], g["sine"], g["lft"], g["spk_emb"])
    assert np.abs(y64 - g["y"]).max() <= TOL


def _full_weights(seed_w):
    return S.fold_weight_norm(S.synth_state_dict(S.FULL_CONFIG, seed_w))


def test_full_f7_short_utterance():
    g = load_golden("full_forward_f7.npz")
    seed_w, seed_x, B, F = (int(v) for v in g["meta"])
    b = S.synth_batch(S.FULL_CONFIG, B, F, seed_x)
    w = _full_weights(seed_w)
    y = O.forward_dedup(w, S.FULL_CONFIG.upsampling_scales, b.ppg, b.sine, b.lft, b.spk_emb).numpy()
    assert np.abs(y - g["y"]).max() <= TOL
    y0 = O.forward_dedup(w, S.FULL_CONFIG.upsampling_scales, b.ppg, b.sine, b.lft, None).numpy()
    assert np.abs(y0 - g["y_nospk"]).max() <= TOL * max(1.0, float(np.abs(g["y_nospk"]).max()))


def test_full_cfg1_forward():
    g = load_golden("full_forward_f300.npz")
    seed_w, seed_x, B, F = (int(v) for v in g["meta"])
    b = S.synth_batch(S.FULL_CONFIG, B, F, seed_x)
    w = _full_weights(seed_w)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    for fn in (O.forward_as_executed, O.forward_dedup):
        y = fn(w, S.FULL_CONFIG.upsampling_scales, b.ppg, b.sine, b.lft, b.spk_emb).numpy()
        assert y.shape == (B, 1, 160 * F)
        assert np.abs(y - g["y"]).max() <= TOL


def test_inference_call_sequence_golden():
    """inference(): (F,144),(F,1),(T,1) time-major in, (T,1) out (fastsvc.py:364-383) with the
    reference SignalGenerator at noise_amp=0; the stored sine is the reference's own."""
    g = load_golden("inference_f40.npz")
    seed_w, seed_x, B, F = (int(v) for v in g["meta"])
    b = S.synth_batch(S.FULL_CONFIG, B, F, seed_x)
    w = _full_weights(seed_w)
    y = O.forward_dedup(w, S.FULL_CONFIG.upsampling_scales, b.ppg, g["sine"], b.lft, b.spk_emb).numpy()
    assert np.abs(y[0].T - g["y"]).max() <= TOL


def test_shape_invariant_is_validated():
    g, sd = _tiny()
    w = S.fold_weight_norm(sd)
    with pytest.raises(ValueError):
        O.forward_dedup(w, S.TINY_CONFIG.upsampling_scales, g["ppg"], g["sine"][..., :-1], g["lft"][..., :-1])


@pytest.mark.skipif(not reference_available(), reason="live reference only exists in the build container")
def test_oracle_against_live_reference():
    from oracle.refimport import import_reference
    M = import_reference()
    cfg = S.TINY_CONFIG
    sd = S.synth_state_dict(cfg, 555)
    ref = M.FastSVCGenerator(in_channels=cfg.in_channels, mid_channels=list(cfg.mid_channels),
                             upsampling_scales=list(cfg.upsampling_scales), out_channels=1,
                             spk_emb_size=cfg.spk_emb_size, use_spk_emb=True)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    b = S.synth_batch(cfg, 3, 31, 556)
    with torch.no_grad():
        y_ref = ref(torch.from_numpy(b.ppg), torch.from_numpy(b.sine), torch.from_numpy(b.lft),
                    torch.from_numpy(b.spk_emb)).numpy()
    y = O.forward_as_executed(S.fold_weight_norm(sd), cfg.upsampling_scales, b.ppg, b.sine, b.lft, b.spk_emb).numpy()
    assert np.abs(y - y_ref).max() <= TOL


@pytest.mark.parametrize("s", [2, 4, 5])
def test_polyphase_identity_of_stretched_conv(s):
    """The algebra the MODE_POLY kernel relies on (SURVEY note P), checked against stock torch ops:
    Conv3_d1(Stretch_s(x))[s*j + ph] = z[j] + (ph == 0) a[j] + (ph == s-1) c[j] with
    z = (W0+W1+W2) x[j], a = W0 (x[j-1] - x[j]), c = W2 (x[j+1] - x[j]) and zero padding."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(100 + s)
    C, T = 6, 17
    x = torch.randn(2, C, T, generator=g, dtype=torch.float64)
    w = torch.randn(C, C, 3, generator=g, dtype=torch.float64)
    bias = torch.randn(C, generator=g, dtype=torch.float64)
    ref = F.conv1d(torch.repeat_interleave(x, s, dim=-1), w, bias, padding=1)
    xm = F.pad(x, (1, 0))[..., :-1]
    xp = F.pad(x, (0, 1))[..., 1:]
    mm = lambda wk, v: torch.einsum("oc,bct->bot", wk, v)
    z = mm(w.sum(-1), x)
    a = mm(w[..., 0], xm - x)
    c = mm(w[..., 2], xp - x)
    out = z.repeat_interleave(s, dim=-1).clone()
    out[..., 0::s] += a
    out[..., s - 1::s] += c
    out += bias[None, :, None]
    assert float((out - ref).abs().max()) < 1e-12


@pytest.mark.parametrize("d", [1, 2, 4])
def test_winograd_f23_identity_of_dilated_conv(d):
    """The algebra of the MODE_WINO kernel, against stock torch ops: for output pairs (t, t+d) of a
    k=3 conv with dilation d, m0 = (d0-d2) g0, m1 = (d1+d2) g1, m2 = (d2-d1) g2, m3 = (d1-d3) g3 with
    g = w0 | (w0+w1+w2)/2 | (w0-w1+w2)/2 | w2 give y[t] = m0+m1+m2 and y[t+d] = m1-m2-m3; pairs tile
    the axis as t = 2d*p + r (r < d), zero padding outside [0, T)."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(200 + d)
    C, T = 5, 8 * d * 3
    x = torch.randn(2, C, T, generator=g, dtype=torch.float64)
    w = torch.randn(C, C, 3, generator=g, dtype=torch.float64)
    ref = F.conv1d(x, w, None, padding=d, dilation=d)
    xp = F.pad(x, (d, 2 * d))                       # xp[t + d] = x[t]
    t_e = torch.tensor([2 * d * p + r for p in range(T // (2 * d)) for r in range(d)])
    d0, d1, d2, d3 = (xp[..., t_e + k * d] for k in range(4))
    mm = lambda wk, v: torch.einsum("oc,bct->bot", wk, v)
    m0 = mm(w[..., 0], d0 - d2)
    m1 = mm(w.sum(-1) / 2, d1 + d2)
    m2 = mm((w[..., 0] - w[..., 1] + w[..., 2]) / 2, d2 - d1)
    m3 = mm(w[..., 2], d1 - d3)
    out = torch.zeros_like(ref)
    out[..., t_e] = m0 + m1 + m2
    out[..., t_e + d] = m1 - m2 - m3
    assert float((out - ref).abs().max()) < 1e-12
