"""Loudness feature (harana/bin/preprocess_fastsvc.py:60-75).  The reference function needs
librosa 0.8.1, which is absent here: the oracle restates librosa's algorithm (oracle/loudness_oracle.py, "parity
unpinned") and is anchored on the IEC 61672 A-weighting table, numpy's FFT and a direct DFT; the HIP kernels are
compared with the oracle."""
import numpy as np
import pytest
import torch

from oracle import loudness_oracle as LO


def _audio(seed, T, sr=24000, amp=0.3):
    rng = np.random.default_rng(seed)
    t = np.arange(T) / sr
    f0 = 110.0 * (1.0 + 0.3 * np.sin(2 * np.pi * 0.7 * t))
    y = sum(amp / (k + 1) * np.sin(2 * np.pi * np.cumsum(f0 * (k + 1)) / sr) for k in range(6))
    env = 0.5 * (1.0 + np.sin(2 * np.pi * 1.3 * t)) ** 2
    return (y * env + 0.01 * amp * rng.standard_normal(T)).astype(np.float32)


def test_a_weighting_reproduces_the_iec_61672_table():
    got = LO.a_weighting(np.array([20.0, 100.0, 1000.0, 2000.0, 10000.0, 0.0]))
    want = np.array([-50.5, -19.1, 0.0, 1.2, -2.5, -80.0])          # IEC 61672-1 table 3 (A), and librosa's floor at f = 0
    assert np.abs(got - want).max() <= 0.11


def test_oracle_stft_against_a_direct_dft_and_geometry():
    y = _audio(1, 5000)
    hop = 64
    P = LO.stft_power(y, hop)
    assert P.shape == (1025, 1 + 5000 // hop)
    ypad = np.pad(y.astype(np.float64), 1024, mode="reflect")
    k = np.arange(2048)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * k / 2048)
    for f, b in ((0, 3), (17, 40), (78, 1024)):
        direct = np.abs(np.sum(ypad[f * hop: f * hop + 2048] * w * np.exp(-2j * np.pi * b * k / 2048))) ** 2
        assert abs(P[b, f] - direct) <= 1e-9 * max(1.0, direct)
    out = LO.loudness_extract(y, 24000, hop)
    assert out.shape == ((1 + 5000 // hop) * hop,) and np.all(out[:hop] == out[0]) and np.isfinite(out).all()
    # a pure tone: 40 dB more amplitude moves the feature by log(100) wherever the tone dominates the 1e-5 floor
    tone = np.sin(2 * np.pi * 1000.0 * np.arange(8000) / 24000.0).astype(np.float32)
    a, c = LO.loudness_extract(0.5 * tone, 24000, 160), LO.loudness_extract(0.005 * tone, 24000, 160)
    assert abs((a - c)[4000] - np.log(100.0)) <= 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("hop", [64, 160])
def test_loudness_kernels_match_the_oracle(hop):
    import svcc23_fastsvc_amd as A
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    T = 36000 + 17                                        # 1.5 s, not a multiple of the hop
    rows = [_audio(2, T), _audio(3, T, amp=1e-3), np.zeros(T, np.float32)]
    rows[2][100] = 1e-4                                   # (almost) silence: everything sits on the floors
    got = A.loudness_extract(torch.from_numpy(np.stack(rows)).to(dev), 24000, hop).cpu().numpy()
    assert got.shape == (3, (1 + T // hop) * hop)
    for i, y in enumerate(rows):
        want = LO.loudness_extract(y, 24000, hop)
        assert np.abs(got[i] - want).max() <= 2e-3, i
    one = A.loudness_extract(torch.from_numpy(rows[0]).to(dev), 24000, hop).cpu().numpy()
    assert np.array_equal(one, got[0])
    with pytest.raises(A.FastSVCError):
        A.loudness_extract(torch.from_numpy(rows[0]), 24000, hop)
