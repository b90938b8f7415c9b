"""Multi-resolution STFT loss (harana/losses/stft_loss.py:21-180) - HIP forward and backward
(csrc/fastsvc_stftloss.hip) against the reference's own values and autograd gradients (tests/golden/stft_loss.npz,
`tests/golden/make_golden.py stft_loss`) and against the float64 oracle (oracle/stft_loss_oracle.py).

Tolerances (float32 kernels vs float64 oracle / float32 reference):
  losses            1e-5 relative
  d sc / dx         2e-4 of the gradient's largest magnitude
  d mag / dx        5e-3 of the gradient's largest magnitude - the REFERENCE's own float32 gradient sits 0.7-1.5e-3 from the
                    float64 oracle there (sign(log X - log Y) / X of bins near the floor flips with the last bit of an
                    FFT), the HIP kernels are held to the oracle at the same width
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
from svcc23_fastsvc_amd import training as TR
from oracle import stft_loss_oracle as O


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (and fail loudly without one)"
    A.load_library()
    return torch.device("cuda:0")


def _cases():
    g = load_golden("stft_loss.npz")
    recipe = dict(fft_sizes=g["recipe/fft_sizes"].tolist(), hop_sizes=g["recipe/hop_sizes"].tolist(),
                  win_lengths=g["recipe/win_lengths"].tolist(), window="hann_window")
    assert recipe == TR.RECIPE["stft_loss_params"]
    return g, S.stft_loss_cases(recipe)


def _geometry(params):
    return params["fft_sizes"], params["hop_sizes"], params["win_lengths"]


def test_oracle_matches_reference_losses_and_gradients():
    g, cases = _cases()
    for tag, x, y, params in cases:
        sc, mag = O.mr_stft_loss(x, y, *_geometry(params))
        assert abs(sc - float(g[f"{tag}/sc"])) <= 1e-6 * float(g[f"{tag}/sc"]), tag
        assert abs(mag - float(g[f"{tag}/mag"])) <= 1e-6 * float(g[f"{tag}/mag"]), tag
        gs = O.mr_stft_loss_grad(x, y, *_geometry(params), 1.0, 0.0)
        gm = O.mr_stft_loss_grad(x, y, *_geometry(params), 0.0, 1.0)
        assert np.abs(gs - g[f"{tag}/grad_sc"]).max() <= 2e-6 * np.abs(gs).max(), tag
        assert np.abs(gm - g[f"{tag}/grad_mag"]).max() <= 3e-3 * np.abs(gm).max(), tag
    # the silent prediction of the `floor` case sits at the clamp everywhere: no gradient at all
    assert not g["floor/grad_sc"][0].any() and not g["floor/grad_mag"][0].any()


def test_oracle_gradient_is_the_derivative_of_the_oracle_loss():
    rng = np.random.default_rng(5)
    x, y = rng.standard_normal((2, 700)) * 0.2, rng.standard_normal((2, 700)) * 0.2
    geo = ([256, 64], [50, 16], [200, 64])
    grad = O.mr_stft_loss_grad(x, y, *geo, 0.7, 1.3)
    d = rng.standard_normal(x.shape)
    eps = 1e-6
    f = lambda z: np.dot([0.7, 1.3], O.mr_stft_loss(z, y, *geo))
    num = (f(x + eps * d) - f(x - eps * d)) / (2 * eps)
    assert abs(num - float((grad * d).sum())) <= 1e-5 * abs(num)


def test_scratch_size_and_rejections():
    lib = A.load_library()
    i32 = lambda v: (ctypes.c_int32 * len(v))(*v)
    n = lib.fastsvc_stft_loss_scratch_bytes(2, 4000, 2, i32([1024, 64]), i32([256, 16]))
    frames = (1 + 4000 // 256) * 1024 + (1 + 4000 // 16) * 64
    assert n >= 2 * frames * 4 and n % 4 == 0
    assert lib.fastsvc_stft_loss_scratch_bytes(2, 4000, 1, i32([600]), i32([120])) == 0      # not a power of two
    assert lib.fastsvc_stft_loss_scratch_bytes(2, 4000, 1, i32([4096]), i32([1024])) == 0    # beyond the LDS transform
    assert lib.fastsvc_stft_loss_scratch_bytes(2, 1000, 1, i32([2048]), i32([512])) == 0     # reflect padding: N / 2 < T
    crit = A.MultiResolutionSTFTLoss()
    assert sorted(crit.state_dict()) == [f"stft_losses.{i}.window" for i in range(3)]
    with pytest.raises(A.FastSVCError):
        crit(torch.zeros(1, 4000), torch.zeros(1, 4000))                                     # CPU tensors: no fallback


@pytest.mark.gpu
def test_hip_loss_and_gradients_match_reference_and_oracle(dev):
    g, cases = _cases()
    for tag, x, y, params in cases:
        crit = A.MultiResolutionSTFTLoss(**params).to(dev)
        yt = torch.from_numpy(y).to(dev)
        got = {}
        for which in (0, 1):
            xt = torch.from_numpy(x).to(dev).requires_grad_(True)
            losses = crit(xt, yt)
            losses[which].backward()
            got[which] = xt.grad.cpu().numpy()
        sc, mag = float(losses[0].detach()), float(losses[1].detach())
        osc, omag = O.mr_stft_loss(x, y, *_geometry(params))
        for want_sc, want_mag in ((float(g[f"{tag}/sc"]), float(g[f"{tag}/mag"])), (osc, omag)):
            assert abs(sc - want_sc) <= 1e-5 * want_sc, (tag, sc, want_sc)
            assert abs(mag - want_mag) <= 1e-5 * want_mag, (tag, mag, want_mag)
        gs = O.mr_stft_loss_grad(x, y, *_geometry(params), 1.0, 0.0)
        gm = O.mr_stft_loss_grad(x, y, *_geometry(params), 0.0, 1.0)
        assert np.isfinite(got[0]).all() and np.isfinite(got[1]).all()
        assert np.abs(got[0] - gs).max() <= 2e-4 * np.abs(gs).max(), (tag, np.abs(got[0] - gs).max(), np.abs(gs).max())
        assert np.abs(got[1] - gm).max() <= 5e-3 * np.abs(gm).max(), (tag, np.abs(got[1] - gm).max(), np.abs(gm).max())
        assert np.abs(got[0] - g[f"{tag}/grad_sc"]).max() <= 2e-4 * np.abs(gs).max(), tag
        assert np.abs(got[1] - g[f"{tag}/grad_mag"]).max() <= 5e-3 * np.abs(gm).max(), tag
        # mean error far below the worst bin's
        assert np.abs(got[1] - gm).mean() <= 2e-4 * np.abs(gm).max(), tag
        if tag == "floor":
            assert not got[0][0].any() and not got[1][0].any()


@pytest.mark.gpu
def test_hip_loss_recipe_batch_weighted_gradient_and_reproducibility(dev):
    """BASELINE config 5's batch (32 x 16000, the recipe's six resolutions): against the torch composition on the same GPU
    (training.py, pinned to the reference by tests/test_training.py), the (B, 1, T) layout the trainer passes, arbitrary
    incoming gradients, and bit-identical results run to run (no atomics)."""
    B, T = 32, 16000
    gen = torch.Generator().manual_seed(3)
    y = (torch.randn((B, 1, T), generator=gen) * 0.2).to(dev)
    x0 = (y.cpu() * 0.9 + torch.randn((B, 1, T), generator=gen) * 0.05).to(dev)
    hip = A.MultiResolutionSTFTLoss(**TR.RECIPE["stft_loss_params"]).to(dev)
    ref = TR.MultiResolutionSTFTLoss(**TR.RECIPE["stft_loss_params"]).to(dev)
    outs = []
    for crit in (hip, ref, hip):
        x = x0.clone().requires_grad_(True)
        sc, mag = crit(x, y)
        (2.0 * sc + 0.5 * mag).backward()
        outs.append((float(sc.detach()), float(mag.detach()), x.grad.clone()))
    (sc, mag, gx), (rsc, rmag, rgx), (sc2, mag2, gx2) = outs
    assert abs(sc - rsc) <= 1e-5 * rsc and abs(mag - rmag) <= 1e-5 * rmag, (sc, rsc, mag, rmag)
    assert gx.shape == x0.shape
    err = (gx - rgx).abs()
    assert float(err.max()) <= 5e-3 * float(rgx.abs().max()) and float(err.mean()) <= 2e-4 * float(rgx.abs().max())
    assert sc == sc2 and mag == mag2 and torch.equal(gx, gx2)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(1, 1100, [2048, 8], [512, 3], [2048, 8]), (5, 777, [64, 16], [16, 5], [48, 16]),
                                  (3, 130, [256], [256], [200]), (2, 4097, [1024, 128], [1, 127], [1024, 100])], ids=str)
def test_hip_loss_edge_geometries_vs_oracle(dev, case):
    """Shortest admissible signals (fft_size / 2 < T by a few samples), a hop of one sample, hops that do not divide the
    frame, an odd number of frames (the second frame of the last pair is absent), one frame per utterance, windows shorter
    than the frame: losses and both gradients against the float64 oracle."""
    B, T, ffts, hops, wins = case
    rng = np.random.default_rng(B * 100 + T)
    y = (rng.standard_normal((B, T)) * 0.3).astype(np.float32)
    x = (y * 0.7 + rng.standard_normal((B, T)) * 0.1).astype(np.float32)
    crit = A.MultiResolutionSTFTLoss(fft_sizes=ffts, hop_sizes=hops, win_lengths=wins).to(dev)
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    sc, mag = crit(xt, torch.from_numpy(y).to(dev))
    (1.5 * sc + 0.25 * mag).backward()
    osc, omag = O.mr_stft_loss(x, y, ffts, hops, wins)
    og = O.mr_stft_loss_grad(x, y, ffts, hops, wins, 1.5, 0.25)
    assert abs(float(sc.detach()) - osc) <= 2e-5 * osc and abs(float(mag.detach()) - omag) <= 2e-5 * omag, (float(sc.detach()), osc, float(mag.detach()), omag)
    err = np.abs(xt.grad.cpu().numpy() - og)
    assert err.max() <= 5e-3 * np.abs(og).max() and err.mean() <= 2e-4 * np.abs(og).max(), (err.max(), err.mean(), np.abs(og).max())
