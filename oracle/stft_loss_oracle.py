"""CPU restatement (numpy, float64) of the reference's multi-resolution STFT loss and of the gradient autograd derives
from it for the predicted signal.

TEST INFRASTRUCTURE ONLY: imported by ``tests/`` as the checker of ``csrc/fastsvc_stftloss.hip``; nothing under
``svcc23_fastsvc_amd/`` imports this file.

Follows ``/root/reference/harana/losses/stft_loss.py``:
  * ``stft()`` :21-51 - ``torch.stft(x, fft_size, hop, win_length, window, center=True, onesided=True)``: reflect padding
    by fft_size // 2, frames = 1 + T // hop, the window centred in the frame when shorter; magnitude
    ``sqrt(clamp(re^2 + im^2, min=1e-7))``;
  * ``SpectralConvergenceLoss`` :54-74 - ``||Y - X||_F / ||Y||_F`` over the whole batch;
  * ``LogSTFTMagnitudeLoss`` :77-97 - ``mean |log Y - log X|``;
  * ``MultiResolutionSTFTLoss.forward`` :157-180 - both averaged over the resolutions.
PINNED: ``tests/golden/stft_loss.npz`` holds the reference's own sc / mag / autograd gradients for three cases
(``tests/golden/make_golden.py stft_loss``); ``tests/test_stft_loss.py`` checks this file against them.
"""
from typing import Sequence, Tuple

import numpy as np

FLOOR = 1e-7


def hann_window(n: int) -> np.ndarray:
    """torch.hann_window(n) (periodic)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def _frames(x: np.ndarray, n_fft: int, hop: int, window: np.ndarray):
    B, T = x.shape
    pad = n_fft // 2
    if pad >= T:
        raise ValueError("reflect padding needs fft_size // 2 < T")
    xp = np.pad(x.astype(np.float64), ((0, 0), (pad, pad)), mode="reflect")
    nfr = 1 + T // hop
    w = np.zeros(n_fft)
    off = (n_fft - len(window)) // 2
    w[off:off + len(window)] = window
    idx = np.arange(nfr)[:, None] * hop + np.arange(n_fft)[None, :]
    return xp[:, idx] * w, w, idx, pad


def stft_magnitude(x: np.ndarray, n_fft: int, hop: int, window: np.ndarray) -> np.ndarray:
    fr, _, _, _ = _frames(x, n_fft, hop, window)
    spec = np.fft.rfft(fr, axis=-1)
    return np.sqrt(np.maximum(spec.real ** 2 + spec.imag ** 2, FLOOR))


def mr_stft_loss(x: np.ndarray, y: np.ndarray, fft_sizes: Sequence[int], hop_sizes: Sequence[int],
                 win_lengths: Sequence[int]) -> Tuple[float, float]:
    sc = mag = 0.0
    for n_fft, hop, wl in zip(fft_sizes, hop_sizes, win_lengths):
        w = hann_window(int(wl))
        xm, ym = stft_magnitude(x, int(n_fft), int(hop), w), stft_magnitude(y, int(n_fft), int(hop), w)
        sc += np.linalg.norm(ym - xm) / np.linalg.norm(ym)
        mag += np.mean(np.abs(np.log(ym) - np.log(xm)))
    return sc / len(fft_sizes), mag / len(fft_sizes)


def mr_stft_loss_grad(x: np.ndarray, y: np.ndarray, fft_sizes: Sequence[int], hop_sizes: Sequence[int],
                      win_lengths: Sequence[int], g_sc: float = 1.0, g_mag: float = 1.0) -> np.ndarray:
    """d (g_sc * sc + g_mag * mag) / dx, by the chain rule autograd applies: magnitude -> clamp -> |.|^2 -> rfft ->
    window -> frame gather -> reflect padding."""
    B, T = x.shape
    R = len(fft_sizes)
    grad = np.zeros((B, T))
    for n_fft, hop, wl in zip(fft_sizes, hop_sizes, win_lengths):
        n_fft, hop = int(n_fft), int(hop)
        win = hann_window(int(wl))
        fr, w, idx, pad = _frames(x, n_fft, hop, win)
        spec = np.fft.rfft(fr, axis=-1)
        power = spec.real ** 2 + spec.imag ** 2
        xm = np.sqrt(np.maximum(power, FLOOR))
        ym = stft_magnitude(y, n_fft, hop, win)
        n1, n2 = np.linalg.norm(ym - xm), np.linalg.norm(ym)
        gxm = g_mag / R * np.sign(np.log(xm) - np.log(ym)) / (xm * xm.size)
        if n1 > 0:
            gxm = gxm + g_sc / R * (xm - ym) / (n1 * n2)
        G = np.where(power >= FLOOR, gxm / xm, 0.0) * spec                    # dL/dRe + i dL/dIm
        # d/d frame[n] = Re sum_{k <= N/2} G[k] exp(+2 pi i k n / N): a real inverse DFT of the Hermitian extension
        H = 0.5 * G
        H[..., 0] = G[..., 0].real
        H[..., -1] = G[..., -1].real
        ga = np.fft.irfft(H, n=n_fft, axis=-1) * n_fft * w
        gp = np.zeros((B, T + 2 * pad))
        for b in range(B):
            np.add.at(gp[b], idx, ga[b])
        g = gp[:, pad:pad + T].copy()
        g[:, 1:pad + 1] += gp[:, pad - 1::-1][:, :pad]                         # left reflection: q -> t = pad - q
        g[:, T - 1 - pad:T - 1] += _right_fold(gp, pad, T)                     # right reflection: q -> t = 2 (T - 1) - (q - pad)
        grad += g
    return grad


def _right_fold(gp: np.ndarray, pad: int, T: int) -> np.ndarray:
    """Contributions of the padded positions q = pad + T + j (j < pad), which read sample t = T - 2 - j, laid out for
    the slice [T - 1 - pad, T - 1)."""
    right = gp[:, pad + T:pad + T + pad]              # j = 0 .. pad-1  ->  t = T-2 .. T-1-pad
    return right[:, ::-1]
