"""CPU oracle of the loudness feature.  TEST INFRASTRUCTURE - NOT A PRODUCT PATH (see oracle/fastsvc_oracle.py).

Restates ``loudness_extract(audio, sampling_rate, hop_length)`` (``harana/bin/preprocess_fastsvc.py:60-75``).
The arithmetic lives in a third-party dependency that is ABSENT from this image and from ``/root/reference``:
**librosa==0.8.1** (pinned in ``setup.py:30``).  **Parity unpinned**: the reference function cannot be run here,
so this file restates librosa 0.8.1's published algorithm in numpy (float64) from its call site, function by
function, and is anchored only on (a) the IEC 61672 A-weighting table values librosa's formula reproduces and
(b) numpy's FFT:

* ``librosa.stft(y, hop_length=hop)``: n_fft = 2048, win_length = n_fft, ``scipy.signal.get_window("hann", 2048,
  fftbins=True)`` (periodic), ``center=True`` -> ``np.pad(y, 1024, mode="reflect")``, frames at multiples of hop:
  ``1 + len(y) // hop`` columns, ``rfft`` of each windowed frame.
* ``librosa.fft_frequencies(sr)``: ``linspace(0, sr / 2, 1025)``.
* ``librosa.perceptual_weighting(S, f)`` = ``A_weighting(f)[:, None] + power_to_db(S)`` with ``power_to_db(S,
  ref=1.0, amin=1e-10, top_db=80.0)`` = ``10 log10(max(amin, S))`` floored at (its maximum over the WHOLE array - 80).
* ``librosa.A_weighting(f, min_db=-80)``: ``2.0 + 20 (log10(c0) + 2 log10(f^2) - log10(f^2 + c0) - log10(f^2 + c1)
  - 0.5 log10(f^2 + c2) - 0.5 log10(f^2 + c3))``, ``c = [12200, 20.6, 107.7, 737.9]^2``, clipped below at -80.
* ``librosa.db_to_amplitude(x)`` = ``10^(x / 20)``; then ``log(mean over bins + 1e-5)`` and ``Stretch2d(hop, 1)``
  (nearest repeat, ``harana/layers/upsample.py:21-50``).
"""
import numpy as np

N_FFT = 2048


def a_weighting(freqs: np.ndarray, min_db: float = -80.0) -> np.ndarray:
    f_sq = np.asarray(freqs, dtype=np.float64) ** 2
    const = np.array([12200.0, 20.6, 107.7, 737.9]) ** 2
    with np.errstate(divide="ignore"):
        w = 2.0 + 20.0 * (np.log10(const[0]) + 2 * np.log10(f_sq) - np.log10(f_sq + const[0]) - np.log10(f_sq + const[1])
                          - 0.5 * np.log10(f_sq + const[2]) - 0.5 * np.log10(f_sq + const[3]))
    return np.maximum(min_db, w)


def stft_power(audio: np.ndarray, hop: int) -> np.ndarray:
    y = np.asarray(audio, dtype=np.float64)
    ypad = np.pad(y, N_FFT // 2, mode="reflect")
    k = np.arange(N_FFT)
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * k / N_FFT)
    frames = 1 + len(y) // hop
    cols = np.stack([ypad[f * hop: f * hop + N_FFT] * window for f in range(frames)], axis=1)   # (n_fft, frames)
    return np.abs(np.fft.rfft(cols, axis=0)) ** 2                                                # (1025, frames)


def loudness_extract(audio: np.ndarray, sampling_rate: int, hop_length: int) -> np.ndarray:
    P = stft_power(audio, hop_length)
    db = 10.0 * np.log10(np.maximum(1e-10, P))
    db = np.maximum(db, db.max() - 80.0)
    db = db + a_weighting(np.linspace(0.0, sampling_rate / 2.0, N_FFT // 2 + 1))[:, None]
    loud = np.log(np.mean(10.0 ** (db / 20.0), axis=0) + 1e-5)
    return np.repeat(loud, hop_length).astype(np.float32)
