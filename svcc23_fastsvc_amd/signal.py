"""``SignalGenerator`` - same surface as ``harana.utils.features.SignalGenerator``
(``harana/utils/features.py:111-213``), synthesised by a HIP kernel (``csrc/fastsvc_signal.hip``).

It is the step right before the generator forward inside ``FastSVCGenerator.inference``
(``fastsvc.py:381``) and the last host-side PyTorch op on the decode path.
GPU tensors only; no CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import List, Sequence

import torch

from .engine import FastSVCError, load_library

_TYPE_CODE = {"noise": 0, "sine": 1, "uv": 2}


class SignalGenerator:
    """Input signal generator (NSF-style sine excitation).

    Args (reference defaults, features.py:114-121): sample_rate=16000, hop_size=640,
    sine_amp=0.1, noise_amp=0.003, signal_types=["sine", "noise"]; plus ``seed`` for the
    counter-based noise generator (the reference draws ``torch.randn``).
    """

    def __init__(self, sample_rate: int = 16000, hop_size: int = 640, sine_amp: float = 0.1,
                 noise_amp: float = 0.003, signal_types: Sequence[str] = ("sine", "noise"), seed: int = 0):
        for t in signal_types:
            if t not in _TYPE_CODE:
                raise ValueError(f"{t} is not a supported signal type (noise, sine, uv)")
        self.sample_rate = sample_rate
        self.hop_size = hop_size
        self.sine_amp = sine_amp
        self.noise_amp = noise_amp
        self.signal_types = list(signal_types)
        self.seed = int(seed)
        self._calls = 0
        self._lib = load_library()

    @torch.no_grad()
    def __call__(self, f0: torch.Tensor) -> torch.Tensor:
        """f0 (B, 1, F) in Hz (0 = unvoiced) -> (B, len(signal_types), F * hop_size)."""
        if not isinstance(f0, torch.Tensor) or not f0.is_cuda:
            raise FastSVCError("SignalGenerator (HIP) needs a GPU tensor; there is no CPU fallback")
        if f0.dim() != 3 or f0.shape[1] != 1:
            raise ValueError(f"f0 must be (B, 1, F), got {tuple(f0.shape)}")
        f0 = f0.to(torch.float32).contiguous()
        B, _, F = f0.shape
        n = len(self.signal_types)
        out = torch.empty((B, n, F * self.hop_size), dtype=torch.float32, device=f0.device)
        scratch = torch.empty(self._lib.fastsvc_signal_scratch_bytes(B, F), dtype=torch.uint8, device=f0.device)
        types = (ctypes.c_int32 * n)(*[_TYPE_CODE[t] for t in self.signal_types])
        self._calls += 1
        seed = (self.seed * 0x9E3779B97F4A7C15 + self._calls) & 0xFFFFFFFFFFFFFFFF
        with torch.cuda.device(f0.device):
            rc = self._lib.fastsvc_signal_generate(
                ctypes.c_void_p(f0.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                ctypes.c_void_p(scratch.data_ptr()), B, F, int(self.hop_size),
                ctypes.c_float(float(self.sample_rate)), ctypes.c_float(float(self.sine_amp)),
                ctypes.c_float(float(self.noise_amp)), types, n, ctypes.c_uint64(seed),
                ctypes.c_void_p(torch.cuda.current_stream(f0.device).cuda_stream))
        if rc != 0:
            raise FastSVCError(f"fastsvc_signal_generate failed ({rc})")
        return out
