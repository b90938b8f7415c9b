"""``loudness_extract`` - the A-weighted log-loudness feature of the reference's preprocessing
(``harana/bin/preprocess_fastsvc.py:60-75``, the producer of the generator's ``l`` input),
computed by HIP kernels (``csrc/fastsvc_loudness.hip``: 2048-point FFT per frame in LDS, A-weighting, 80 dB
floor below the utterance maximum, mean over bins, log, nearest stretch by the hop).  GPU tensors only."""
from __future__ import annotations

import ctypes

import torch

from .engine import FastSVCError, load_library


@torch.no_grad()
def loudness_extract(audio: torch.Tensor, sampling_rate: int, hop_length: int) -> torch.Tensor:
    """audio (T,) or (B, T) float on the GPU -> (frames * hop,) or (B, frames * hop), frames = 1 + T // hop
    (the reference's length; its dataset code trims it to the other features, ``preprocess_fastsvc.py:239-260``).
    Each row of a batch is one utterance (the 80 dB floor is relative to that row's own maximum)."""
    if not isinstance(audio, torch.Tensor) or not audio.is_cuda:
        raise FastSVCError("loudness_extract (HIP) needs a GPU tensor; there is no CPU fallback")
    single = audio.dim() == 1
    a = (audio[None] if single else audio).to(torch.float32).contiguous()
    if a.dim() != 2 or a.shape[1] < 2:
        raise ValueError(f"audio must be (T,) or (B, T) with T >= 2, got {tuple(audio.shape)}")
    lib = load_library()
    B, T = a.shape
    frames = int(lib.fastsvc_loudness_frames(T, int(hop_length)))
    out = torch.empty((B, frames * hop_length), dtype=torch.float32, device=a.device)
    scratch = torch.empty(int(lib.fastsvc_loudness_scratch_bytes(B, T, int(hop_length))), dtype=torch.uint8, device=a.device)
    with torch.cuda.device(a.device):
        rc = lib.fastsvc_loudness_extract(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                          ctypes.c_void_p(scratch.data_ptr()), B, T, int(hop_length),
                                          ctypes.c_float(float(sampling_rate)),
                                          ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream))
    if rc != 0:
        raise FastSVCError(f"fastsvc_loudness_extract failed ({rc})")
    return out[0] if single else out
