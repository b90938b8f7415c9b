"""``MultiResolutionSTFTLoss`` - the auxiliary loss of the reference's training step
(``harana/losses/stft_loss.py:131-180``, called at ``harana/bin/train_fastsvc.py:163-170``), forward
AND backward as HIP kernels (``csrc/fastsvc_stftloss.hip``: LDS radix-2 FFTs, two frames per complex transform, the
sums folded in a fixed order, the adjoint transform and a gather for the waveform gradient - no atomics).

Same constructor surface, buffer names (``stft_losses.<i>.window``) and return value as the reference module:
``sc_loss, mag_loss = criterion(y_hat, y)``.  The gradient flows to the FIRST argument only (the reference's target
comes from the data loader and never requires one).  GPU tensors only: there is no CPU fallback."""
from __future__ import annotations

import ctypes
from typing import Sequence, Tuple

import torch
from torch import nn

from torch.autograd.function import once_differentiable

from .engine import FastSVCError, load_library


class _Resolutions:
    """Host-side description shared by the forward and the backward of one call."""

    def __init__(self, resolutions: Sequence[Tuple[int, int, int]], windows: Sequence[torch.Tensor]):
        n = len(resolutions)
        self.n = n
        self.fft = (ctypes.c_int32 * n)(*[r[0] for r in resolutions])
        self.hop = (ctypes.c_int32 * n)(*[r[1] for r in resolutions])
        self.win = (ctypes.c_int32 * n)(*[r[2] for r in resolutions])
        self.windows = list(windows)                      # keeps the device buffers alive
        self.ptrs = (ctypes.c_void_p * n)(*[w.data_ptr() for w in self.windows])


class _STFTLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, y: torch.Tensor, res: _Resolutions):
        lib = load_library()
        B, T = x.shape
        nbytes = int(lib.fastsvc_stft_loss_scratch_bytes(B, T, res.n, res.fft, res.hop))
        if nbytes == 0:
            raise ValueError(f"unsupported STFT loss geometry: B={B}, T={T}, fft sizes {list(res.fft)} (powers of two in "
                             f"[8, 2048], fft_size / 2 < T), hops {list(res.hop)}")
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        loss = torch.empty(2, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.fastsvc_stft_loss_forward(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), B, T, res.n,
                                               res.fft, res.hop, res.win, res.ptrs, ctypes.c_void_p(loss.data_ptr()),
                                               ctypes.c_void_p(scratch.data_ptr()),
                                               ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise FastSVCError(f"fastsvc_stft_loss_forward failed ({rc})")
        ctx.save_for_backward(x, y)
        ctx.res, ctx.scratch = res, scratch
        return loss[0], loss[1]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_sc, g_mag):
        x, y = ctx.saved_tensors
        res, scratch = ctx.res, ctx.scratch
        lib = load_library()
        B, T = x.shape
        zero = torch.zeros((), dtype=torch.float32, device=x.device)
        g = torch.stack([(g_sc if g_sc is not None else zero).to(torch.float32).reshape(()),
                         (g_mag if g_mag is not None else zero).to(torch.float32).reshape(())]).contiguous()
        grad_x = torch.empty_like(x)
        with torch.cuda.device(x.device):
            rc = lib.fastsvc_stft_loss_backward(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), B, T, res.n,
                                                res.fft, res.hop, res.win, res.ptrs, ctypes.c_void_p(g.data_ptr()),
                                                ctypes.c_void_p(grad_x.data_ptr()), ctypes.c_void_p(scratch.data_ptr()),
                                                ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise FastSVCError(f"fastsvc_stft_loss_backward failed ({rc})")
        return grad_x, None, None


class MultiResolutionSTFTLoss(nn.Module):
    """``(sc, mag) = loss(y_hat, y)``: spectral convergence ``||Y - X||_F / ||Y||_F`` and log-magnitude L1, each
    averaged over the resolutions (defaults: the reference class's, ``stft_loss.py:134-140``)."""

    def __init__(self, fft_sizes=(1024, 2048, 512), hop_sizes=(120, 240, 50), win_lengths=(600, 1200, 240),
                 window: str = "hann_window"):
        super().__init__()
        if not (len(fft_sizes) == len(hop_sizes) == len(win_lengths)):
            raise ValueError("fft_sizes, hop_sizes and win_lengths must have the same length")
        self.resolutions = [(int(f), int(h), int(w)) for f, h, w in zip(fft_sizes, hop_sizes, win_lengths)]
        self.stft_losses = nn.ModuleList()
        for _, _, w in self.resolutions:
            holder = nn.Module()
            holder.register_buffer("window", getattr(torch, window)(w))
            self.stft_losses.append(holder)

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if not (isinstance(x, torch.Tensor) and x.is_cuda and y.is_cuda):
            raise FastSVCError("MultiResolutionSTFTLoss (HIP) needs GPU tensors; there is no CPU fallback")
        if x.shape != y.shape:
            raise ValueError(f"x {tuple(x.shape)} and y {tuple(y.shape)} must have the same shape")
        if y.requires_grad:
            raise FastSVCError("the gradient with respect to the target signal is not implemented (the reference never asks)")
        if x.dim() == 3:                                  # (B, C, T) -> (B x C, T)   (stft_loss.py:165-167)
            x = x.reshape(-1, x.size(2))
            y = y.reshape(-1, y.size(2))
        if x.dim() != 2:
            raise ValueError(f"signals must be (B, T) or (B, C, T), got {tuple(x.shape)}")
        windows = [h.window.to(device=x.device, dtype=torch.float32).contiguous() for h in self.stft_losses]
        res = _Resolutions(self.resolutions, windows)
        return _STFTLossFn.apply(x.to(torch.float32).contiguous(), y.detach().to(torch.float32).contiguous(), res)
