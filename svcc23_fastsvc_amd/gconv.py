"""Grouped strided convolution + LeakyReLU of the recipe's discriminator with hand-written HIP kernels forward and
backward (``csrc/fastsvc_gconv.hip``; SURVEY.md §8 f2, BASELINE config 5).

The reference's ``MelGANDiscriminator`` (``harana/models/fastsvc.py:386-520``) downsamples with
``Conv1d(c, min(4c, 512), kernel_size=41, stride=4, padding=20, groups=c // 4)`` + ``LeakyReLU(0.2)``; PyTorch-ROCm runs each as
per-sample im2col + GEMMs + layout transposes (forward, backward data and backward weight: half the kernel time of a
training step).  ``GroupedConv1d`` is an ``nn.Conv1d`` (same parameters, same state-dict keys, weight-norm hooks work on it)
whose call takes the activation's slope and runs ONE launch forward and three backward; ``ConvAct`` is the
``nn.Sequential(conv, LeakyReLU)`` of the reference's module tree that hands the slope over.  Unsupported shapes, dtypes or
CPU tensors run the stock operators - the module then IS the reference's pair of modules."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from .conv_grad import _guard, _lib, _ptr
from .engine import FastSVCError

USE_HIP = True            # module switch (A/B timing, tests of the stock route)
# how often a call took which route since import (bench.py reports them: a shape or table change that moves the step back onto
# the stock operators must not go unnoticed)
ROUTES = {"hip": 0, "stock": 0}


def supported(conv: nn.Conv1d) -> bool:
    return (conv.padding_mode == "zeros" and conv.dilation == (1,) and
            bool(_lib().fastsvc_gconv1d_supported(conv.in_channels, conv.out_channels, conv.groups, conv.kernel_size[0],
                                                  conv.stride[0], conv.padding[0] if isinstance(conv.padding, tuple) else -1)))


def _stream(t: torch.Tensor) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class _GroupedConvActFn(torch.autograd.Function):
    """y = leaky_relu(conv1d(x, w, b, stride, padding, groups), slope); float32, GPU."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups: int, stride: int, padding: int, slope: float):
        x = x.detach().to(torch.float32).contiguous()
        w = weight.detach().to(torch.float32).contiguous()
        b = None if bias is None else bias.detach().to(torch.float32).contiguous()
        B, Cin, T = x.shape
        Cout, _, K = w.shape
        Tout = (T + 2 * padding - K) // stride + 1
        y = torch.empty((B, Cout, Tout), dtype=torch.float32, device=x.device)
        with _guard(x):
            rc = _lib().fastsvc_gconv1d_forward(_ptr(x), _ptr(w), _ptr(b), _ptr(y), B, Cin, Cout, groups, T, K, stride, padding,
                                                ctypes.c_float(slope), _stream(x))
        if rc != 0:
            raise FastSVCError(f"fastsvc_gconv1d_forward failed ({rc}): x {tuple(x.shape)} w {tuple(w.shape)} groups {groups}")
        ctx.save_for_backward(x, w, y)
        ctx.conf = (groups, stride, padding, slope, bias is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        groups, stride, padding, slope, has_bias = ctx.conf
        dy = dy.detach().to(torch.float32).contiguous()
        B, Cin, T = x.shape
        Cout, _, K = w.shape
        lib = _lib()
        yact = y if slope != 1.0 else None
        dx = dw = db = None
        with _guard(x):
            st = _stream(x)
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                rc = lib.fastsvc_gconv1d_backward_data(_ptr(dy), _ptr(yact), _ptr(w), _ptr(dx), B, Cin, Cout, groups, T, K, stride,
                                                       padding, ctypes.c_float(slope), st)
                if rc != 0:
                    raise FastSVCError(f"fastsvc_gconv1d_backward_data failed ({rc})")
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                dw = torch.empty_like(w)
                db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if has_bias else None
                scratch = torch.empty(int(lib.fastsvc_gconv1d_backward_weight_scratch_bytes(B, Cout, T)), dtype=torch.uint8,
                                      device=x.device)
                rc = lib.fastsvc_gconv1d_backward_weight(_ptr(x), _ptr(dy), _ptr(yact), _ptr(dw), _ptr(db), _ptr(scratch), B, Cin,
                                                         Cout, groups, T, K, stride, padding, ctypes.c_float(slope), st)
                if rc != 0:
                    raise FastSVCError(f"fastsvc_gconv1d_backward_weight failed ({rc})")
        return dx, dw, db, None, None, None, None


class GroupedConv1d(nn.Conv1d):
    """``nn.Conv1d`` whose call may fuse the LeakyReLU behind it: ``conv(x, act_slope=0.2)``."""

    def forward(self, input: torch.Tensor, act_slope: Optional[float] = None) -> torch.Tensor:  # noqa: A002
        # (the fused backward takes the activation's derivative from the sign of the ACTIVATED output: right for slopes in
        # (0, 1] only - a negative slope flips the sign, slope 0 loses it - every other slope runs the stock pair)
        slope_ok = act_slope is None or 0.0 < float(act_slope) <= 1.0
        if (USE_HIP and slope_ok and input.is_cuda and input.dtype == torch.float32 and self.weight.dtype == torch.float32 and
                not torch.is_autocast_enabled() and input.dim() == 3 and supported(self)):
            ROUTES["hip"] += 1
            return _GroupedConvActFn.apply(input, self.weight, self.bias, self.groups, self.stride[0], self.padding[0],
                                           1.0 if act_slope is None else float(act_slope))
        ROUTES["stock"] += 1
        y = super().forward(input)
        return y if act_slope is None else F.leaky_relu(y, act_slope)


class ConvAct(nn.Sequential):
    """``nn.Sequential(conv, LeakyReLU)`` (children "0" and "1": the reference's state-dict keys) run as one node when the
    convolution is a `GroupedConv1d`."""

    def forward(self, x):
        conv, act = self[0], self[1]
        if isinstance(conv, GroupedConv1d) and isinstance(act, nn.LeakyReLU):
            return conv(x, act_slope=act.negative_slope)
        return act(conv(x))
