"""``conv1d`` with hand-written HIP backward kernels - the convolution node of the generator's training graph
(SURVEY.md §8 f2).  The reference trainer differentiates the generator with autograd
(``harana/bin/train_fastsvc.py:157-205``); every convolution of the generator is a stride-1 "same" convolution with
k = 1 or 3 (``harana/layers/residual_block.py:27-48``, ``harana/models/fastsvc.py:34-232``).  Here that node is

* forward        ``fastsvc_conv1d_forward``            (float32 matrix-core kernel, ``csrc/fastsvc_convgrad.hip``)
* backward data  the same kernel, weight read transposed with flipped taps
* backward weight / bias   ``fastsvc_conv1d_backward_weight``

instead of MIOpen's ``convolution_backward``.  float32 in, float32 out (master weights and gradients are float32);
GPU tensors only - there is no CPU route."""
from __future__ import annotations

import contextlib
import ctypes
from typing import Optional

import torch
from torch.autograd.function import once_differentiable

from .engine import FastSVCError, load_library


def _ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = load_library()
    return _LIB


def _guard(t: torch.Tensor):
    """The kernels launch on the CURRENT device's stream: switch only when the tensor lives elsewhere (the context manager
    costs ~10 us per call; the train step runs ~200 of these nodes)."""
    return contextlib.nullcontext() if t.device.index == torch.cuda.current_device() else torch.cuda.device(t.device)


def _launch_forward(x, w, bias, Cout: int, dilation: int, transposed: bool) -> torch.Tensor:
    lib = _lib()
    B, Cin, T = x.shape
    K = w.shape[-1]
    y = torch.empty((B, Cout, T), dtype=torch.float32, device=x.device)
    with _guard(x):
        rc = lib.fastsvc_conv1d_forward(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, Cin, Cout, T, K, int(dilation), int(transposed),
                                        ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc != 0:
        raise FastSVCError(f"fastsvc_conv1d_forward failed ({rc}): B={B} Cin={Cin} Cout={Cout} T={T} K={K} dilation={dilation}")
    return y


class _Conv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, dilation: int):
        x = x.detach().to(torch.float32).contiguous()
        w = weight.detach().to(torch.float32).contiguous()
        b = None if bias is None else bias.detach().to(torch.float32).contiguous()
        ctx.save_for_backward(x, w)
        ctx.dilation, ctx.has_bias = int(dilation), bias is not None
        return _launch_forward(x, w, b, w.shape[0], dilation, False)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.to(torch.float32).contiguous()
        B, Cin, T = x.shape
        Cout, _, K = w.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _launch_forward(dy, w, None, Cin, ctx.dilation, True)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            lib = _lib()
            dw = torch.empty_like(w)
            db = torch.empty(Cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            scratch = torch.empty(int(lib.fastsvc_conv1d_backward_weight_scratch_bytes(B, Cin, Cout, T, K)), dtype=torch.uint8,
                                  device=x.device)
            with _guard(x):
                rc = lib.fastsvc_conv1d_backward_weight(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(scratch), B, Cin, Cout, T, K, ctx.dilation,
                                                        ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
            if rc != 0:
                raise FastSVCError(f"fastsvc_conv1d_backward_weight failed ({rc})")
        return dx, dw, db, None


# calls of conv1d() by route since import (bench.py reports them next to the train step)
ROUTES = {"hip": 0, "stock": 0}


def conv1d_supported(x: torch.Tensor, weight: torch.Tensor, dilation: int = 1) -> bool:
    """What csrc/fastsvc_convgrad.hip takes: k = 1 or 3, halo (k // 2) * dilation <= 27, every tensor below 2 GiB."""
    k = int(weight.shape[-1])
    big = max(x.numel(), x.shape[0] * weight.shape[0] * x.shape[2]) * 4 >= (1 << 31)
    return k in (1, 3) and (k // 2) * int(dilation) <= 27 and not big


def conv1d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, dilation: int = 1) -> torch.Tensor:
    """``F.conv1d(x, weight, bias, padding=(k // 2) * dilation, dilation=dilation)`` for k = 1 or 3, differentiable with
    respect to x, weight and bias through the HIP kernels."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and weight.is_cuda):
        raise FastSVCError("conv1d (HIP) needs GPU tensors; there is no CPU fallback")
    if x.dim() != 3 or weight.dim() != 3 or weight.shape[1] != x.shape[1]:
        raise ValueError(f"conv1d: x {tuple(x.shape)} / weight {tuple(weight.shape)} do not fit")
    if not conv1d_supported(x, weight, dilation):
        ROUTES["stock"] += 1
        # shapes the kernels decline (FASTSVC_E_UNSUPPORTED: other tap counts, halos past 27, tensors of 2 GiB and more): the
        # stock operator on the same GPU tensors - still no CPU route
        import torch.nn.functional as F
        return F.conv1d(x, weight, bias, padding=(weight.shape[-1] // 2) * int(dilation), dilation=int(dilation))
    ROUTES["hip"] += 1
    return _Conv1dFn.apply(x, weight, bias, int(dilation))


class _FilmNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift, bias, eps: float, slope: float):
        x, scale, shift = (t.detach().to(torch.float32).contiguous() for t in (x, scale, shift))
        B, C, T = x.shape
        bias = bias.detach().to(torch.float32).reshape(B * C).contiguous()
        lib = _lib()
        out = torch.empty_like(x)
        mean = torch.empty(B * C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        with _guard(x):
            rc = lib.fastsvc_film_norm_forward(_ptr(x), _ptr(scale), _ptr(shift), _ptr(bias), _ptr(out), _ptr(mean), _ptr(rstd), B * C, T,
                                               ctypes.c_float(eps), ctypes.c_float(slope),
                                               ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise FastSVCError(f"fastsvc_film_norm_forward failed ({rc})")
        ctx.save_for_backward(x, scale, shift, bias, mean, rstd)
        ctx.slope = float(slope)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x, scale, shift, bias, mean, rstd = ctx.saved_tensors
        dout = dout.to(torch.float32).contiguous()
        B, C, T = x.shape
        lib = _lib()
        dx, dsc, dsh = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        dbias = torch.empty(B * C, dtype=torch.float32, device=x.device)
        with _guard(x):
            rc = lib.fastsvc_film_norm_backward(_ptr(dout), _ptr(x), _ptr(scale), _ptr(shift), _ptr(bias), _ptr(mean), _ptr(rstd),
                                                _ptr(dx), _ptr(dsc), _ptr(dsh), _ptr(dbias), B * C, T, ctypes.c_float(ctx.slope),
                                                ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise FastSVCError(f"fastsvc_film_norm_backward failed ({rc})")
        return dx, dsc, dsh, dbias.view(B, C, 1), None, None


def film_norm_lrelu(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5,
                    slope: float = 0.2) -> torch.Tensor:
    """``leaky_relu(instance_norm(scale * x + shift, eps) + bias, slope)`` - `_feature_affine` with a speaker embedding
    (``fastsvc.py:115-139``) and the LeakyReLU behind it as one graph node with HIP forward and backward kernels
    (``csrc/fastsvc_filmnorm.hip``).  x, scale, shift (B, C, T); bias (B, C, 1)."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        raise FastSVCError("film_norm_lrelu (HIP) needs GPU tensors; there is no CPU fallback")
    if x.dim() != 3 or scale.shape != x.shape or shift.shape != x.shape or bias.numel() != x.shape[0] * x.shape[1]:
        raise ValueError(f"film_norm_lrelu: x {tuple(x.shape)}, scale {tuple(scale.shape)}, shift {tuple(shift.shape)}, "
                         f"bias {tuple(bias.shape)} do not fit")
    return _FilmNormFn.apply(x, scale, shift, bias, float(eps), float(slope))


_WN_MAX = 56


def _ptr_array(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


class _WeightNormFn(torch.autograd.Function):
    """w_i = g_i * v_i / ||v_i|| for a list of layers: ONE launch forward, ONE backward (per 56 layers)."""

    @staticmethod
    def forward(ctx, n: int, *vg):
        vs = [t.detach().to(torch.float32).contiguous() for t in vg[:n]]
        gs = [t.detach().to(torch.float32).contiguous() for t in vg[n:]]
        lib = _lib()
        ws = [torch.empty_like(v) for v in vs]
        norms = [torch.empty(v.shape[0], dtype=torch.float32, device=v.device) for v in vs]
        rows = [v.shape[0] for v in vs]
        cols = [v.numel() // v.shape[0] for v in vs]
        st = ctypes.c_void_p(torch.cuda.current_stream(vs[0].device).cuda_stream)
        with _guard(vs[0]):
            for a in range(0, n, _WN_MAX):
                b = min(n, a + _WN_MAX)
                rc = lib.fastsvc_weight_norm_forward(b - a, _ptr_array(vs[a:b]), _ptr_array(gs[a:b]), _ptr_array(ws[a:b]),
                                                     _ptr_array(norms[a:b]), (ctypes.c_int32 * (b - a))(*rows[a:b]),
                                                     (ctypes.c_int32 * (b - a))(*cols[a:b]), st)
                if rc != 0:
                    raise FastSVCError(f"fastsvc_weight_norm_forward failed ({rc})")
        ctx.n, ctx.rows, ctx.cols = n, rows, cols
        ctx.save_for_backward(*vs, *gs, *norms)
        return tuple(ws)

    @staticmethod
    @once_differentiable
    def backward(ctx, *dws):
        n, rows, cols = ctx.n, ctx.rows, ctx.cols
        saved = ctx.saved_tensors
        vs, gs, norms = saved[:n], saved[n:2 * n], saved[2 * n:]
        dws = [torch.zeros_like(v) if d is None else d.to(torch.float32).contiguous() for d, v in zip(dws, vs)]
        lib = _lib()
        dvs = [torch.empty_like(v) for v in vs]
        dgs = [torch.empty_like(g) for g in gs]
        st = ctypes.c_void_p(torch.cuda.current_stream(vs[0].device).cuda_stream)
        with _guard(vs[0]):
            for a in range(0, n, _WN_MAX):
                b = min(n, a + _WN_MAX)
                rc = lib.fastsvc_weight_norm_backward(b - a, _ptr_array(vs[a:b]), _ptr_array(gs[a:b]), _ptr_array(dws[a:b]),
                                                      _ptr_array(norms[a:b]), _ptr_array(dvs[a:b]), _ptr_array(dgs[a:b]),
                                                      (ctypes.c_int32 * (b - a))(*rows[a:b]), (ctypes.c_int32 * (b - a))(*cols[a:b]), st)
                if rc != 0:
                    raise FastSVCError(f"fastsvc_weight_norm_backward failed ({rc})")
        return (None, *dvs, *dgs)


def weight_norm_fold(vs, gs):
    """``[g * v / ||v|| for v, g in zip(vs, gs)]`` (norm over all dims but 0: ``torch.nn.utils.weight_norm``, as the reference
    applies it, ``fastsvc.py:354-362``), differentiable with respect to every v and g, one HIP launch each way."""
    vs, gs = list(vs), list(gs)
    if not vs or not all(t.is_cuda for t in vs + gs):
        raise FastSVCError("weight_norm_fold (HIP) needs GPU tensors; there is no CPU fallback")
    return list(_WeightNormFn.apply(len(vs), *vs, *gs))
