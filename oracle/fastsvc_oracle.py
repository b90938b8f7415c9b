"""CPU oracle for the FastSVC generator forward pass.  TEST INFRASTRUCTURE - NOT A PRODUCT PATH.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module, and only as the checker / reported CPU baseline.  Nothing under
``svcc23_fastsvc_amd/`` imports it; the product path fails loudly without its HIP library.

What it restates (reference ``/root/reference``, package ``harana``):

* ``forward_as_executed``  - op-for-op the sequence the reference runs
  (``harana/models/fastsvc.py:305-340``): the down-sampling chains are re-evaluated from the raw
  signal for every stage (net0 x4, net1 x3, ...), the 1x1 residual conv runs at the input rate and
  is decimated afterwards (``fastsvc.py:164-167``), the FiLM sums and the speaker projection are
  recomputed three times per block (``fastsvc.py:115-140``).  This is the variant timed as the
  CPU baseline ("port" of the reference's CPU PyTorch path).
* ``forward_dedup``        - the de-duplicated dataflow of SURVEY.md Appendix B that the HIP path
  implements (each chain once, 1x1 after decimation, FiLM pre-summed).  ``return_taps=True``
  exposes every intermediate tensor for per-kernel parity tests.
* ``forward_numpy64``      - float64 numpy restatement with explicit shift-and-accumulate
  convolutions (no torch operator involved); small sizes only.  Independent cross-check of the
  two torch variants.

Arithmetic provenance: the reference's arithmetic is PyTorch aten (pinned ``torch==1.12.0`` in
``setup.py:27``; 2.10.0 here).  The reference holds no golden vectors or known-answer tests for
this path, so the oracle is pinned against outputs of the live reference generated
in the build container: ``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``
(``tests/test_oracle_golden.py`` checks them on every machine, and checks the live reference
directly when ``/root/reference`` is present).

Weights are passed as a *folded* dict (``.weight`` / ``.bias`` keys, i.e. the state-dict layout
after ``remove_weight_norm()``, ``fastsvc.py:342-352``); use
``svcc23_fastsvc_amd.synth.fold_weight_norm`` for ``weight_g``/``weight_v`` checkpoints.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.2      # every nn.LeakyReLU in the generator (fastsvc.py:57-71,172-177,213)
IN_EPS = 1e-5          # nn.InstanceNorm2d default eps (fastsvc.py:76)
L2_EPS = 1e-12         # F.normalize default eps (fastsvc.py:136)


def _as_torch(w: Dict[str, np.ndarray], dtype=torch.float32) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in w.items():
        t = torch.as_tensor(np.asarray(v)).to(dtype)
        if t.dim() == 4:           # Conv2d (Cout, Cin, 1, 3) -> Conv1d (Cout, Cin, 3)
            t = t.squeeze(2)
        out[k] = t.contiguous()
    return out


def _conv(x, w, prefix, dilation=1):
    """Cross-correlation with zero 'same' padding (= dilation for k=3, 0 for k=1) and bias
    (Conv1d1x3 / Conv2d1x3 ``upsample.py:76-83,99-106``; Conv1d1x1 ``residual_block.py:41-48``)."""
    weight = w[prefix + ".weight"]
    k = weight.shape[-1]
    return F.conv1d(x, weight, w[prefix + ".bias"], padding=(k // 2) * dilation, dilation=dilation)


def _lrelu(x):
    return F.leaky_relu(x, LRELU_SLOPE)


def _squeeze(x, s):
    """Squeeze2d (upsample.py:53-74): nearest interpolation to size int(T/s) == x[..., ::s]."""
    size = int(x.shape[-1] / s)
    return x[..., ::s][..., :size]


def _stretch(x, s):
    """Stretch2d (upsample.py:21-50): nearest up-sampling by an integer == repeat_interleave."""
    return torch.repeat_interleave(x, s, dim=-1)


def _down_net_as_executed(x, w, prefix, s):
    """FastSVCDownsampleNet.forward (fastsvc.py:180-193)."""
    r = _squeeze(_conv(x, w, f"{prefix}.residual_block.0"), s)
    h = _lrelu(_squeeze(x, s))
    h = _conv(h, w, f"{prefix}.downsample_block.2", 1)
    h = _conv(_lrelu(h), w, f"{prefix}.downsample_block.4", 2)
    h = _conv(_lrelu(h), w, f"{prefix}.downsample_block.6", 4)
    return h + r


def _down_net_dedup(x, w, prefix, s):
    """Same values with the 1x1 commuted past the decimation."""
    xd = _squeeze(x, s)
    r = _conv(xd, w, f"{prefix}.residual_block.0")
    h = _conv(_lrelu(xd), w, f"{prefix}.downsample_block.2", 1)
    h = _conv(_lrelu(h), w, f"{prefix}.downsample_block.4", 2)
    h = _conv(_lrelu(h), w, f"{prefix}.downsample_block.6", 4)
    return h + r


def _film(x, w, prefix):
    """FastSVCFiLMNet.forward (fastsvc.py:220-232)."""
    h = _lrelu(_conv(x, w, f"{prefix}.conv"))
    return _conv(h, w, f"{prefix}.conv_scale"), _conv(h, w, f"{prefix}.conv_shift")


def _instance_norm(x):
    """InstanceNorm2d without affine / running stats: per (b, c) over the whole time axis,
    biased variance, eps 1e-5 (fastsvc.py:76,138)."""
    mean = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + IN_EPS)


def _speaker_bias(spk_emb, w, prefix):
    """emb_projector(F.normalize(spk_emb)) (fastsvc.py:135-137) -> (B, C, 1)."""
    e = spk_emb / spk_emb.norm(dim=1, keepdim=True).clamp_min(L2_EPS)
    return F.linear(e, w[f"{prefix}.emb_projector.weight"], w[f"{prefix}.emb_projector.bias"]).unsqueeze(-1)


def _affine_as_executed(x, sine, lft, spk_emb, w, prefix):
    """FastSVCUpsampleNet._feature_affine (fastsvc.py:115-140), recomputing the sums each call."""
    scale = sine[0] + lft[0]
    shift = sine[1] + lft[1]
    x = scale * x
    x = x + shift
    if spk_emb is not None:
        p = _speaker_bias(spk_emb, w, prefix)
        x = _instance_norm(x)
        x = x + p
    return x


def _up_net_as_executed(x, sine, lft, spk_emb, w, prefix, s):
    """FastSVCUpsampleNet.forward (fastsvc.py:80-113); the dummy height axis is dropped."""
    a = _conv(x, w, f"{prefix}.conv_first", 1)
    xr = _conv(_stretch(a, s), w, f"{prefix}.residual_block.1", 1)
    t = _lrelu(_conv(_stretch(_lrelu(a), s), w, f"{prefix}.upsample_block0.2", 1))
    t = _affine_as_executed(t, sine, lft, spk_emb, w, prefix)
    t = _conv(_lrelu(t), w, f"{prefix}.conv_block1.1", 3)
    x_ = t + xr
    t = _affine_as_executed(x_, sine, lft, spk_emb, w, prefix)
    t = _conv(_lrelu(t), w, f"{prefix}.conv_block2.1", 9)
    t = _affine_as_executed(t, sine, lft, spk_emb, w, prefix)
    t = _conv(_lrelu(t), w, f"{prefix}.conv_block3.1", 27)
    return t + x_


def _n_stages(w) -> int:
    n = 0
    while f"upsampling_nets.{n}.conv_first.weight" in w:
        n += 1
    return n


def _check_shapes(x, s, l, scales):
    hop = int(np.prod(scales))
    if s.shape[-1] != l.shape[-1] or s.shape[-1] != x.shape[-1] * hop:
        raise ValueError(f"length mismatch: ppg frames {x.shape[-1]} x hop {hop} vs sine {s.shape[-1]} / lft {l.shape[-1]}")


@torch.no_grad()
def forward_as_executed(weights: Dict[str, np.ndarray], scales, x, s, l, spk_emb=None,
                        dtype=torch.float32):
    """FastSVCGenerator.forward exactly as the reference executes it (fastsvc.py:305-340).

    x (B, C_in, F) - s, l (B, 1, hop*F) - spk_emb (B, E) or None  ->  (B, 1, hop*F) torch tensor.
    """
    w = _as_torch(weights, dtype)
    x, s, l = (torch.as_tensor(t).to(dtype) for t in (x, s, l))
    if spk_emb is not None:
        spk_emb = torch.as_tensor(spk_emb).to(dtype)
    scales = list(scales)
    _check_shapes(x, s, l, scales)
    n = len(scales)
    down_scales = [1] + scales[::-1][:-1]

    def chain(sig, name, didx):            # downsampling_loop, fastsvc.py:334-340
        h = sig
        for k in range(didx + 1):
            h = _down_net_as_executed(h, w, f"downsampling_{name}.{k}", down_scales[k])
        return h

    for i in range(n):
        didx = n - i - 1
        lft = _film(chain(l, "lft", didx), w, f"film_lft.{didx}")
        sine = _film(chain(s, "sine", didx), w, f"film_sine.{didx}")
        x = _up_net_as_executed(x, sine, lft, spk_emb, w, f"upsampling_nets.{i}", scales[i])
    return _conv(x, w, "conv_last")


@torch.no_grad()
def forward_dedup(weights: Dict[str, np.ndarray], scales, x, s, l, spk_emb=None,
                  dtype=torch.float32, return_taps: bool = False):
    """De-duplicated dataflow - the one the HIP path implements."""
    w = _as_torch(weights, dtype)
    x, s, l = (torch.as_tensor(t).to(dtype) for t in (x, s, l))
    if spk_emb is not None:
        spk_emb = torch.as_tensor(spk_emb).to(dtype)
    scales = list(scales)
    _check_shapes(x, s, l, scales)
    n = len(scales)
    down_scales = [1] + scales[::-1][:-1]
    taps = {}

    scale_sum = [None] * n
    shift_sum = [None] * n
    for name, sig in (("lft", l), ("sine", s)):
        h = sig
        for k in range(n):
            h = _down_net_dedup(h, w, f"downsampling_{name}.{k}", down_scales[k])
            sc, sh = _film(h, w, f"film_{name}.{k}")
            taps[f"down_{name}.{k}"] = h
            taps[f"film_{name}.{k}.scale"] = sc
            taps[f"film_{name}.{k}.shift"] = sh
            scale_sum[k] = sc if scale_sum[k] is None else scale_sum[k] + sc
            shift_sum[k] = sh if shift_sum[k] is None else shift_sum[k] + sh
    # reference sums sine + lft (fastsvc.py:129-130); float addition is commutative
    for k in range(n):
        taps[f"scale.{k}"] = scale_sum[k]
        taps[f"shift.{k}"] = shift_sum[k]

    for i in range(n):
        k = n - i - 1
        sc, sh = scale_sum[k], shift_sum[k]
        prefix = f"upsampling_nets.{i}"
        p = _speaker_bias(spk_emb, w, prefix) if spk_emb is not None else None

        def aff(t):
            u = sc * t + sh
            if p is not None:
                u = _instance_norm(u) + p
            return u

        a = _conv(x, w, f"{prefix}.conv_first", 1)
        xr = _conv(_stretch(a, scales[i]), w, f"{prefix}.residual_block.1", 1)
        t0 = _lrelu(_conv(_stretch(_lrelu(a), scales[i]), w, f"{prefix}.upsample_block0.2", 1))
        t1 = _conv(_lrelu(aff(t0)), w, f"{prefix}.conv_block1.1", 3)
        x_ = t1 + xr
        t2 = _conv(_lrelu(aff(x_)), w, f"{prefix}.conv_block2.1", 9)
        t3 = _conv(_lrelu(aff(t2)), w, f"{prefix}.conv_block3.1", 27)
        x = t3 + x_
        taps[f"up.{i}.a"] = a
        taps[f"up.{i}.xr"] = xr
        taps[f"up.{i}.t0"] = t0
        taps[f"up.{i}.xmid"] = x_
        taps[f"up.{i}.t2"] = t2
        taps[f"up.{i}.out"] = x
        # FiLM-affined tensors (before InstanceNorm) - what the HIP path materialises
        taps[f"up.{i}.u1"] = sc * t0 + sh
        taps[f"up.{i}.u2"] = sc * x_ + sh
        taps[f"up.{i}.u3"] = sc * t2 + sh
        if p is not None:
            taps[f"up.{i}.spk"] = p.squeeze(-1)
    y = _conv(x, w, "conv_last")
    if return_taps:
        return y, taps
    return y


# --------------------------------------------------------------------------------------------
# float64 numpy restatement (no torch operators) - small sizes only
# --------------------------------------------------------------------------------------------
def _np_conv(x, weight, bias, dilation=1):
    """x (B, Cin, T), weight (Cout, Cin, K) -> (B, Cout, T); zero 'same' padding."""
    B, Cin, T = x.shape
    Cout, _, K = weight.shape
    pad = (K // 2) * dilation
    xp = np.zeros((B, Cin, T + 2 * pad), dtype=np.float64)
    xp[:, :, pad:pad + T] = x
    y = np.zeros((B, Cout, T), dtype=np.float64)
    for j in range(K):
        seg = xp[:, :, j * dilation: j * dilation + T]
        y += np.einsum("oc,bct->bot", weight[:, :, j], seg)
    return y + bias[None, :, None]


def forward_numpy64(weights: Dict[str, np.ndarray], scales, x, s, l, spk_emb=None) -> np.ndarray:
    """Direct float64 evaluation of SURVEY.md Appendix B with explicit loops over taps."""
    w = {}
    for k, v in weights.items():
        v = np.asarray(v, dtype=np.float64)
        if v.ndim == 4:
            v = v[:, :, 0, :]
        w[k] = v
    x, s, l = (np.asarray(t, dtype=np.float64) for t in (x, s, l))
    scales = list(scales)
    n = len(scales)
    down_scales = [1] + scales[::-1][:-1]

    def conv(t, prefix, d=1):
        return _np_conv(t, w[prefix + ".weight"], w[prefix + ".bias"], d)

    def lrelu(t):
        return np.where(t >= 0, t, LRELU_SLOPE * t)

    sc_sum, sh_sum = [0.0] * n, [0.0] * n
    for name, sig in (("lft", l), ("sine", s)):
        h = sig
        for k in range(n):
            size = int(h.shape[-1] / down_scales[k])
            hd = h[..., ::down_scales[k]][..., :size]
            p = f"downsampling_{name}.{k}"
            r = conv(hd, f"{p}.residual_block.0")
            t = conv(lrelu(hd), f"{p}.downsample_block.2", 1)
            t = conv(lrelu(t), f"{p}.downsample_block.4", 2)
            t = conv(lrelu(t), f"{p}.downsample_block.6", 4)
            h = t + r
            u = lrelu(conv(h, f"film_{name}.{k}.conv"))
            sc_sum[k] = sc_sum[k] + conv(u, f"film_{name}.{k}.conv_scale")
            sh_sum[k] = sh_sum[k] + conv(u, f"film_{name}.{k}.conv_shift")
    for i in range(n):
        k = n - i - 1
        prefix = f"upsampling_nets.{i}"
        p = None
        if spk_emb is not None:
            e = np.asarray(spk_emb, dtype=np.float64)
            e = e / np.maximum(np.sqrt((e * e).sum(axis=1, keepdims=True)), L2_EPS)
            p = (e @ w[f"{prefix}.emb_projector.weight"].T + w[f"{prefix}.emb_projector.bias"])[:, :, None]

        def aff(t):
            u = sc_sum[k] * t + sh_sum[k]
            if p is not None:
                m = u.mean(axis=-1, keepdims=True)
                v = ((u - m) ** 2).mean(axis=-1, keepdims=True)
                u = (u - m) / np.sqrt(v + IN_EPS) + p
            return u

        a = conv(x, f"{prefix}.conv_first")
        xr = conv(np.repeat(a, scales[i], axis=-1), f"{prefix}.residual_block.1")
        t = lrelu(conv(np.repeat(lrelu(a), scales[i], axis=-1), f"{prefix}.upsample_block0.2"))
        t = conv(lrelu(aff(t)), f"{prefix}.conv_block1.1", 3)
        x_ = t + xr
        t = conv(lrelu(aff(x_)), f"{prefix}.conv_block2.1", 9)
        t = conv(lrelu(aff(t)), f"{prefix}.conv_block3.1", 27)
        x = t + x_
    return conv(x, "conv_last")
